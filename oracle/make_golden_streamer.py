#!/usr/bin/env python3
"""Generate tests/golden/ref_streamer_failures.json by running the REFERENCE's reader loop
(app/sdr/streamer.py:95-174, `_stream_data`, with `_attempt_reconnect` :83-93) under scripted failures.

TEST INFRASTRUCTURE.  Runs only in the build container (it needs /root/reference); the committed JSON —
scripts in, event traces out: data only — is what travels.  Same set-up as oracle/make_golden.py
(`sys.modules['adi'] = MagicMock()`, the reference's own test idiom), plus:
  * a fake `sdr` whose `rx()` follows a script: "ok" returns a frame, {"raise": "OSError", "errno": 110}
    raises, {"return": "garbage"} returns None (the transform then fails); when the script is used up it
    clears `running` and returns one last frame;
  * `streamer.reconnect` replaced by a scripted stand-in (the real one needs a radio): each call takes the
    next boolean of the scenario's list (False once exhausted), sets `connected` like `connect()` does
    (:41-47) and re-installs the fake sdr on success;
  * `time.sleep` of the reference module replaced by a recorder (no real waiting).
The trace lists, in order, every sleep (seconds), every reconnect call (outcome) and every frame pushed,
then the final `running` / `connected` / `total_frames`.  tests/test_next_rows_cpu.py replays the same
scripts through sdr_iq_visualizer_amd.streaming.SpectrumStreamer and demands the same trace.

    python oracle/make_golden_streamer.py
"""
from __future__ import annotations

import json
import os
import sys
from unittest.mock import MagicMock

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SDRK_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden", "ref_streamer_failures.json")

SCENARIOS = [
    {"name": "three_generic_errors_then_reconnect_ok",
     "rx": [{"raise": "Exception"}] * 3 + ["ok", "ok"], "reconnect": [True]},
    {"name": "generic_errors_forever_reconnect_never",
     "rx": [{"raise": "Exception"}] * 6, "reconnect": []},
    {"name": "two_errors_success_resets_backoff",
     "rx": [{"raise": "Exception"}, {"raise": "Exception"}, "ok", {"raise": "Exception"}, "ok"], "reconnect": []},
    {"name": "fatal_errno9_reconnect_ok_first_try",
     "rx": ["ok", {"raise": "OSError", "errno": 9}, "ok"], "reconnect": [True]},
    {"name": "fatal_errno10054_reconnect_third_try",
     "rx": [{"raise": "OSError", "errno": 10054}, "ok"], "reconnect": [False, False, True]},
    {"name": "fatal_errno9_reconnect_fails",
     "rx": [{"raise": "OSError", "errno": 9}, "ok"], "reconnect": []},
    {"name": "timeout_errno110_reconnect_ok",
     "rx": [{"raise": "OSError", "errno": 110}, "ok"], "reconnect": [True]},
    {"name": "unreachable_errno113_reconnect_fails_then_backoff",
     "rx": [{"raise": "OSError", "errno": 113}, "ok", "ok"], "reconnect": []},
    {"name": "timeouts_three_times_reconnects_all_fail",
     "rx": [{"raise": "OSError", "errno": 110}] * 3 + ["ok"], "reconnect": []},
    {"name": "other_oserror_errno5_is_a_plain_error",
     "rx": [{"raise": "OSError", "errno": 5}, "ok"], "reconnect": []},
    {"name": "oserror_without_errno",
     "rx": [{"raise": "OSError"}, {"raise": "OSError"}, "ok"], "reconnect": []},
    {"name": "valueerror_and_runtimeerror_are_generic",
     "rx": [{"raise": "ValueError"}, {"raise": "RuntimeError"}, "ok"], "reconnect": []},
    {"name": "starts_disconnected_reconnect_second_try",
     "rx": ["ok", "ok"], "reconnect": [False, True], "start_connected": False},
    {"name": "starts_disconnected_never_reconnects",
     "rx": ["ok"], "reconnect": [], "start_connected": False},
    # rx() succeeds but hands back something the transform chokes on: the counters were already reset after rx()
    # (:115-116), so every such frame counts as error #1 and sleeps 0.2 s — it never escalates to a reconnect
    {"name": "garbage_frames_never_escalate",
     "rx": [{"return": "garbage"}] * 4 + ["ok"], "reconnect": []},
    {"name": "four_errors_reconnect_ok_then_more_errors",
     "rx": [{"raise": "Exception"}] * 3 + [{"raise": "Exception"}, "ok"], "reconnect": [True]},
]

ERRORS = {"Exception": Exception, "OSError": OSError, "ValueError": ValueError, "RuntimeError": RuntimeError}


def make_exception(step):
    cls = ERRORS[step["raise"]]
    if "errno" in step:
        return cls(step["errno"], "scripted failure")
    return cls("scripted failure")


class ScriptedSdr:
    def __init__(self, streamer, script, trace):
        self.streamer, self.script, self.trace, self.i = streamer, list(script), trace, 0
        self.frame = np.ones(64, dtype=np.complex64)

    def rx(self):
        if self.i >= len(self.script):
            self.streamer.running = False          # script used up: one last good frame ends the loop
            return self.frame
        step = self.script[self.i]
        self.i += 1
        if step == "ok":
            return self.frame
        if "return" in step:
            return None                             # np.fft.fft(None) raises inside the loop body (:119)
        raise make_exception(step)


def run_scenario(module, S, sc):
    trace = []
    s = S()
    sdr = ScriptedSdr(s, sc["rx"], trace)
    s.sdr = sdr
    s.connected = bool(sc.get("start_connected", True))
    outcomes = list(sc["reconnect"])

    def fake_reconnect():
        ok = outcomes.pop(0) if outcomes else False
        trace.append(["reconnect", ok])
        s.connected = ok                            # connect() sets it either way (:41,46)
        if ok:
            s.sdr = sdr
        return ok

    s.reconnect = fake_reconnect
    pushed = s._push

    def recording_push(data):
        trace.append(["frame"])
        pushed(data)

    s._push = recording_push
    real_sleep = module.time.sleep
    module.time.sleep = lambda d: trace.append(["sleep", round(float(d), 6)])
    try:
        s.running = True
        s._stream_data()
    finally:
        module.time.sleep = real_sleep
    return {"events": trace, "running": bool(s.running), "connected": bool(s.connected),
            "total_frames": int(s.total_frames)}


def main():
    if "adi" not in sys.modules:
        sys.modules["adi"] = MagicMock()
    sys.path.insert(0, REF)
    import app.sdr.streamer as module
    import logging
    logging.disable(logging.CRITICAL)
    out = {"source": "app/sdr/streamer.py:83-174 driven by oracle/make_golden_streamer.py", "scenarios": []}
    for sc in SCENARIOS:
        rec = dict(sc)
        rec["trace"] = run_scenario(module, module.SDRDataStreamer, sc)
        out["scenarios"].append(rec)
        print(sc["name"], rec["trace"]["events"], {k: v for k, v in rec["trace"].items() if k != "events"})
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
