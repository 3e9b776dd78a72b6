"""What the dashboard's `update_graphs` (app/dashboard/callbacks.py:95-196) reads from the streamer and plots, replayed
call by call against the record of the reference's own callback (tests/golden/ref_update_graphs.npz, made by
oracle/make_golden_dashboard.py).  The replay is the callback WITH the change INTEGRATION.md describes — a
`WaterfallBuffer`-like object in place of `deque(maxlen=100)` / `np.array(deque)` — and nothing of the reference is
needed to run it: streamer and waterfall are handed in, so the same function checks the host logic on CPU (oracle
transform, oracle deque) and the HIP path on the GPU box (SpectrumStreamer + WaterfallBuffer)."""
import json

import numpy as np

from oracle import cpu_ref
from oracle.make_golden_dashboard import BurstRadio, run_burst, scenario_frames


def scenarios(g):
    return json.loads(str(g["scenarios_json"]))


def replay(g, sc, make_streamer, make_waterfall, check_row, check_peaks=True):
    """Run scenario `sc` through `make_streamer(radio)` / `make_waterfall(nfft)` and compare every call with the record.
    `check_row(got, ref, what)` compares a power_db row (exact for the oracle, parity bar for the GPU)."""
    k = sc["name"]
    power_db, trace_x, heat_x = g[f"{k}/power_db"], g[f"{k}/trace_x"], g[f"{k}/heat_x"]
    kind, frame = g[f"{k}/call_kind"], g[f"{k}/call_frame"]
    ragged = {key: (g[f"{k}/{key}_concat"], g[f"{k}/{key}_off"]) for key in ("peaks", "z_ids", "y")}
    part = lambda key, c: ragged[key][0][ragged[key][1][c]: ragged[key][1][c + 1]]        # noqa: E731
    radio = BurstRadio(scenario_frames(sc))
    streamer = make_streamer(radio)
    wf = make_waterfall(sc["nfft"])
    wf_x = None
    c = 0
    drawn = 0
    for count in sc["bursts"]:
        run_burst(streamer, radio, count)
        while True:
            this = int(kind[c])
            if this == 0:                                   # paused: the callback returns before touching the streamer (:99-101)
                c += 1
                continue
            data = streamer.get_latest_data()               # :104
            if this == 1:                                   # "Waiting for data..." (:106-108)
                assert data is None, (k, c)
                c += 1
                break
            assert data is not None, (k, c, "the reference's callback drew a frame here")
            f = int(frame[c])
            assert data["samples"] is radio.frames[f] or np.array_equal(data["samples"], radio.frames[f]), (k, c)
            check_row(data["power_db"], power_db[f], f"{k} call {c} frame {f}")
            assert np.array_equal(data["freqs"] / 1e6, trace_x), (k, c)                      # :141
            if check_peaks:
                assert np.array_equal(cpu_ref.dashboard_peak_markers(data["power_db"]), part("peaks", c)), (k, c)   # :148-163
            wf.append(data["power_db"])                     # :176
            if wf_x is None:
                wf_x = data["freqs"] / 1e6                  # :177-178
            ids = part("z_ids", c)
            if len(wf) > 1:                                 # :181
                z = wf.as_array()                           # :182
                assert z.shape == (len(ids), sc["nfft"]) and z.dtype == np.float32, (k, c, z.shape)
                check_row(z, power_db[ids], f"{k} call {c} heatmap")
                assert np.array_equal(wf_x, heat_x)         # :185
                assert list(range(len(wf))) == list(part("y", c))                           # :186
            else:
                assert len(ids) == 0, (k, c)
            drawn += 1
            c += 1
    assert c == len(kind), (k, c, len(kind))
    return drawn
