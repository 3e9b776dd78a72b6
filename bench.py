#!/usr/bin/env python3
"""Headline benchmark: complex64 IQ Msamples/s through Hann-windowed 4096-point
FFT + log-PSD (BASELINE.json configs[1]: 2^20 synthetic frames per GPU), with the
dominant kernel's achieved HBM GB/s against the 8 TB/s roofline and the numpy
oracle timed on the host cores of the same box.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
one rank per GPU; rank g owns frames [g*2^20, (g+1)*2^20) of BASELINE.json config 4
(frame-range sharding, weak scaling, no data-path collective: the only RCCL traffic
is the barrier and the max-reduce of the elapsed time; if RCCL does not come up on
every rank alike, those move to gloo and config.rendezvous_note says so).

A "step" is one pass of the hot path over the rank's device-resident batch.  Rank 0
prints ONE JSON line.  At N = 1 the same line also carries:
  launch_ms            min / median / max of the K timed launches (HIP events between launches)
  roofline.measured_copy_GBps / frac_of_measured_copy
                       a no-arithmetic streaming kernel with the same bytes, same buffers, same process
  telemetry            shader/memory clocks and socket power sampled around and during the timed region
  cpu_baseline         the oracle on one core, plus `all_cores` (one worker per CPU this process may use)
  secondary            BASELINE configs 3 and 5 (device resident), the numpy boundary (PCIe inclusive) and
                       the per-row feature reductions — each with its own algorithmic-bytes formula
`--no-secondary` skips those extra legs.  At N > 1 the only extra leg is `secondary.config5_channels`: one continuous
N = 2^20 channel per GPU (BASELINE.json configs[4]), all ranks at the same time, a few tens of milliseconds.
"""
from __future__ import annotations

import argparse
import ctypes
import glob
import json
import os
import statistics
import sys
import threading
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

NFFT = 4096
ALGO_BYTES_PER_SAMPLE = 12          # 8 B complex64 read + 4 B float32 written
HBM_PEAK_GBPS = 8000.0              # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
METRIC = "complex64 IQ Msamples/s through windowed FFT+log-PSD, N=4096; % HBM3E peak"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=1 << 20, help="frames per GPU (default 2^20)")
    ap.add_argument("--first-frame", type=int, default=0,
                    help="frame number of rank 0's first frame (rank g owns [first + g*F, first + (g+1)*F)); tests use it to put "
                         "small runs past sample 2^32, where config 4's ranks 1..7 live")
    ap.add_argument("--window", default="hann", choices=["hann", "rect"])
    ap.add_argument("--cpu-seconds", type=float, default=10.0,
                    help="budget for the single-core numpy cpu_baseline leg (0 disables both CPU legs)")
    ap.add_argument("--cpu-all-cores-seconds", type=float, default=5.0,
                    help="budget for the all-cores leg (0 disables it; pure CPU, forked before any GPU runtime loads)")
    ap.add_argument("--cpu-large-seconds", type=float, default=2.5,
                    help="budget of EACH of the four CPU legs beside configs 3 and 5 (one core / all cores; 0 disables)")
    ap.add_argument("--parity-frames", type=int, default=1024,
                    help="random frames of the timed output checked against the oracle (SURVEY.md §8d: >= 1024)")
    ap.add_argument("--placement-candidates", type=int, default=6,
                    help="output-buffer placements probed for the resident pair (1 = plain allocation).  The level a pair runs at is a "
                         "lottery (DESIGN.md 4.1): of the 144 pairings probed in rounds 3-5, 3 % (one box) to 22 % (another) were on the "
                         "fast level.  Six since round 6 (round 5 drew ten: +15 s of wall time, and the driver's series 0.714 / 0.693 / "
                         "0.706 could not see a gain — the box-to-box spread is larger)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip configs 3/5, the numpy boundary and the feature-reduction legs")
    return ap.parse_args()


# ---- CPU baseline (oracle = checker; a reported number, not the target) -----------------------

def cpu_baseline(window: str, seconds: float):
    """The oracle (oracle/cpu_ref.py, the numpy restatement of streamer.py:119-121) on a
    bounded sample of the same workload, single host core."""
    import numpy as np
    from oracle import cpu_ref
    from sdr_iq_visualizer_amd import synth

    chunk, n_chunks = 128, 16                       # 2^11 frames = 8.4 Msamples per pass
    frames = [synth.synth_iq(1234, c * chunk, chunk, NFFT) for c in range(n_chunks)]
    w = np.hanning(NFFT).astype(np.float32) if window == "hann" else None
    cpu_ref.spectrum_db(frames[0], window=w)        # warm-up
    done, t0 = 0, time.perf_counter()
    while True:
        for f in frames:
            cpu_ref.spectrum_db(f, window=w)
        done += chunk * n_chunks
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    batched = done * NFFT / dt / 1e6
    # the reference's literal per-frame expression (rectangular, as it ships)
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < min(2.0, seconds / 4):
        for f in frames[0]:
            20 * np.log10(np.abs(np.fft.fftshift(np.fft.fft(f))) + 1e-12)
        k += chunk
    per_frame = k * NFFT / (time.perf_counter() - t0) / 1e6
    return {
        "value": round(batched, 2),
        "unit": "Msamples/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{done} frames x {NFFT} ({window}), oracle/cpu_ref.spectrum_db in chunks of {chunk} "
                  f"for {dt:.1f} s, numpy {np.__version__}",
        "per_frame_reference_expression_rect": round(per_frame, 2),
    }


def _cpu_worker(job):
    """One forked worker of the all-cores leg: the batched oracle on its own frames for `seconds`."""
    import numpy as np
    from oracle import cpu_ref
    from sdr_iq_visualizer_amd import synth
    seed, window, seconds = job
    frames = [synth.synth_iq(seed, c * 128, 128, NFFT) for c in range(4)]
    w = np.hanning(NFFT).astype(np.float32) if window == "hann" else None
    cpu_ref.spectrum_db(frames[0], window=w)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for f in frames:
            cpu_ref.spectrum_db(f, window=w)
        done += 128 * len(frames)
    return done, time.perf_counter() - t0


def usable_cpus():
    """CPUs this process can actually run on at once: the affinity mask capped by the cgroup CPU quota."""
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota = None if txt[0] == "max" else int(txt[0]) / int(txt[1])
            else:
                q = int(txt[0])
                quota = None if q <= 0 else q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    workers = affinity if quota is None else max(1, min(affinity, int(quota)))
    return workers, affinity, quota


def cpu_baseline_all_cores(window: str, seconds: float):
    import multiprocessing as mp
    workers, affinity, quota = usable_cpus()
    with mp.get_context("fork").Pool(workers) as pool:
        res = pool.map(_cpu_worker, [(1234 + i, window, seconds) for i in range(workers)])
    total = sum(r[0] for r in res) * NFFT / max(r[1] for r in res) / 1e6
    return {"value": round(total, 1), "unit": "Msamples/s", "cores": workers,
            "visible_cpus": affinity, "cgroup_cpu_quota": None if quota is None else round(quota, 1),
            "sample": f"one forked oracle worker per usable CPU for {seconds:.0f} s each, same generator and window"}


def _large_frame_loop(job):
    """The oracle on one large-frame configuration for `seconds` (one core; also the all-cores worker): the stream is
    made of the same generator frames the device legs use (seed 99), `rows` frames of `nfft` at spacing `hop`;
    hop < nfft goes through cpu_ref.stft_db (row r = spectrum_db(iq[r*hop : r*hop + nfft]), the reference
    expression of streamer.py:119-121 per row), packed frames through cpu_ref.spectrum_db one frame at a time."""
    import numpy as np
    from oracle import cpu_ref
    from sdr_iq_visualizer_amd import synth
    seed, nfft, hop, rows, window, seconds = job
    in_samples = (rows - 1) * hop + nfft
    iq = synth.synth_iq(seed, 0, (in_samples + 4095) // 4096, 4096).reshape(-1)[:in_samples]
    w = np.hanning(nfft).astype(np.float32) if window == "hann" else None
    cpu_ref.spectrum_db(iq[:nfft], window=w)                                  # warm-up (pocketfft plan, pages)
    done, t0 = 0, time.perf_counter()
    while True:
        if hop < nfft:
            cpu_ref.stft_db(iq, nfft, hop, window=w)
        else:
            for r in range(rows):
                cpu_ref.spectrum_db(iq[r * hop: r * hop + nfft], window=w)
        done += rows
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return done, dt


def cpu_baseline_large(nfft, hop, rows, window, seconds, all_cores_seconds):
    """BASELINE.md §3 item 4 / SURVEY.md §8(d) "CPU baseline": the reference expression timed at the large-frame
    configurations in the same run — one core, then one forked worker per usable CPU (pure numpy, before any GPU
    runtime is up).  Rates in frame-samples/s (rows*N per second, the unit of `frame_Msamples_per_s` beside it)."""
    import numpy as np
    done, dt = _large_frame_loop((99, nfft, hop, rows, window, seconds))
    out = {"value": round(done * nfft / dt / 1e6, 2), "unit": "Msamples/s (frame-samples: rows*N per second)",
           "cores": 1, "kind": "port",
           "input_stream_Msamples_per_s": round(done * hop / dt / 1e6, 2),
           "sample": f"{done} rows of N={nfft} at hop {hop} ({window}) in {dt:.1f} s: oracle/cpu_ref."
                     + ("stft_db" if hop < nfft else "spectrum_db per frame")
                     + f" over a {rows}-row stream, numpy {np.__version__}"}
    if all_cores_seconds > 0:
        import multiprocessing as mp
        workers, affinity, quota = usable_cpus()
        with mp.get_context("fork").Pool(workers) as pool:
            res = pool.map(_large_frame_loop, [(99 + i, nfft, hop, rows, window, all_cores_seconds) for i in range(workers)])
        total = sum(r[0] for r in res) * nfft / max(r[1] for r in res) / 1e6
        out["all_cores"] = {"value": round(total, 1), "unit": "Msamples/s (frame-samples)", "cores": workers,
                            "visible_cpus": affinity, "cgroup_cpu_quota": None if quota is None else round(quota, 1),
                            "sample": f"one forked oracle worker per usable CPU for {all_cores_seconds:.0f} s each, own stream per worker"}
    return out


def cpu_baselines_secondary(seconds=2.5, all_cores_seconds=2.5):
    """The host numbers that stand beside secondary.config3 / config5 (about 4 x 2.5 s in all)."""
    return {"config3": cpu_baseline_large(65536, 32768, 64, "hann", seconds, all_cores_seconds),
            "config5": cpu_baseline_large(1 << 20, 1 << 20, 4, "hann", seconds, all_cores_seconds)}


# ---- clock / power telemetry -----------------------------------------------------------------

class Telemetry:
    """sysfs readings of the GPU under test (matched by PCI address): current sclk / mclk level and the
    socket power, before / during / after the timed region.  Everything is best effort: a missing file
    yields null fields, never an error."""

    def __init__(self, device_info: str):
        self.dir = None
        pci = device_info.rsplit("pci ", 1)[-1].strip() if "pci " in device_info else None
        for d in sorted(glob.glob("/sys/class/drm/card*/device")):
            try:
                if pci and os.path.basename(os.path.realpath(d)).lower() == pci.lower():
                    self.dir = d
                    break
            except OSError:
                continue
        if self.dir is None:
            cands = [d for d in sorted(glob.glob("/sys/class/drm/card*/device")) if os.path.exists(d + "/pp_dpm_sclk")]
            self.dir = cands[0] if len(cands) == 1 else None
        self.hwmon = None
        if self.dir:
            h = sorted(glob.glob(self.dir + "/hwmon/hwmon*"))
            self.hwmon = h[0] if h else None
        self.samples = []
        self._stop = threading.Event()
        self._thread = None

    @staticmethod
    def _read(path):
        try:
            return open(path).read()
        except Exception:
            return None

    def _dpm(self, name):
        txt = self._read(f"{self.dir}/{name}") if self.dir else None
        if not txt:
            return None
        for line in txt.splitlines():
            if line.rstrip().endswith("*"):
                try:
                    return int(line.split(":")[1].strip().split("M")[0])
                except Exception:
                    return None
        return None

    def _hw(self, *names):
        if not self.hwmon:
            return None
        for n in names:
            txt = self._read(f"{self.hwmon}/{n}")
            if txt:
                try:
                    return int(txt.strip())
                except ValueError:
                    continue
        return None

    def snapshot(self):
        sclk_hz = self._hw("freq1_input")
        p = self._hw("power1_average", "power1_input")
        return {"sclk_mhz": self._dpm("pp_dpm_sclk") if sclk_hz is None else round(sclk_hz / 1e6),
                "mclk_mhz": self._dpm("pp_dpm_mclk"), "fclk_mhz": self._dpm("pp_dpm_fclk"),
                "power_w": None if p is None else round(p / 1e6, 1)}

    def start(self, period=0.002):
        if not self.dir:
            return

        def run():
            while not self._stop.is_set():
                self.samples.append(self.snapshot())
                time.sleep(period)

        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()

    def stop(self):
        self._stop.set()
        if self._thread:
            self._thread.join()

    def summary(self):
        def agg(key):
            v = [s[key] for s in self.samples if s.get(key) is not None]
            return None if not v else {"min": min(v), "mean": round(sum(v) / len(v), 1), "max": max(v)}
        return {"n": len(self.samples), "sclk_mhz": agg("sclk_mhz"), "mclk_mhz": agg("mclk_mhz"),
                "power_w": agg("power_w")}


# ---- secondary workloads (rank 0, N = 1) ---------------------------------------------------------

def _median(v):
    return float(statistics.median(v))


def warm_up_by_time(fn, ms=100.0, at_least=3):
    """Call fn() (a synchronous GPU step) until `ms` have passed: the shader clock of an idle MI355X sits near 1.0-1.4 GHz and
    reaches its sustained 1.85-2.0 GHz only after tens of milliseconds of load (round 5: thirty N = 2^20 transforms from idle
    take 1.53, 1.46, 1.46, 1.45, 1.42 ... 1.35 ms), so a leg that times its first few launches measures the ramp."""
    t0, n = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < ms or n < at_least:
        fn()
        n += 1
    return n


def device_config(lib, _ffi, SpectrumPlan, dev, nfft, n_frames, stride, window, reps, scratch_candidates=6):
    """A device-resident run of one large-frame configuration: median launch time over `reps`, after the plan's
    scratch has been placed on this workload (sdrk_plan_tune_scratch; the probe times are reported)."""
    in_samples = (n_frames - 1) * stride + nfft
    gen_frames = (in_samples + 4095) // 4096
    d_gen, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    _ffi.check(lib.sdrk_dev_alloc(dev, gen_frames * 4096 * 8, ctypes.byref(d_gen)))
    try:
        _ffi.check(lib.sdrk_dev_alloc(dev, n_frames * nfft * 4, ctypes.byref(d_out)))
        try:
            _ffi.check(lib.sdrk_synth_fill(dev, 99, 0, gen_frames, 4096, d_gen, None))
            with SpectrumPlan(nfft, window=window, device=dev) as plan:
                try:
                    plan.exec_device(d_gen.value, n_frames, d_out.value, frame_stride=stride)
                    plan.sync()
                except _ffi.SdrkError:
                    # nfft = 65536: a persistent launch that could not get all its workgroups resident (a neighbour on the device)
                    # reports a failed hand-over, and the plan takes the two tiled launches from then on — the leg goes on with those
                    if not plan.fused_status()["fallen_back"]:
                        raise
                    plan.exec_device(d_gen.value, n_frames, d_out.value, frame_stride=stride)
                    plan.sync()
                probe, chosen, report = [], 0, None
                if scratch_candidates > 1:
                    probe, chosen = plan.tune_scratch(d_gen.value, n_frames, d_out.value, scratch_candidates, frame_stride=stride)
                    report = plan.last_placement
                warm_up_by_time(lambda: plan.exec_device_timed(d_gen.value, n_frames, d_out.value, 1, frame_stride=stride))
                ms = plan.exec_device_timed_each(d_gen.value, n_frames, d_out.value, reps, frame_stride=stride)
                fused = plan.fused_status()
                took_fused = bool(fused["launches"]) and not fused["fallen_back"]
            other = None
            if nfft == 65536:
                # N = 65536 has two forms with bit-identical rows; the default plan above took one of them —
                # the other one timed the same way beside it, and the rows of both compared
                import numpy as np
                picks = sorted({0, n_frames // 3, n_frames - 1})
                keep = np.empty((len(picks), nfft), np.float32)
                for i, r in enumerate(picks):
                    _ffi.check(lib.sdrk_memcpy_d2h(dev, keep[i].ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_out.value + r * nfft * 4), nfft * 4))
                with SpectrumPlan(nfft, window=window, device=dev, fused64k=not took_fused) as alt:
                    warm_up_by_time(lambda: alt.exec_device_timed(d_gen.value, n_frames, d_out.value, 1, frame_stride=stride))
                    ms_alt = alt.exec_device_timed_each(d_gen.value, n_frames, d_out.value, reps, frame_stride=stride)
                    alt.sync()
                again = np.empty_like(keep)
                for i, r in enumerate(picks):
                    _ffi.check(lib.sdrk_memcpy_d2h(dev, again[i].ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_out.value + r * nfft * 4), nfft * 4))
                other = {"form": "two tiled launches (fft_tiled2.hip)" if took_fused else "one persistent launch (fft_fused64k.hip)",
                         "ms": round(_median(ms_alt), 3), "ms_min": round(min(ms_alt), 3), "ms_max": round(max(ms_alt), 3),
                         "rows_identical_to_the_default_form": bool(np.array_equal(keep, again)), "rows_compared": picks}
        finally:
            lib.sdrk_dev_free(dev, d_out)
    finally:
        lib.sdrk_dev_free(dev, d_gen)
    t = _median(ms) * 1e-3
    algo = 8 * in_samples + 4 * n_frames * nfft        # unique input once + rows once
    return {"nfft": nfft, "frames": n_frames, "hop": stride, "window": window or "rect",
            "ms": round(t * 1e3, 3), "ms_min": round(min(ms), 3), "ms_max": round(max(ms), 3),
            "input_Msamples_per_s": round(in_samples / t / 1e6, 1),
            "frame_Msamples_per_s": round(n_frames * nfft / t / 1e6, 1),
            "algorithmic_bytes": algo, "algorithmic_formula": "8*L + 4*rows*N",
            "GBps": round(algo / t / 1e9, 1), "frac": round(algo / t / 1e9 / HBM_PEAK_GBPS, 4),
            **({} if nfft != 65536 else {
                "form": "one persistent launch, intermediate in each XCD's L2 (fft_fused64k.hip)" if took_fused
                        else "two tiled launches (fft_tiled2.hip)",
                "persistent_launches": fused["launches"], "fell_back_to_tiled": fused["fallen_back"], "other_form": other}),
            "warmup": "the plan's own transform for >= 100 ms before the timed launches (an idle device needs tens of ms of load "
                      "to reach its sustained shader clock: tools/cfg_steady.py)",
            "scratch_placement": None if report is None else {
                "probe_ms": [round(v, 3) for v in probe], "chosen": chosen, **report,
                "what": "sdrk_plan_tune_scratch: warm-up by time, the plan's transform timed with its own scratch (index 0) and with "
                        "freshly allocated ones, index 0 timed AGAIN after the last; a candidate replaces the present scratch only "
                        "if it beats both timings of it by 1 %; gain_vs_retimed_first = 1 - chosen / re-timed first"}}


def channel_config5(lib, _ffi, pkg, SpectrumPlan, dev, n_frames=256, batch=24, px=4096):
    """BASELINE.json configs[4] as SURVEY.md §8(d) words it, one channel on this GPU: back-to-back N = 2^20 Hann
    frames from a device-resident stream, every row appended to the device waterfall ring (100 x 4 MiB), and after
    each batch of rows a decimated (max-hold to `px` bins) read-out of the new rows to the host.  `batch` = the 24 frames of
    one scratch chunk (192 MiB / 8 MiB), so a batch is one col-pass and one row-pass launch (two where the ring wraps); the row
    pass leaves every row max-hold-decimated by 16 beside the ring, and the read-out reduces those (4 MiB per 16 rows) instead
    of reading the 4 MiB rows again."""
    import numpy as np
    nfft = 1 << 20
    d_in = ctypes.c_void_p()
    _ffi.check(lib.sdrk_dev_alloc(dev, n_frames * nfft * 8, ctypes.byref(d_in)))
    wf = pkg.WaterfallBuffer(nfft, maxlen=100, device=dev, window="hann")
    try:
        _ffi.check(lib.sdrk_synth_fill(dev, 1234 + dev, 0, n_frames * nfft // 4096, 4096, d_in, None))
        with SpectrumPlan(nfft, window="hann", device=dev) as plan:
            def run():
                wf.clear()
                got = 0
                for f0 in range(0, n_frames, batch):
                    nb = min(batch, n_frames - f0)
                    _ffi.check(lib.sdrk_waterfall_append_iq_device(wf._h(), plan.handle,
                                                                  ctypes.c_void_p(d_in.value + f0 * nfft * 8),
                                                                  ctypes.c_size_t(nb), ctypes.c_size_t(nfft)))
                    got += wf.as_array(max_rows=nb, decimate=nfft // px).shape[0]
                return got

            slots = [pkg.pinned_empty((batch, px), np.float32) for _ in range(3)]

            def run_pipelined():
                # the same work with nothing waited for that does not have to be: batch i's transform and its read-out (on the
                # ring's second stream) are enqueued BEFORE batch i - 1's rows are collected — two read-outs in flight — so the
                # host's wait for rows that are one batch old never leaves the transform stream without the next batch
                wf.clear()
                got, in_flight = 0, 0
                for i, f0 in enumerate(range(0, n_frames, batch)):
                    nb = min(batch, n_frames - f0)
                    wf.append_iq_device(d_in.value + f0 * nfft * 8, nb, nfft, wait=False)
                    wf.gather_begin(max_rows=nb, decimate=nfft // px, out=slots[i % 3])
                    in_flight += 1
                    if in_flight == 2:
                        got += wf.gather_end().shape[0]
                        in_flight -= 1
                while in_flight:
                    got += wf.gather_end().shape[0]
                    in_flight -= 1
                return got

            warm_up_by_time(run)
            ts = []
            for _ in range(7):
                t0 = time.perf_counter()
                rows = run()
                ts.append(time.perf_counter() - t0)
            warm_up_by_time(run_pipelined, ms=30.0)
            tp = []
            for _ in range(7):
                t0 = time.perf_counter()
                rows_p = run_pipelined()
                tp.append(time.perf_counter() - t0)
            companions = wf.maxhold16_rows()
            # the ring's newest row bit for bit what the plain transform gives, and the decimated rows what numpy makes of the ring's
            check_rows = wf.as_array(max_rows=2)
            d_chk = ctypes.c_void_p()
            _ffi.check(lib.sdrk_dev_alloc(dev, 2 * nfft * 4, ctypes.byref(d_chk)))
            try:
                plan.exec_device(d_in.value + (n_frames - 2) * nfft * 8, 2, d_chk.value)
                plan.sync()
                plain = np.empty((2, nfft), dtype=np.float32)
                _ffi.check(lib.sdrk_memcpy_d2h(dev, plain.ctypes.data_as(ctypes.c_void_p), d_chk, plain.nbytes))
            finally:
                lib.sdrk_dev_free(dev, d_chk)
            ring_equals_transform = bool(np.array_equal(check_rows, plain))
            decimated_equals_numpy = bool(np.array_equal(wf.as_array(max_rows=2, decimate=nfft // px),
                                                         check_rows.reshape(2, px, nfft // px).max(-1)))
    finally:
        wf.close()
        lib.sdrk_dev_free(dev, d_in)
    t = _median(ts)
    rate = n_frames * nfft / t / 1e6
    return {"nfft": nfft, "frames": n_frames, "window": "hann", "ring_rows": 100, "gather": f"every {batch} rows, max-hold to {px} bins, D2H",
            "ring_rows_with_maxhold16_companion": companions, "ring_rows_equal_plain_transform": ring_equals_transform,
            "decimated_rows_equal_numpy_max_of_ring_rows": decimated_equals_numpy,
            "rows_gathered": rows, "ms": round(t * 1e3, 3), "ms_min": round(min(ts) * 1e3, 3), "ms_max": round(max(ts) * 1e3, 3),
            "Msamples_per_s": round(rate, 1), "realtime_factor_at_61.44_Msps": round(rate / 61.44, 1),
            "realtime_61.44_Msps_holds": bool(rate >= 61.44),
            "what": "transform -> device waterfall ring -> decimated host gather, host-timed (perf_counter) including every sync",
            "pipelined": {"ms": round(_median(tp) * 1e3, 3), "ms_min": round(min(tp) * 1e3, 3), "ms_max": round(max(tp) * 1e3, 3),
                          "rows_gathered": rows_p, "Msamples_per_s": round(n_frames * nfft / _median(tp) / 1e6, 1),
                          "realtime_factor_at_61.44_Msps": round(n_frames * nfft / _median(tp) / 1e6 / 61.44, 1),
                          "what": "append_iq_device(wait=False) + gather_begin / gather_end into pinned slots, two read-outs in flight: batch i+1's "
                                  "transform is enqueued before batch i's rows are collected"}}


def numpy_boundary(lib, _ffi, pkg, synth, dev):
    """spectrum_db(host array) -> host array, PCIe and staging inclusive (never `value`)."""
    import numpy as np
    a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    _ffi.check(lib.sdrk_host_link_probe(dev, 1 << 30, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
    out = {"what": "pkg.spectrum_db(complex64 numpy (B,4096)) -> float32 numpy, pageable arrays, N=4096 rect",
           "link_probe_GBps": {"h2d": round(a.value, 1), "d2h": round(b.value, 1), "duplex_upstream": round(c.value, 1)},
           "host_helper_threads": int(lib.sdrk_host_threads()), "by_batch": {}}
    for bsz in (1, 16, 256, 4096, 32768, 65536):
        if bsz <= 32768:
            x = synth.synth_iq(1, 0, bsz, NFFT)
        else:                                                   # SURVEY.md 8(d)'s largest batch: the 32768 frames twice
            x = np.concatenate([x, x])                          # (the generator in numpy would take longer than the leg)
        one = x[0] if bsz == 1 else x
        pkg.spectrum_db(one, device=dev)                        # plan + staging warm-up
        reps = 300 if bsz <= 16 else (9 if bsz <= 4096 else 3)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            y = pkg.spectrum_db(one, device=dev)
            ts.append(time.perf_counter() - t0)
            del y                                               # the free is the caller's cost, not the call's
        rec = {"ms_per_call": round(_median(ts) * 1e3, 4),
               "Msamples_per_s": round(bsz * NFFT / _median(ts) / 1e6, 1),
               "input_GBps": round(bsz * NFFT * 8 / _median(ts) / 1e9, 2)}
        if bsz >= 256:                                          # result array reused by the caller (out=)
            res = np.empty((bsz, NFFT), dtype=np.float32)
            pkg.spectrum_db(x, device=dev, out=res)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                pkg.spectrum_db(x, device=dev, out=res)
                ts.append(time.perf_counter() - t0)
            rec["reused_out_ms_per_call"] = round(_median(ts) * 1e3, 4)
            rec["reused_out_input_GBps"] = round(bsz * NFFT * 8 / _median(ts) / 1e9, 2)
            rec["reused_out_frac_of_duplex_link"] = round(bsz * NFFT * 8 / _median(ts) / 1e9 / c.value, 3)
        if bsz >= 256:                                          # input and result in pinned arrays: no staging copies
            xp, rp = pkg.pinned_empty(x.shape, np.complex64), pkg.pinned_empty(x.shape, np.float32)
            xp[...] = x
            pkg.spectrum_db(xp, device=dev, out=rp)
            ts, cpu = [], []
            for _ in range(reps):
                c0, t0 = time.process_time(), time.perf_counter()
                pkg.spectrum_db(xp, device=dev, out=rp)
                ts.append(time.perf_counter() - t0)
                cpu.append(time.process_time() - c0)
            gib = bsz * NFFT * 8 / 2 ** 30
            rec["pinned_ms_per_call"] = round(_median(ts) * 1e3, 4)
            rec["pinned_input_GBps"] = round(bsz * NFFT * 8 / _median(ts) / 1e9, 2)
            rec["pinned_frac_of_duplex_link"] = round(bsz * NFFT * 8 / _median(ts) / 1e9 / c.value, 3)
            rec["pinned_host_cpu_ms_per_GiB"] = round(_median(cpu) * 1e3 / gib, 2)
            cpu = []
            for _ in range(reps):                               # the same for the staged pageable path (result reused)
                c0 = time.process_time()
                pkg.spectrum_db(x, device=dev, out=res)
                cpu.append(time.process_time() - c0)
            rec["pageable_host_cpu_ms_per_GiB"] = round(_median(cpu) * 1e3 / gib, 2)
            del xp, rp
        if bsz == 32768:
            # library-side placement of the plan's own staging (SDRK_PLAN_TUNE_STAGING) against a plain plan, same
            # arrays, result reused: does the pairing effect of the resident buffers reach the numpy boundary?
            from sdr_iq_visualizer_amd.spectrum import SpectrumPlan
            ab = {}
            for name, kw in (("plain_plan", {}), ("tuned_staging", {"tune_staging": True})):
                with SpectrumPlan(NFFT, device=dev, **kw) as tp:
                    tp.spectrum_db(x, out=res)
                    ts = []
                    for _ in range(5):
                        t0 = time.perf_counter()
                        tp.spectrum_db(x, out=res)
                        ts.append(time.perf_counter() - t0)
                    ab[name] = {"ms_per_call": round(_median(ts) * 1e3, 3), "ms_min": round(min(ts) * 1e3, 3)}
                    if kw:
                        ab[name]["probe_ms_3_candidates_per_slot"] = [round(v, 4) for v in tp.staging_probe()]
            rec["staging_placement"] = ab
        out["by_batch"][f"B{bsz}"] = rec
    out["pinned_what"] = ("input and result in pkg.pinned_empty arrays (sdrk_host_alloc): chunks are DMA'd straight from / to the "
                          "caller's memory; host_cpu = process CPU time (all threads) per GiB of input")
    return out


def feature_reductions(lib, _ffi, SpectrumPlan, features, dev, n_frames=1 << 18):
    """classifier.py:163-212 next to the rows, device resident, N = 4096 Hann: (a) the fused
    form — reductions as the epilogue of the transform, rows never written; (b) the transform followed by the
    stand-alone single-read reduction kernel over its rows."""
    n, max_peaks = NFFT, 64
    rank, gamma = features.percentile_rank(n, 20.0), float(features.percentile_gamma(n, 20.0))
    bufs = {}

    def alloc(name, nbytes):
        bufs[name] = ctypes.c_void_p()
        _ffi.check(lib.sdrk_dev_alloc(dev, nbytes, ctypes.byref(bufs[name])))

    try:
        alloc("iq", n_frames * n * 8); alloc("rows", n_frames * n * 4); alloc("stats", n_frames * 16 * 8)
        alloc("thr", n_frames * 8); alloc("idx", n_frames * max_peaks * 4); alloc("cnt", n_frames * 4)
        _ffi.check(lib.sdrk_synth_fill(dev, 4321, 0, n_frames, n, bufs["iq"], None))
        out = {"nfft": n, "frames": n_frames, "window": "hann", "max_peaks": max_peaks}
        with SpectrumPlan(n, window="hann", device=dev) as plan:
            def fused(rows_ptr):
                _ffi.check(lib.sdrk_frame_features_device(plan.handle, bufs["iq"], n_frames, n, rows_ptr, rank,
                                                          ctypes.c_float(gamma), max(3, n // 300), max_peaks, bufs["stats"],
                                                          bufs["thr"], bufs["idx"], bufs["cnt"], None))
                plan.sync()

            def timed(fn):
                warm_up_by_time(fn)
                ts = []
                for _ in range(7):
                    t0 = time.perf_counter()
                    fn()
                    ts.append(time.perf_counter() - t0)
                return _median(ts)

            t = timed(lambda: fused(None))
            out["fused_rows_not_written"] = {"ms": round(t * 1e3, 3), "rows_per_s": round(n_frames / t),
                                             "Msamples_per_s": round(n_frames * n / t / 1e6, 1),
                                             "algorithmic_bytes": 8 * n_frames * n, "algorithmic_formula": "8*frames*N (IQ read only)",
                                             "GBps": round(8 * n_frames * n / t / 1e9, 1),
                                             "frac": round(8 * n_frames * n / t / 1e9 / HBM_PEAK_GBPS, 4)}
            t = timed(lambda: fused(bufs["rows"]))
            out["fused_rows_also_written"] = {"ms": round(t * 1e3, 3), "rows_per_s": round(n_frames / t),
                                              "GBps": round(12 * n_frames * n / t / 1e9, 1)}

            import numpy as np
            host = {"stats": np.empty((n_frames, 16), dtype=np.float64), "thr": np.empty(n_frames, dtype=np.float64),
                    "idx": np.empty((n_frames, max_peaks), dtype=np.int32), "cnt": np.empty(n_frames, dtype=np.int32)}
            hp = {k: v.ctypes.data_as(ctypes.c_void_p) for k, v in host.items()}

            def separate():
                plan.exec_device(bufs["iq"].value, n_frames, bufs["rows"].value)
                plan.sync()
                _ffi.check(lib.sdrk_row_features(dev, bufs["rows"], 1, n_frames, n, rank, ctypes.c_float(gamma),
                                                 max(3, n // 300), max_peaks, hp["stats"], hp["thr"], hp["idx"], hp["cnt"]))

            t = timed(separate)
            # at-scale consistency: the fused epilogue and the stand-alone kernel run the same routine on the same row
            # values, so their results must agree on every one of the rows (statistics, threshold, peak list)
            fused(bufs["rows"])
            chk = {"stats": np.empty_like(host["stats"]), "thr": np.empty_like(host["thr"]), "idx": np.empty_like(host["idx"]),
                   "cnt": np.empty_like(host["cnt"])}
            for k in chk:
                _ffi.check(lib.sdrk_memcpy_d2h(dev, chk[k].ctypes.data_as(ctypes.c_void_p), bufs[k], chk[k].nbytes))
            kept = np.minimum(host["cnt"], max_peaks)[:, None] > np.arange(max_peaks)[None, :]
            out["fused_equals_stand_alone_on_all_rows"] = bool(
                np.array_equal(chk["stats"], host["stats"]) and np.array_equal(chk["thr"], host["thr"]) and
                np.array_equal(chk["cnt"], host["cnt"]) and np.array_equal(chk["idx"][kept], host["idx"][kept]))
            out["peaks_per_row_mean"] = round(float(host["cnt"].mean()), 1)
            out["transform_then_single_read_reduction"] = {
                "ms": round(t * 1e3, 3), "rows_per_s": round(n_frames / t),
                "what": "fft4096_kernel writes the rows, row_features_kernel reads them once (staged in LDS); "
                        "includes the D2H of the per-row results (stats, threshold, 64 peak slots)"}
        # the numpy boundary of the same measurement (PCIe inclusive, never `value`): host IQ in, arrays of results out
        import sdr_iq_visualizer_amd as pkg
        b = 1 << 15                                                   # 1 GiB of IQ
        x = pkg.pinned_empty((b, n), np.complex64)
        x[...] = (np.random.default_rng(5).standard_normal((b, 2 * n), dtype=np.float32) * 200).view(np.complex64)
        pageable = np.array(x)
        rec = {"frames": b, "max_peaks": max_peaks,
               "what": "features.frame_features(complex64 numpy (B,4096), as_arrays=True): chunks staged through pinned slots "
                       "(pageable) or DMA'd from the caller's pinned array, fused kernel per chunk, results back at the end"}
        for name, arr in (("pageable", pageable), ("pinned", x)):
            features.frame_features(arr[:2048], 1e6, 0.0, window="hann", device=dev, max_peaks=max_peaks, as_arrays=True)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                features.frame_features(arr, 1e6, 0.0, window="hann", device=dev, max_peaks=max_peaks, as_arrays=True)
                ts.append(time.perf_counter() - t0)
            t = _median(ts)
            rec[name] = {"ms_per_call": round(t * 1e3, 2), "rows_per_s": round(b / t), "input_GBps": round(b * n * 8 / t / 1e9, 2)}
        out["numpy_boundary"] = rec
        return out
    finally:
        for b in bufs.values():
            lib.sdrk_dev_free(dev, b)


def negotiate_backend(dist, world, rank, gloo_group, stages):
    """The nccl-or-gloo choice as a COLLECTIVE decision (every rank calls this with the same number of stages).
    Each stage is a callable that raises on failure; after each stage the ranks vote over gloo (which is already up).
    All ranks fine after the last stage -> ("nccl", whatever the last stage returned, None).  No rank fine at some
    stage -> ("gloo", gloo_group, note): the data path has no collective, so the barrier and the reductions of the
    times can run over gloo and the line says so.  Some fine, some not -> every rank exits non-zero with the reason:
    no rank is ever left inside an nccl collective that others have abandoned."""
    import torch
    result = None
    for i, stage in enumerate(stages):
        try:
            result, err = stage(), None
        except Exception as e:                                   # noqa: BLE001 - voted on below
            result, err = None, f"{type(e).__name__}: {str(e)[:160]}"
        votes = torch.tensor([0 if err else 1, 1 if err else 0], dtype=torch.int64)
        dist.all_reduce(votes, group=gloo_group)
        n_ok, n_bad = int(votes[0]), int(votes[1])
        if n_bad == 0:
            continue
        reasons = [None] * world
        dist.all_gather_object(reasons, err, group=gloo_group)
        bad = [r for r, why in enumerate(reasons) if why]
        first = reasons[bad[0]]
        if n_ok == 0:
            return "gloo", gloo_group, f"nccl did not come up on any rank (stage {i}: {first}); barrier and reductions on gloo"
        raise SystemExit(f"[bench] rank {rank}: nccl came up on {n_ok} of {world} ranks only (stage {i}, failed on ranks "
                         f"{bad}: {first}); no consistent rendezvous backend, every rank gives up")
    return "nccl", result, None


def _injected_nccl_failure(rank):
    """Rehearsal hook (tools / tests only): SDRK_BENCH_FAIL_NCCL=all (or 1) fails every rank, =rank:K only rank K."""
    spec = os.environ.get("SDRK_BENCH_FAIL_NCCL")
    if not spec:
        return False
    if spec.startswith("rank:"):
        return rank == int(spec.split(":", 1)[1])
    return True


def child_command(args, port):
    """The one-rank-per-GPU launch of this file (what the driver itself runs for N > 1)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--frames", str(args.frames), "--first-frame", str(args.first_frame), "--window", args.window,
           "--parity-frames", str(args.parity_frames),
           "--placement-candidates", str(args.placement_candidates), "--cpu-seconds", "0"]
    if args.no_secondary:
        cmd.append("--no-secondary")
    return cmd


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) with no outer launcher: this process never touches a GPU.  It times the
    CPU baseline (pure numpy), hands it to rank 0 through a file named in the environment, starts the ranks as a
    CHILD process (never an exec: nothing that has initialised a GPU is ever replaced) and exits with its code;
    the child's rank 0 prints the one JSON line on the stdout both share."""
    import socket
    import subprocess
    import tempfile
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    tmp = None
    if args.cpu_seconds > 0:
        cpu = cpu_baseline(args.window, args.cpu_seconds)
        if args.cpu_all_cores_seconds > 0:
            cpu["all_cores"] = cpu_baseline_all_cores(args.window, args.cpu_all_cores_seconds)
        tmp = tempfile.NamedTemporaryFile("w", suffix=".json", prefix="sdrk_cpu_baseline_", delete=False)
        json.dump(cpu, tmp)
        tmp.close()
        env["SDRK_BENCH_CPU_BASELINE"] = tmp.name
    with socket.socket() as sock:                    # a free rendezvous port on the loopback
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    try:
        rc = subprocess.call(child_command(args, port), env=env)
    finally:
        if tmp is not None:
            try:
                os.unlink(tmp.name)
            except OSError:
                pass
    sys.exit(rc)


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "RANK" not in os.environ and args.gpus > 1:
            self_launch(args)                        # does not return
        args.gpus = world
    # ONE JSON line on stdout: libraries print to file descriptor 1 behind Python's back (gloo's "[Gloo] Rank 0 is
    # connected ..." and RCCL's version banner do), so descriptor 1 points at stderr until the line itself is printed
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    solo = rank == 0 and world == 1

    # cpu baseline first: plain numpy, before any GPU runtime is up.  N = 1: this process; N > 1 self-launched:
    # the parent timed it before the ranks existed and left it in a file; N > 1 under an outer launcher: rank 0
    # times the single-core leg while the other ranks wait at the rendezvous (they hold no CPU meanwhile).
    cpu = None
    handed = os.environ.get("SDRK_BENCH_CPU_BASELINE")
    if rank == 0 and handed and os.path.exists(handed):
        try:
            cpu = json.load(open(handed))
        except Exception:
            cpu = None
    elif rank == 0 and args.cpu_seconds > 0:
        cpu = cpu_baseline(args.window, args.cpu_seconds if solo else min(args.cpu_seconds, 5.0))
        if solo and args.cpu_all_cores_seconds > 0:
            cpu["all_cores"] = cpu_baseline_all_cores(args.window, args.cpu_all_cores_seconds)
    cpu_large = None
    if solo and not args.no_secondary and args.cpu_seconds > 0 and args.cpu_large_seconds > 0:
        cpu_large = cpu_baselines_secondary(args.cpu_large_seconds, args.cpu_large_seconds)

    import torch  # before libsdrk: one shared HIP runtime in the process (see _ffi.py)
    import numpy as np
    from oracle import cpu_ref
    import sdr_iq_visualizer_amd as pkg
    from sdr_iq_visualizer_amd import _ffi, synth
    from sdr_iq_visualizer_amd.spectrum import SpectrumPlan

    dist = None
    n_dev = torch.cuda.device_count()
    dev = local_rank % max(n_dev, 1)
    # One rank per GPU over RCCL ("nccl").  Rehearsal on a box with fewer GPUs than ranks
    # (ranks then share a device, which RCCL refuses): gloo carries the barrier/max-reduce.
    backend = "nccl" if n_dev >= world else "gloo"
    backend_note, grp = None, None
    if n_dev < world:
        args.placement_candidates = 1               # ranks share a GPU here: no probing with its memory
        backend_note = (f"{world} ranks on {n_dev} visible GPU(s): RCCL does not form a communicator over ranks that share a "
                        "device, so the barrier and the reductions of the times run over gloo (a rehearsal of the control path, "
                        "not a scaling measurement)")
    if "RANK" in os.environ:
        import datetime
        import torch.distributed as dist
        torch.cuda.set_device(dev)
        # gloo first: it always comes up, carries the votes below, and is the fallback for the barrier and the
        # reductions of the times (the data path itself has no collective)
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=900))
        grp = dist.group.WORLD
        if backend == "nccl":
            state = {}

            def willing():                          # local preconditions, voted on BEFORE anybody enters new_group
                if _injected_nccl_failure(rank):    # (which ends in a barrier over all ranks)
                    raise RuntimeError("SDRK_BENCH_FAIL_NCCL is set")
                if not torch.cuda.is_available():
                    raise RuntimeError("no GPU visible to this rank")

            def form():
                state["g"] = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=180))
                return state["g"]

            def prove():                            # the communicator is formed and proven here, before the warm-up
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe, group=state["g"])
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError(f"all-reduce of ones over {world} ranks gave {probe.item()}")
                return state["g"]

            backend, grp, backend_note = negotiate_backend(dist, world, rank, dist.group.WORLD, [willing, form, prove])
    red_dev = "cuda" if backend == "nccl" else "cpu"
    lib = _ffi.lib()
    _ffi.require_device(dev)
    info = _ffi.device_info(dev)
    tel = Telemetry(info) if rank == 0 else None

    frames = args.frames
    first_frame = args.first_frame + rank * frames  # config 4: GPU g owns [g*F, (g+1)*F)
    plan = SpectrumPlan(NFFT, window=None if args.window == "rect" else args.window, device=dev)

    # the resident IQ / row buffers, the rows placed by sdrk_dev_alloc_stream_pair (the streaming rate of a
    # read+write pair depends on which two allocations are paired: DESIGN.md §4.1); the probe is the plan's own
    # transform, run before the warm-up on the not yet generated input (its values do not matter)
    d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    probe_ms, chosen = (ctypes.c_float * args.placement_candidates)(), ctypes.c_int(0)
    _ffi.check(lib.sdrk_dev_alloc_stream_pair(dev, frames * NFFT * 8, frames * NFFT * 4, args.placement_candidates,
                                              plan.handle, ctypes.byref(d_in), ctypes.byref(d_out), probe_ms,
                                              ctypes.byref(chosen)))
    from sdr_iq_visualizer_amd.spectrum import placement_report
    placement = {"candidates": args.placement_candidates, "probe_ms": [round(float(v), 4) for v in probe_ms],
                 "chosen": int(chosen.value), **placement_report(),
                 "what": "output buffer chosen among candidates by timing the plan's transform over each pairing with "
                         "the input buffer (sdrk_dev_alloc_stream_pair: warm-up by time, candidate 0 timed again after the "
                         "last one — retimed_first_ms — and counted with the better of its two timings); 1 candidate = plain allocation"}
    _ffi.check(lib.sdrk_synth_fill(dev, 1234, first_frame, frames, NFFT, d_in, None))

    def barrier():
        plan.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(group=grp, **({"device_ids": [dev]} if backend == "nccl" else {}))

    for _ in range(args.warmup):
        plan.exec_device(d_in.value, frames, d_out.value)
    barrier()
    before = tel.snapshot() if tel else None
    if tel:
        tel.start()
    t0 = time.perf_counter()
    # K steps; HIP events on the stream the kernel runs on, one between every two launches
    each_ms = plan.exec_device_timed_each(d_in.value, frames, d_out.value, launches=args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if tel:
        tel.stop()
    after = tel.snapshot() if tel else None
    kernel_ms = float(sum(each_ms))
    per_rank = None
    if dist is not None:
        # every rank's own median launch and wall time, so that a straggler (placement level, clocks) shows in the line
        # ... beside the placement level each rank's buffer pair was probed at (which pairing a process gets is a
        # lottery; a rank whose launches are slow by what its probe already showed drew a slow pair, it is not a scaling loss)
        probed = [float(v) for v in probe_ms if float(v) > 0]
        mine = torch.tensor([_median(list(each_ms)), elapsed * 1e3 / args.steps,
                             float(probe_ms[int(chosen.value)]) if probed else 0.0, min(probed) if probed else 0.0,
                             max(probed) if probed else 0.0, float(len(probed))], device=red_dev, dtype=torch.float64)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine, group=grp)
        per_rank = {"launch_ms_median": [round(float(v[0]), 4) for v in everyone],
                    "wall_ms_per_step": [round(float(v[1]), 4) for v in everyone],
                    "placement_probe_ms": {"chosen": [round(float(v[2]), 4) for v in everyone],
                                           "fastest_candidate": [round(float(v[3]), 4) for v in everyone],
                                           "slowest_candidate": [round(float(v[4]), 4) for v in everyone],
                                           "candidates_timed": [int(v[5]) for v in everyone],
                                           "frames_per_probe": "one launch over the rank's own frames (0 = plain allocation, nothing probed)"}}
        t = torch.tensor([elapsed, kernel_ms], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=grp)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    # parity on this rank's output of the timed launches (oracle = checker only)
    parity, n_checked = 0.0, 0
    if args.parity_frames > 0:
        rng = np.random.default_rng(rank)
        picks = np.unique(np.concatenate([[0, frames - 1], rng.integers(0, frames, args.parity_frames)]))
        w = np.hanning(NFFT) if args.window == "hann" else None
        row = np.empty(NFFT, dtype=np.float32)
        for f in picks:
            _ffi.check(lib.sdrk_memcpy_d2h(dev, row.ctypes.data_as(ctypes.c_void_p),
                                           ctypes.c_void_p(d_out.value + int(f) * NFFT * 4), row.nbytes))
            ref = cpu_ref.spectrum_db(synth.synth_iq(1234, first_frame + int(f), 1, NFFT)[0], window=w)
            mg, mr = 10.0 ** (row.astype(np.float64) / 20), 10.0 ** (ref.astype(np.float64) / 20)
            parity = max(parity, float(np.abs(mg - mr).max() / mr.max()))
        n_checked = int(picks.size)
    if dist is not None:
        mine = torch.tensor([parity, float(first_frame), float(first_frame + frames)], device=red_dev, dtype=torch.float64)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine, group=grp)
        per_rank["parity_max_rel_err"] = [float(v[0]) for v in everyone]
        per_rank["frame_range"] = [[int(v[1]), int(v[2])] for v in everyone]     # rank g: [first + g F, first + (g + 1) F)
        per_rank["first_sample_index"] = [int(v[1]) * NFFT for v in everyone]     # >= 2^32 from config 4's rank 1 on: 64-bit frame numbers in the generator
        parity = max(per_rank["parity_max_rel_err"])

    # the same bytes with no arithmetic, same buffers, same process: the measured-copy ceiling (rank 0)
    copy_gbps = None
    if rank == 0:
        try:
            ms = (ctypes.c_float * 10)()
            _ffi.check(lib.sdrk_stream_ceiling_probe(dev, d_in, d_out, frames, 10, ms))
            copy_gbps = ALGO_BYTES_PER_SAMPLE * frames * NFFT / (_median(list(ms)) * 1e-3) / 1e9
        except Exception as e:                                   # a probe must never cost the bench line
            copy_gbps = None
            print(f"[bench] stream ceiling probe failed: {e}", file=sys.stderr)

    # ... and the guide's reference shape, a 1:1 float4 copy (MI355X_MICROARCH.md: 6.29 TB/s), between the same
    # two buffers: the first 16 GiB of the IQ buffer into the row buffer
    copy11_gbps = None
    if rank == 0:
        try:
            ms = (ctypes.c_float * 10)()
            nbytes = frames * NFFT * 4
            _ffi.check(lib.sdrk_copy_probe(dev, d_in, d_out, nbytes, 10, ms))
            copy11_gbps = 2 * nbytes / (_median(list(ms)) * 1e-3) / 1e9
        except Exception as e:
            print(f"[bench] 1:1 copy probe failed: {e}", file=sys.stderr)

    # ... and the same 1:1 copy on a small pair (4 + 2 GiB) allocated beside the resident one, same process: the footprint
    # dependence of the copy ceiling on record (the guide's 6.29 TB/s is a small-footprint figure)
    copy11_small = None
    if solo and not args.no_secondary and frames * NFFT * 8 >= (8 << 30):
        try:
            a_s, b_s = ctypes.c_void_p(), ctypes.c_void_p()
            _ffi.check(lib.sdrk_dev_alloc(dev, 4 << 30, ctypes.byref(a_s)))
            try:
                _ffi.check(lib.sdrk_dev_alloc(dev, 2 << 30, ctypes.byref(b_s)))
                try:
                    ms = (ctypes.c_float * 20)()
                    _ffi.check(lib.sdrk_copy_probe(dev, a_s, b_s, 2 << 30, 20, ms))
                    copy11_small = {"GBps": round(2 * (2 << 30) / (_median(list(ms)) * 1e-3) / 1e9, 1),
                                    "what": "sdrk_copy_probe on a fresh 4 GiB + 2 GiB pair (2 GiB copied), same process, the "
                                            "32 + 16 GiB pair still allocated"}
                finally:
                    lib.sdrk_dev_free(dev, b_s)
            finally:
                lib.sdrk_dev_free(dev, a_s)
        except Exception as e:
            print(f"[bench] small-footprint copy probe failed: {e}", file=sys.stderr)

    # the other window of configs[1] on the same buffers (SURVEY.md §8d names Hann and rect); rank 0, N = 1 only
    other_window = None
    if solo and not args.no_secondary:
        try:
            ow = None if args.window != "rect" else "hann"
            with SpectrumPlan(NFFT, window=ow, device=dev) as p2:
                p2.exec_device(d_in.value, frames, d_out.value)
                p2.sync()
                ms2 = p2.exec_device_timed_each(d_in.value, frames, d_out.value, launches=7)
            t2 = _median(ms2) * 1e-3
            other_window = {"window": ow or "rect", "frames": frames, "ms": round(t2 * 1e3, 4), "ms_min": round(min(ms2), 4),
                            "ms_max": round(max(ms2), 4), "Msamples_per_s": round(frames * NFFT / t2 / 1e6, 1),
                            "GBps": round(ALGO_BYTES_PER_SAMPLE * frames * NFFT / t2 / 1e9, 1),
                            "frac": round(ALGO_BYTES_PER_SAMPLE * frames * NFFT / t2 / 1e9 / HBM_PEAK_GBPS, 4),
                            "what": "same resident buffers as the timed run, median of 7 launches (HIP events)"}
        except Exception as e:
            other_window = {"error": f"{type(e).__name__}: {e}"}

    plan.close()
    lib.sdrk_dev_free(dev, d_in)
    lib.sdrk_dev_free(dev, d_out)

    secondary = None
    if solo and not args.no_secondary:
        secondary = {"config2_other_window": other_window}
        try:
            L = 614_400_000                                       # 10 s @ 61.44 Msps
            r = device_config(lib, _ffi, SpectrumPlan, dev, 65536, 1 + (L - 65536) // 32768, 32768, "hann", 7)
            r["realtime_factor_at_61.44_Msps"] = round(10.0 / (r["ms"] * 1e-3), 1)
            r["workload"] = "BASELINE.json configs[2]: waterfall STFT N=65536, 50 % overlap, 10 s @ 61.44 Msps"
            r["kernels"] = "sdrk::col_pass_kernel<8,..> + sdrk::row_pass_kernel<8,0> per 384-frame chunk (192 MiB scratch)"
            if cpu_large:
                r["cpu_baseline"] = cpu_large["config3"]
                r["gpu_over_one_core"] = round(r["frame_Msamples_per_s"] / cpu_large["config3"]["value"], 1)
            secondary["config3"] = r
            # (scratch placement: probed here for the record only — warm, config 5 shows no placement effect, its 192 MiB of scratch
            #  lives in the Infinity Cache; the library does not probe unless asked, and what it is asked it reports honestly)
            r = device_config(lib, _ffi, SpectrumPlan, dev, 1 << 20, 256, 1 << 20, "hann", 15, scratch_candidates=4)
            r["workload"] = "BASELINE.json configs[4], one channel: 256 back-to-back N=2^20 frames"
            r["realtime_factor_at_61.44_Msps"] = round((256 * (1 << 20) / 61.44e6) / (r["ms"] * 1e-3), 1)
            if cpu_large:
                r["cpu_baseline"] = cpu_large["config5"]
                r["gpu_over_one_core"] = round(r["frame_Msamples_per_s"] / cpu_large["config5"]["value"], 1)
            secondary["config5_one_channel"] = r
            r = device_config(lib, _ffi, SpectrumPlan, dev, 1 << 20, 256, 1 << 20, None, 15, scratch_candidates=1)
            r["workload"] = "BASELINE.json configs[4], one channel, rectangular window (SURVEY.md 8d: rect and Hann)"
            r["realtime_factor_at_61.44_Msps"] = round((256 * (1 << 20) / 61.44e6) / (r["ms"] * 1e-3), 1)
            secondary["config5_one_channel_rect"] = r
            secondary["config5_channel_with_ring_and_gather"] = channel_config5(lib, _ffi, pkg, SpectrumPlan, dev)
            secondary["numpy_boundary"] = numpy_boundary(lib, _ffi, pkg, synth, dev)
            from sdr_iq_visualizer_amd import features
            secondary["row_features"] = feature_reductions(lib, _ffi, SpectrumPlan, features, dev)
        except Exception as e:
            secondary["error"] = f"{type(e).__name__}: {e}"
    elif dist is not None and not args.no_secondary:
        # N > 1: BASELINE.json configs[4] as it is worded — one continuous N = 2^20 channel PER GPU, all at the same
        # time: 256 back-to-back Hann frames from a resident stream -> the device waterfall ring -> max-hold to 4096
        # bins and a host gather every 16 rows (channel seed = 1234 + device).  Every rank runs its channel between
        # two barriers; rank 0 reports each channel's sustained rate and whether 61.44 Msps real time holds on all.
        rec = None

        def rank_barrier():                                      # (the bench's own plan is closed by now)
            torch.cuda.synchronize()
            dist.barrier(group=grp, **({"device_ids": [dev]} if backend == "nccl" else {}))

        rank_barrier()
        try:
            rec = channel_config5(lib, _ffi, pkg, SpectrumPlan, dev)
        except Exception as e:                                   # noqa: BLE001 - a secondary leg never costs the line
            rec = {"error": f"{type(e).__name__}: {e}"}
        finally:
            rank_barrier()                                       # every rank issues the same sequence of collectives, whatever its leg did
        everyone = [None] * world
        dist.all_gather_object(everyone, rec)                    # the default (gloo) group: plain Python objects
        if rank == 0:
            ok = [r for r in everyone if r and "pipelined" in r]
            secondary = {"config5_channels": {
                "workload": "BASELINE.json configs[4]: one continuous N=2^20 channel per GPU, all concurrently "
                            "(256 frames each, device waterfall ring of 100 rows, decimated host gather every 16 rows)",
                "channels": world,
                "per_channel_Msamples_per_s": [r["pipelined"]["Msamples_per_s"] if r and "pipelined" in r else None for r in everyone],
                "per_channel_ms": [r["pipelined"]["ms"] if r and "pipelined" in r else None for r in everyone],
                "per_channel_ms_with_every_sync": [r["ms"] if r and "ms" in r else None for r in everyone],
                "aggregate_Msamples_per_s": round(sum(r["pipelined"]["Msamples_per_s"] for r in ok), 1),
                "realtime_61.44_Msps_holds_on_every_channel": bool(ok) and len(ok) == world and
                                                              all(r["pipelined"]["Msamples_per_s"] >= 61.44 for r in ok),
                "checks": {k: [bool(r.get(k)) if r else False for r in everyone]
                           for k in ("ring_rows_equal_plain_transform", "decimated_rows_equal_numpy_max_of_ring_rows")},
                "errors": [r["error"] for r in everyone if r and "error" in r] or None}}

    if rank == 0:
        samples_per_step = frames * NFFT * world
        ms_per_step = elapsed * 1e3 / args.steps
        value = samples_per_step / (elapsed / args.steps) / 1e6
        launch_ms = kernel_ms / args.steps
        achieved = ALGO_BYTES_PER_SAMPLE * frames * NFFT / (launch_ms * 1e-3) / 1e9   # per GPU
        traffic, traffic_src = None, None
        # PMC-derived bytes per launch from the newest round's collection (profiles/rNN/hbm_traffic.json, written by
        # tools/install_profiles.py; see profiles/README.md)
        import glob
        rounds = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9]*", "hbm_traffic.json")))
        if rounds:
            try:
                rec = json.load(open(rounds[-1]))
                if rec.get("frames") == frames and rec.get("window") == args.window:
                    traffic = rec.get("bytes_per_launch")
                    traffic_src = (f"{os.path.relpath(rounds[-1], REPO)}: a SEPARATE rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run "
                                   "of this command (not measured by this run)")
            except Exception:
                traffic = None
        line = {
            "metric": METRIC,
            "value": round(value, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"batched N={NFFT} {args.window}-windowed FFT + fftshift + 20*log10(|X|+1e-12), "
                            f"{frames} device-resident complex64 frames per GPU (BASELINE.json configs[1]"
                            + ("; configs[3] frame-range sharding" if world > 1 else "") + ")",
                "nfft": NFFT, "frames_per_gpu": frames, "window": args.window,
                "sharding": f"frame-range x{world}, no collectives", "device": info,
                "rendezvous_backend": (backend if dist is not None else None),
                **({"rendezvous_note": backend_note} if backend_note else {}),
            },
            "hbm_peak_frac": round(value * 1e6 * ALGO_BYTES_PER_SAMPLE / 1e9 / (HBM_PEAK_GBPS * world), 4),
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": "sdrk::fft4096_kernel",
                "kernel_ms_per_launch": round(launch_ms, 4),
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_SAMPLE * frames * NFFT,
                "measured_copy_GBps": None if copy_gbps is None else round(copy_gbps, 1),
                "frac_of_measured_copy": None if not copy_gbps else round(achieved / copy_gbps, 4),
                "measured_copy_what": "sdrk_stream_ceiling_probe: 32 KiB read + 16 KiB written per frame, no arithmetic, "
                                      "same buffers, median of 10 launches after the timed region",
                "copy_1to1_GBps": None if copy11_gbps is None else round(copy11_gbps, 1),
                "frac_of_copy_1to1": None if not copy11_gbps else round(achieved / copy11_gbps, 4),
                "copy_1to1_what": "sdrk_copy_probe: plain 1:1 copy, 16 B per lane each way, first half of the IQ buffer into "
                                  "the row buffer (read + written bytes / time); MI355X_MICROARCH.md quotes 6.29 TB/s for this shape",
                "copy_1to1_small_footprint": copy11_small,
            },
            "launch_ms": {"min": round(min(each_ms), 4), "median": round(_median(each_ms), 4),
                          "max": round(max(each_ms), 4), "mean": round(sum(each_ms) / len(each_ms), 4),
                          "n": len(each_ms), "scope": "rank 0", "series": [round(float(v), 3) for v in each_ms]},
            "parity_max_rel_err": parity,
            "parity_frames_checked": n_checked,
            "placement": placement,
        }
        if per_rank is not None:
            line["per_rank"] = per_rank
            line["aggregate_frac_of_n_gpu_peak"] = round(ALGO_BYTES_PER_SAMPLE * value * 1e6 / 1e9 / (world * HBM_PEAK_GBPS), 4)
        if tel is not None:
            line["telemetry"] = {"before": before, "during": tel.summary(), "after": after,
                                 "source": tel.dir or "no sysfs node found for this device"}
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if secondary is not None:
            line["secondary"] = secondary
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()                               # the default (gloo) group: every rank is done with its GPU
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
