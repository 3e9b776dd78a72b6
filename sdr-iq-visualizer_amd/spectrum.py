"""numpy-in / numpy-out spectrum functions backed by the gfx950 kernels.

This module is the host-side mirror of the reference's hot path.  The reference
has no function boundary there — it is three inline expressions in the SDR
reader thread — so the functions below *are* the boundary a maintainer would
call from those lines:

    app/sdr/streamer.py:119  fft_data = np.fft.fftshift(np.fft.fft(samples))
    app/sdr/streamer.py:120  freqs = np.fft.fftshift(np.fft.fftfreq(len(samples), 1/self.sample_rate)) + self.center_freq
    app/sdr/streamer.py:121  power_db = 20 * np.log10(np.abs(fft_data) + 1e-12)
    app/sdr/streamer.py:123-130  plot_data = {...}

Defaults reproduce the reference exactly: rectangular window, un-normalised
forward DFT, fftshift, additive floor 1e-12 on |X|.  Contract: complex64 in,
float32 out (complex128 input is down-cast; the reference would then compute in
float64 — documented difference, well inside the 1e-5 parity bar).

All arithmetic on samples happens on the GPU through ``libsdrk.so``; nothing in
this module computes a spectrum with numpy, and every entry point raises if the
library or a device is missing.
"""
from __future__ import annotations

import threading
import time
from typing import Optional, Sequence, Union

import numpy as np

from . import _ffi
from ._ffi import byref, c_float, c_int, c_size_t, c_void_p, check, lib

WindowArg = Union[None, str, np.ndarray, Sequence[float]]


def _window_spec(window: WindowArg, nfft: int):
    """-> (kind, float32 array or None, cache key)."""
    if window is None:
        return _ffi.WINDOW_RECT, None, "rect"
    if isinstance(window, str):
        name = window.lower()
        if name in ("rect", "rectangular", "boxcar", "none"):
            return _ffi.WINDOW_RECT, None, "rect"
        if name in ("hann", "hanning"):
            return _ffi.WINDOW_HANN, None, "hann"
        raise ValueError(f"unknown window {window!r} (use None, 'hann' or an array of nfft floats)")
    w = np.ascontiguousarray(np.asarray(window, dtype=np.float32))
    if w.ndim != 1 or w.shape[0] != nfft:
        raise ValueError(f"window must have shape ({nfft},), got {w.shape}")
    return _ffi.WINDOW_CUSTOM, w, ("custom", w.tobytes())


def _as_c64(a) -> np.ndarray:
    """complex64, C-contiguous view or copy of `a` (down-casts complex128)."""
    arr = np.asarray(a)
    if arr.dtype != np.complex64:
        arr = arr.astype(np.complex64)
    return np.ascontiguousarray(arr)


class SpectrumPlan:
    """A compiled plan for one (nfft, window, eps, shift, device) combination.

    Thin owner of an ``sdrk_plan``; methods take and return numpy arrays.  A plan
    is used by one thread at a time (an internal lock enforces it); create one
    plan per device to drive several GPUs from several threads.
    """

    def __init__(self, nfft: int, *, window: WindowArg = None, eps: float = 1e-12,
                 shift: bool = True, device: int = 0, max_batch: int = 1 << 30, fused64k: Optional[bool] = None,
                 overlap_passes: bool = False, tune_staging: bool = False):
        nfft = int(nfft)
        pow2 = nfft >= 2 and not (nfft & (nfft - 1))
        if nfft < 2 or nfft > (1 << _ffi.MAX_LOG2_NFFT) or (not pow2 and nfft > (1 << (_ffi.MAX_LOG2_NFFT - 1))):
            raise ValueError(
                f"nfft={nfft}: frames must have 2..2^{_ffi.MAX_LOG2_NFFT} samples (powers of two) or "
                f"2..2^{_ffi.MAX_LOG2_NFFT - 1} (other lengths, via Bluestein)")
        kind, warr, self._wkey = _window_spec(window, nfft)
        _ffi.require_device(device)
        self.nfft = nfft
        self.eps = float(eps)
        self.shift = bool(shift)
        self.device = int(device)
        self._lock = threading.Lock()
        self.last_placement: Optional[dict] = None     # report of the last tune_scratch() on this plan
        self._handle = c_void_p()
        wptr = warr.ctypes.data_as(c_void_p) if warr is not None else None
        # fused64k (nfft = 65536 only): None = the library's default (the single persistent launch for calls
        # of 512 frames or more, the two tiled launches below that), True = the persistent launch for every call, False = never
        # (sdrk.h SDRK_PLAN_FUSED64K / SDRK_PLAN_TILED64K).  An explicit plan option, so the path is visible in the API.
        self.fused64k = None if fused64k is None else bool(fused64k)
        # overlap_passes: the two passes of a large frame on two streams (slower, kept for A/B)
        self.overlap_passes = bool(overlap_passes)
        # tune_staging: the numpy boundary's device staging placed at creation (sdrk.h SDRK_PLAN_TUNE_STAGING; measured:
        # no effect at the shipped chunk size)
        self.tune_staging = bool(tune_staging)
        flags = ((_ffi.PLAN_FUSED64K if self.fused64k else 0) |
                 (_ffi.PLAN_TILED64K if self.fused64k is False and nfft == 65536 else 0) |
                 (_ffi.PLAN_OVERLAP_PASSES if self.overlap_passes else 0) |
                 (_ffi.PLAN_TUNE_STAGING if self.tune_staging else 0))
        check(lib().sdrk_plan_create_ex(self.device, nfft, c_size_t(int(max_batch)), kind, wptr,
                                        c_float(self.eps), int(self.shift), flags, byref(self._handle)))

    # -- lifetime ---------------------------------------------------------------
    def close(self) -> None:
        h, self._handle = self._handle, c_void_p()
        if h:
            lib().sdrk_plan_destroy(h)

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self) -> c_void_p:
        if not self._handle:
            raise RuntimeError("plan is closed")
        return self._handle

    # -- host arrays ------------------------------------------------------------
    def _run_host(self, fn, iq: np.ndarray, n_frames: int, stride: int, out: np.ndarray) -> None:
        """One C call for the whole batch: libsdrk cuts it into pinned, pipelined chunks itself."""
        with self._lock:
            check(fn(self.handle, iq.ctypes.data_as(c_void_p), c_size_t(n_frames), c_size_t(stride),
                     out.ctypes.data_as(c_void_p)))

    def _frames(self, samples):
        x = _as_c64(samples)
        if x.ndim == 1:
            if x.shape[0] != self.nfft:
                raise ValueError(f"frame has {x.shape[0]} samples, plan nfft is {self.nfft}")
            return x.reshape(1, -1), True
        if x.ndim != 2 or x.shape[1] != self.nfft:
            raise ValueError(f"expected shape ({self.nfft},) or (B, {self.nfft}), got {x.shape}")
        return x, False

    @staticmethod
    def _out_array(out, shape, dtype) -> np.ndarray:
        """A caller-provided result array (reused across calls: no page faults on fresh memory, no munmap of
        the previous result — at 256 MiB those cost as much as the transfer) or a new one."""
        if out is None:
            return np.empty(shape, dtype=dtype)
        if not isinstance(out, np.ndarray) or out.dtype != dtype or out.shape != tuple(shape) or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous {np.dtype(dtype).name} array of shape {tuple(shape)}")
        return out

    def spectrum_db(self, samples, out: Optional[np.ndarray] = None) -> np.ndarray:
        """float32 ``20*log10(|fftshift(fft(w*x))| + eps)`` for one frame or a batch; ``out`` (same shape as
        ``samples``, float32) receives the rows in place when given."""
        x, one = self._frames(samples)
        res = self._out_array(out, x.shape[1:] if one else x.shape, np.float32)
        if x.shape[0]:
            self._run_host(lib().sdrk_exec_host, x, x.shape[0], self.nfft, res)
        return res

    def fft(self, samples) -> np.ndarray:
        """complex64 spectrum ``fft(w*x)`` (fftshifted if the plan shifts), no log."""
        x, one = self._frames(samples)
        out = np.empty(x.shape, dtype=np.complex64)
        if x.shape[0]:
            self._run_host(lib().sdrk_exec_fft_host, x, x.shape[0], self.nfft, out)
        return out[0] if one else out

    def stft_db(self, iq, hop: Optional[int] = None, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Rows of a spectrogram over one contiguous stream: row r covers samples
        ``[r*hop, r*hop + nfft)``; ``rows = 1 + (len(iq) - nfft) // hop`` (0 if the
        stream is shorter than one frame).  Frames are cut on the device from the
        single uploaded stream; overlapped samples are not duplicated on the host."""
        x = _as_c64(iq).reshape(-1)
        hop = self.nfft if hop is None else int(hop)
        if hop < 1:
            raise ValueError("hop must be >= 1")
        rows = 0 if x.shape[0] < self.nfft else 1 + (x.shape[0] - self.nfft) // hop
        out = self._out_array(out, (rows, self.nfft), np.float32)
        if rows:
            self._run_host(lib().sdrk_exec_host, x, rows, hop, out)
        return out

    def welch_psd(self, iq, sample_rate: float, hop: Optional[int] = None) -> np.ndarray:
        """Averaged periodogram of one contiguous stream, float32 ``(nfft,)``:
        ``mean_r |fft(w * x_r)|^2 / (sample_rate * sum(w^2))`` over the
        ``1 + (len - nfft)//hop`` full segments — matplotlib's ``mlab.psd`` (no detrend,
        two-sided; in fftshift order when the plan shifts), which is what the reference's
        offline script plots (scripts/process_sigmf_data.py:188-189).  The per-segment
        transforms and the averaging both run on the GPU."""
        x = _as_c64(iq).reshape(-1)
        hop = self.nfft if hop is None else int(hop)
        if hop < 1:
            raise ValueError("hop must be >= 1")
        if x.shape[0] < self.nfft:
            raise ValueError(f"stream of {x.shape[0]} samples is shorter than one {self.nfft}-sample segment")
        rows = 1 + (x.shape[0] - self.nfft) // hop
        if self._wkey == "rect":
            wss = float(self.nfft)
        elif self._wkey == "hann":
            wss = float(np.sum(np.hanning(self.nfft) ** 2))
        else:
            wss = float(np.sum(np.frombuffer(self._wkey[1], dtype=np.float32).astype(np.float64) ** 2))
        scale = 1.0 / (rows * float(sample_rate) * wss)
        out = np.empty(self.nfft, dtype=np.float32)
        with self._lock:
            check(lib().sdrk_welch_psd_host(self.handle, x.ctypes.data_as(c_void_p), c_size_t(rows), c_size_t(hop),
                                            c_float(scale), out.ctypes.data_as(c_void_p)))
        return out

    # -- device pointers (bench / pipelines that keep data resident) -------------
    def exec_device(self, d_iq: int, n_frames: int, d_out: int, *, frame_stride: Optional[int] = None,
                    stream: int = 0) -> None:
        stride = self.nfft if frame_stride is None else int(frame_stride)
        with self._lock:
            check(lib().sdrk_exec_device(self.handle, c_void_p(d_iq), c_size_t(n_frames), c_size_t(stride),
                                         c_void_p(d_out), c_void_p(stream) if stream else None))

    def tune_scratch(self, d_iq: int, n_frames: int, d_out: int, candidates: int = 6, *,
                     frame_stride: Optional[int] = None):
        """Large-frame plans: try ``candidates`` placements of the two-pass scratch on this workload and keep the
        fastest (``sdrk_plan_tune_scratch``).  Returns ``(probe_ms, chosen)``; ``d_out`` is overwritten."""
        stride = self.nfft if frame_stride is None else int(frame_stride)
        ms, chosen = (c_float * int(candidates))(), c_int(0)
        with self._lock:
            check(lib().sdrk_plan_tune_scratch(self.handle, c_void_p(d_iq), c_size_t(n_frames), c_size_t(stride),
                                               c_void_p(d_out), int(candidates), ms, byref(chosen)))
            self.last_placement = placement_report()
        return [float(v) for v in ms], int(chosen.value)

    def exec_device_timed_each(self, d_iq: int, n_frames: int, d_out: int, launches: int = 1, *,
                               frame_stride: Optional[int] = None) -> list:
        """Like ``exec_device_timed`` but returns the milliseconds of each launch (events between
        consecutive launches on the plan's stream)."""
        stride = self.nfft if frame_stride is None else int(frame_stride)
        ms = (c_float * int(launches))()
        with self._lock:
            check(lib().sdrk_exec_device_timed_each(self.handle, c_void_p(d_iq), c_size_t(n_frames),
                                                    c_size_t(stride), c_void_p(d_out), int(launches), ms))
        return [float(v) for v in ms]

    def exec_device_timed(self, d_iq: int, n_frames: int, d_out: int, launches: int = 1, *,
                          frame_stride: Optional[int] = None) -> float:
        """Run `launches` back-to-back transforms on the plan's stream; milliseconds
        between HIP events recorded on that stream (all launches together)."""
        stride = self.nfft if frame_stride is None else int(frame_stride)
        ms = c_float(0.0)
        with self._lock:
            check(lib().sdrk_exec_device_timed(self.handle, c_void_p(d_iq), c_size_t(n_frames),
                                               c_size_t(stride), c_void_p(d_out), int(launches), byref(ms)))
        return float(ms.value)

    def sync(self) -> None:
        check(lib().sdrk_plan_sync(self.handle))

    def fused_status(self) -> dict:
        """``nfft = 65536`` plans: ``{"launches": persistent launches made so far, "fallen_back": a launch reported a failed
        hand-over and the plan now takes the two tiled launches}`` (``sdrk_plan_fused_status``); zeros for other plans."""
        n, fb = _ffi.c_uint32(0), c_int(0)
        check(lib().sdrk_plan_fused_status(self.handle, byref(n), byref(fb)))
        return {"launches": int(n.value), "fallen_back": bool(fb.value)}

    def staging_probe(self) -> list:
        """Probe times (ms) of the staging candidates of a ``tune_staging=True`` plan, three per chunk slot; ``[]``
        when nothing was tuned."""
        ms, n = (c_float * 16)(), c_int(0)
        check(lib().sdrk_plan_staging_probe(self.handle, ms, 16, byref(n)))
        return [float(ms[i]) for i in range(min(int(n.value), 16))]


def placement_report() -> dict:
    """What the last placement probe on this thread did (``sdrk_placement_report``): the warm-up, candidate 0 as first timed
    and as timed again after the last candidate, the kept candidate's time (ms); ``gain_vs_retimed_first`` is what the kept
    candidate saves against candidate 0 measured WARM — the honest figure (a first timing taken before the shader clock had
    ramped up would credit the ramp to the placement)."""
    warm, n, first, again, kept = c_float(0), c_int(0), c_float(0), c_float(0), c_float(0)
    tried = int(lib().sdrk_placement_report(byref(warm), byref(n), byref(first), byref(again), byref(kept)))
    rec = {"candidates_tried": tried, "warmup_ms": round(float(warm.value), 2), "warmup_launches": int(n.value),
           "first_ms": round(float(first.value), 4), "retimed_first_ms": round(float(again.value), 4),
           "chosen_ms": round(float(kept.value), 4)}
    rec["gain_vs_retimed_first"] = (round(1.0 - kept.value / again.value, 4) if again.value > 0 and kept.value > 0 else None)
    return rec


# ---- plan cache for the function API -------------------------------------------
_plans: dict = {}
_plans_lock = threading.Lock()


def _cached_plan(nfft: int, window: WindowArg, eps: float, shift: bool, device: int) -> SpectrumPlan:
    _, _, wkey = _window_spec(window, nfft)
    key = (int(device), int(nfft), wkey, float(eps), bool(shift))
    plan = _plans.get(key)                      # (dict.get is atomic; the lock is only for creation)
    if plan is not None:
        return plan
    with _plans_lock:
        plan = _plans.get(key)
        if plan is None:
            plan = SpectrumPlan(nfft, window=window, eps=eps, shift=shift, device=device)
            _plans[key] = plan
        return plan


def clear_plan_cache() -> None:
    with _plans_lock:
        for p in _plans.values():
            p.close()
        _plans.clear()


def _nfft_of(samples) -> int:
    shape = np.shape(samples)
    if len(shape) not in (1, 2):
        raise ValueError(f"expected a frame (N,) or a batch (B, N), got shape {shape}")
    return int(shape[-1])


def spectrum_db(samples, *, window: WindowArg = None, eps: float = 1e-12, shift: bool = True,
                device: int = 0, devices: Optional[Sequence[int]] = None,
                out: Optional[np.ndarray] = None, pin="auto") -> np.ndarray:
    """Power spectrum in dB of one frame ``(N,)`` or a batch ``(B, N)`` of complex IQ.

    Equivalent to ``20*np.log10(np.abs(np.fft.fftshift(np.fft.fft(samples*window, axis=-1),
    axes=-1)) + eps)`` in float32 — with the defaults, exactly the reference's
    ``power_db`` (app/sdr/streamer.py:119,121).  ``devices=[0,1,...]`` splits a
    batch into contiguous frame ranges, one per GPU (no collectives; see
    sharding.py).  ``out``: a float32 array of the result's shape to fill instead of
    allocating (large batches: reusing it saves the page faults and the munmap of a
    fresh result per call).  ``pin`` (with ``devices``): how pageable arrays reach several GPUs — ``"auto"`` stages
    them until reuse has paid for page-locking them (sharding.spectrum_db_sharded, hostmem.plan_pinning).
    """
    nfft = _nfft_of(samples)
    if devices is not None and len(devices) > 1 and np.ndim(samples) == 2:
        from .sharding import spectrum_db_sharded
        return spectrum_db_sharded(samples, devices, window=window, eps=eps, shift=shift, out=out, pin=pin)
    if devices is not None and len(devices) == 1:
        device = devices[0]
    return _cached_plan(nfft, window, eps, shift, device).spectrum_db(samples, out=out)


def fft_c64(samples, *, window: WindowArg = None, shift: bool = False, device: int = 0) -> np.ndarray:
    """complex64 ``np.fft.fft(samples*window, axis=-1)`` (streamer.py:119 without the
    log), optionally fftshifted."""
    nfft = _nfft_of(samples)
    return _cached_plan(nfft, window, 1e-12, shift, device).fft(samples)


def freq_axis(n: int, sample_rate: float, center_freq: float = 0.0) -> np.ndarray:
    """float64 frequency axis in Hz, bit-identical to the reference's
    ``np.fft.fftshift(np.fft.fftfreq(n, 1/sample_rate)) + center_freq``
    (app/sdr/streamer.py:120): the same float operations in the same order —
    ``val = 1/(n*d)`` with ``d = 1/sample_rate``, integer bin index times ``val``,
    plus ``center_freq`` — on the already-shifted index range.  Host-side; it is
    O(n) float64 arithmetic with no kernel."""
    n = int(n)
    if n < 1:
        raise ValueError("n must be >= 1")
    key = (n, sample_rate, center_freq)
    cached = _freq_cache.get(key)
    if cached is None:
        d = 1 / sample_rate
        val = 1.0 / (n * d)
        k = np.arange(-(n // 2), (n - 1) // 2 + 1, dtype=int)
        cached = k * val + center_freq
        if len(_freq_cache) >= 8:               # the live app has one (n, fs, fc); keep a few
            _freq_cache.clear()
        _freq_cache[key] = cached
    return cached.copy()                        # a fresh array per call, like the reference's expression


_freq_cache: dict = {}


def process_frame(samples, sample_rate: float, center_freq: float, *, window: WindowArg = None,
                  eps: float = 1e-12, device: int = 0) -> dict:
    """One reader-loop iteration of the reference (app/sdr/streamer.py:119-130):
    returns the ``plot_data`` dict with exactly its keys — ``time``, ``samples``
    (the caller's array, same object), ``freqs``, ``power_db``, ``sample_rate``,
    ``center_freq`` — which is what ``update_graphs`` reads
    (app/dashboard/callbacks.py:110-115)."""
    power_db = spectrum_db(samples, window=window, eps=eps, shift=True, device=device)
    freqs = freq_axis(len(samples), sample_rate, center_freq)
    return {
        "time": time.time(),
        "samples": samples,
        "freqs": freqs,
        "power_db": power_db,
        "sample_rate": sample_rate,
        "center_freq": center_freq,
    }


def welch_psd(iq, nfft: int, sample_rate: float, hop: Optional[int] = None, window: WindowArg = "hann", *,
              shift: bool = True, device: int = 0) -> np.ndarray:
    """Linear two-sided PSD (power per Hz) as ``matplotlib.mlab.psd`` computes it for the
    reference's offline plots (scripts/process_sigmf_data.py:188-189: NFFT=1024, Hann,
    noverlap=0).  Returns float32 ``(nfft,)`` in fftshift order (use ``freq_axis`` for x)."""
    return _cached_plan(int(nfft), window, 1e-12, shift, device).welch_psd(iq, sample_rate, hop)


def stft_db(iq, nfft: int, hop: Optional[int] = None, window: WindowArg = None, *, eps: float = 1e-12,
            shift: bool = True, device: int = 0, devices: Optional[Sequence[int]] = None) -> np.ndarray:
    """Spectrogram rows ``(rows, nfft)`` float32 over one contiguous IQ stream (the
    waterfall of BASELINE.json config 3: nfft=65536, hop=nfft//2).  ``devices=[...]`` splits the
    rows into contiguous ranges, one per GPU, each reading its samples plus an ``nfft-hop`` halo."""
    if devices is not None and len(devices) > 1:
        from .sharding import stft_db_sharded
        return stft_db_sharded(iq, int(nfft), int(nfft if hop is None else hop), devices, window=window, eps=eps,
                               shift=shift)
    if devices is not None and len(devices) == 1:
        device = devices[0]
    return _cached_plan(int(nfft), window, eps, shift, device).stft_db(iq, hop)
