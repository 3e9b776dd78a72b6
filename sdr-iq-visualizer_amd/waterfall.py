"""Device-resident waterfall: the last ``maxlen`` spectrum rows, oldest first.

Host-side mirror of the reference's waterfall accumulation in its dashboard
callback:

    app/dashboard/callbacks.py:19   waterfall_data = deque(maxlen=100)
    app/dashboard/callbacks.py:176  waterfall_data.append(power_db)
    app/dashboard/callbacks.py:182  waterfall_array = np.array(waterfall_data)
    app/dashboard/callbacks.py:186  y=list(range(len(waterfall_data)))

``WaterfallBuffer`` keeps the rows in a ring in HBM (``sdrk_waterfall_*`` in
include/sdrk.h).  ``append`` takes either a finished ``power_db`` row (what the
callback has today) or raw IQ frames, which are transformed on the GPU straight
into the ring slots; ``as_array`` returns the ``(rows, nfft)`` float32 array the
heatmap is drawn from.  The reference mutates its module-global deque from Flask
request threads without a lock; this class takes one.
"""
from __future__ import annotations

import threading
from typing import Optional

import numpy as np

from . import _ffi
from ._ffi import byref, c_size_t, c_void_p, check, lib
from .spectrum import SpectrumPlan, WindowArg


class WaterfallBuffer:
    def __init__(self, nfft: int, maxlen: int = 100, *, device: int = 0, window: WindowArg = None,
                 eps: float = 1e-12):
        nfft, maxlen = int(nfft), int(maxlen)
        if nfft < 1:
            raise ValueError("nfft must be >= 1")
        if maxlen < 1:
            raise ValueError("maxlen must be >= 1")
        _ffi.require_device(device)
        self.nfft, self.maxlen, self.device = nfft, maxlen, int(device)
        self._window, self._eps = window, float(eps)
        self._plan: Optional[SpectrumPlan] = None
        # the arrays that gather_begin() calls are filling, oldest first (kept alive until their gather_end()); at most two
        self._gather: Optional[list] = None
        self._lock = threading.Lock()
        self._handle = c_void_p()
        check(lib().sdrk_waterfall_create(self.device, nfft, maxlen, byref(self._handle)))

    # -- lifetime ---------------------------------------------------------------
    def close(self) -> None:
        # under the lock every other method holds across its C call: an append / read in flight on another thread (the
        # reference appends from Flask request threads, callbacks.py:96,176) finishes on the live ring, and whoever comes
        # after finds the handle gone ("waterfall is closed") instead of a destroyed one
        with self._lock:
            h, self._handle = self._handle, c_void_p()
            if h:
                lib().sdrk_waterfall_destroy(h)   # (waits for a copy that gather_begin may have left in flight)
            self._gather = None
            if self._plan is not None:
                self._plan.close()
                self._plan = None

    def __enter__(self) -> "WaterfallBuffer":
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    def _h(self) -> c_void_p:
        if not self._handle:
            raise RuntimeError("waterfall is closed")
        return self._handle

    # -- deque-like interface ------------------------------------------------------
    def __len__(self) -> int:
        with self._lock:
            return self._rows()

    def _rows(self) -> int:                       # caller holds the lock
        return check(lib().sdrk_waterfall_rows(self._h()))

    def maxhold16_rows(self) -> int:
        """How many of the valid rows carry the by-16 max-hold companion the N >= 2^20 transform writes beside the ring
        (``sdrk_waterfall_maxhold16_rows``): a max-mode ``decimate=`` read-out with a factor that is a multiple of 16 over
        such rows reads 1/16 of the bytes."""
        with self._lock:
            return check(lib().sdrk_waterfall_maxhold16_rows(self._h()))

    def clear(self) -> None:
        with self._lock:
            check(lib().sdrk_waterfall_clear(self._h()))

    def append(self, row_or_frame) -> None:
        """Append one row (real, shape ``(nfft,)``), several rows ``(r, nfft)``, or —
        if the array is complex — one or several IQ frames to transform first."""
        a = np.asarray(row_or_frame)
        if np.iscomplexobj(a):
            self.append_iq(a)
        else:
            self.append_rows(a)

    def append_rows(self, rows) -> None:
        r = np.ascontiguousarray(np.asarray(rows, dtype=np.float32))
        if r.ndim == 1:
            r = r.reshape(1, -1)
        if r.ndim != 2 or r.shape[1] != self.nfft:
            raise ValueError(f"expected rows of {self.nfft} values, got shape {r.shape}")
        with self._lock:
            check(lib().sdrk_waterfall_append_rows(self._h(), r.ctypes.data_as(c_void_p), c_size_t(r.shape[0])))

    def append_iq(self, frames, hop: Optional[int] = None) -> None:
        """Transform IQ and append the rows.  ``frames`` is ``(nfft,)``, ``(B, nfft)``,
        or — with ``hop`` — one contiguous stream cut into overlapping frames."""
        x = np.asarray(frames)
        if x.dtype != np.complex64:
            x = x.astype(np.complex64)
        x = np.ascontiguousarray(x)
        if hop is None:
            if x.ndim == 1:
                x = x.reshape(1, -1)
            if x.ndim != 2 or x.shape[1] != self.nfft:
                raise ValueError(f"expected frames of {self.nfft} samples, got shape {x.shape}")
            n_frames, stride = x.shape[0], self.nfft
        else:
            x = x.reshape(-1)
            stride = int(hop)
            if stride < 1:
                raise ValueError("hop must be >= 1")
            n_frames = 0 if x.shape[0] < self.nfft else 1 + (x.shape[0] - self.nfft) // stride
        if n_frames == 0:
            return
        with self._lock:
            check(lib().sdrk_waterfall_append_iq(self._h(), self._ensure_plan().handle, x.ctypes.data_as(c_void_p),
                                                 c_size_t(n_frames), c_size_t(stride)))

    # -- a continuous channel whose IQ is already on the device (BASELINE config 5) ---------------
    def _ensure_plan(self) -> SpectrumPlan:
        if self._plan is None:
            self._plan = SpectrumPlan(self.nfft, window=self._window, eps=self._eps, shift=True, device=self.device)
        return self._plan

    def append_iq_device(self, d_iq: int, n_frames: int, frame_stride: Optional[int] = None, *, wait: bool = True) -> None:
        """Transform ``n_frames`` frames of device-resident complex64 IQ (device pointer ``d_iq``) straight into the
        ring.  ``wait=False`` only enqueues the work on the ring's stream: the next read / ``sync()`` is ordered behind
        it, and the IQ buffer must stay untouched until then."""
        stride = self.nfft if frame_stride is None else int(frame_stride)
        with self._lock:
            fn = lib().sdrk_waterfall_append_iq_device if wait else lib().sdrk_waterfall_append_iq_device_async
            check(fn(self._h(), self._ensure_plan().handle, c_void_p(int(d_iq)), c_size_t(int(n_frames)), c_size_t(stride)))

    def sync(self) -> None:
        with self._lock:
            check(lib().sdrk_waterfall_sync(self._h(), self._plan.handle if self._plan is not None else None))

    def gather_begin(self, max_rows: Optional[int] = None, *, decimate: int = 1, mode: str = "max",
                     out: Optional[np.ndarray] = None) -> np.ndarray:
        """First half of a decimated read-out: the reduction is enqueued behind everything appended so far and its
        result starts crossing PCIe on a second stream; returns the (not yet filled) ``(rows, nfft // decimate)`` array.
        Call ``gather_end()`` before using it.  Between the two, ``append_iq_device(..., wait=False)`` of the next
        batch overlaps this batch's copy.  ``out``: a float32 array (ideally ``pinned_empty``) with room for the rows.
        Up to TWO read-outs may be in flight (``gather_end()`` completes the oldest): a channel that enqueues batch i + 1 and
        begins its read-out before it collects batch i - 1 never lets the transform stream run dry."""
        if mode not in ("max", "mean"):
            raise ValueError("mode must be 'max' or 'mean'")
        if decimate < 1 or self.nfft % decimate:
            raise ValueError(f"decimate={decimate} must divide nfft={self.nfft}")
        with self._lock:
            rows = self._rows() if max_rows is None else min(self._rows(), int(max_rows))
            bins = self.nfft // decimate
            if out is None:
                out = np.empty((rows, bins), dtype=np.float32)
            elif out.dtype != np.float32 or not out.flags.c_contiguous or out.ndim != 2 or out.shape[1] != bins or out.shape[0] < rows:
                raise ValueError(f"out must be a C-contiguous float32 array of at least ({rows}, {bins})")
            got = c_size_t(0)
            check(lib().sdrk_waterfall_read_decimated_begin(self._h(), out.ctypes.data_as(c_void_p), c_size_t(rows),
                                                            int(decimate), 0 if mode == "max" else 1, byref(got)))
            g = out[: got.value]
            self._gather = (self._gather or []) + [g]      # keeps the array alive until its gather_end()
            return g

    def gather_end(self) -> Optional[np.ndarray]:
        """Second half: wait for the copy started by the OLDEST ``gather_begin`` still in flight and return its array
        (``None`` when nothing is in flight)."""
        with self._lock:
            check(lib().sdrk_waterfall_read_decimated_end(self._h()))
            if not self._gather:
                return None
            g, rest = self._gather[0], self._gather[1:]
            self._gather = rest or None
            return g

    def as_array(self, max_rows: Optional[int] = None, *, decimate: int = 1, mode: str = "max") -> np.ndarray:
        """``np.array(deque)``: float32 ``(rows, nfft)``, oldest row at index 0.  With
        ``max_rows`` only the newest ``max_rows`` rows are returned.  ``decimate=f`` reduces every
        run of ``f`` bins on the device (``mode`` "max" = peak hold, "mean" = mean of dB values)
        and returns ``(rows, nfft // f)`` — for drawing long rows (N = 2^20) in a heatmap."""
        if decimate != 1:
            if mode not in ("max", "mean"):
                raise ValueError("mode must be 'max' or 'mean'")
            if decimate < 1 or self.nfft % decimate:
                raise ValueError(f"decimate={decimate} must divide nfft={self.nfft}")
            with self._lock:
                rows = self._rows() if max_rows is None else min(self._rows(), int(max_rows))
                out = np.empty((rows, self.nfft // decimate), dtype=np.float32)
                got = c_size_t(0)
                check(lib().sdrk_waterfall_read_decimated(self._h(), out.ctypes.data_as(c_void_p), c_size_t(rows),
                                                          int(decimate), 0 if mode == "max" else 1, byref(got)))
                return out[: got.value]
        with self._lock:
            rows = self._rows()
            if max_rows is not None:
                rows = min(rows, int(max_rows))
            out = np.empty((rows, self.nfft), dtype=np.float32)
            got = c_size_t(0)
            check(lib().sdrk_waterfall_read(self._h(), out.ctypes.data_as(c_void_p), c_size_t(rows), byref(got)))
            return out[: got.value]
