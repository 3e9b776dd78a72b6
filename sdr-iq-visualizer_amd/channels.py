"""Several independent IQ channels, one per GPU, each with its own device-resident waterfall
(BASELINE.json config 5: 8 channels x N = 2^20, continuous waterfall, host gather).

There is nothing to exchange between channels — each is the reference's reader loop
(app/sdr/streamer.py:95-133) plus its dashboard deque (app/dashboard/callbacks.py:19,176-182)
on a different radio — so the multi-GPU form is one host thread per device driving that
device's plan and ring; the "gather" is each thread copying its (decimated) rows into its
slice of one host array.  ctypes releases the GIL during the C calls, so the threads overlap.
"""
from __future__ import annotations

import threading
from typing import Callable, List, Optional, Sequence

import numpy as np

from .waterfall import WaterfallBuffer


class ChannelBank:
    """``len(devices)`` channels; channel c lives on ``devices[c]``."""

    def __init__(self, nfft: int, devices: Sequence[int], *, maxlen: int = 100, window=None, eps: float = 1e-12):
        self.nfft, self.devices = int(nfft), list(devices)
        self.rings: List[WaterfallBuffer] = [
            WaterfallBuffer(nfft, maxlen, device=d, window=window, eps=eps) for d in self.devices
        ]

    def close(self) -> None:
        for r in self.rings:
            r.close()

    def __enter__(self) -> "ChannelBank":
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    def _per_channel(self, fn: Callable[[int], None]) -> None:
        errors: List[BaseException] = []

        def run(c: int) -> None:
            try:
                fn(c)
            except BaseException as e:  # re-raised on the caller's thread
                errors.append(e)

        threads = [threading.Thread(target=run, args=(c,)) for c in range(len(self.rings))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]

    def append_iq(self, frames_per_channel: Sequence, hop: Optional[int] = None) -> None:
        """``frames_per_channel[c]``: IQ for channel c (``(nfft,)``, ``(B, nfft)`` or, with ``hop``, a stream)."""
        if len(frames_per_channel) != len(self.rings):
            raise ValueError(f"expected {len(self.rings)} channel inputs, got {len(frames_per_channel)}")
        self._per_channel(lambda c: self.rings[c].append_iq(frames_per_channel[c], hop=hop))

    # -- IQ already on each channel's device: a continuous bank with nothing waited for that does not have to be --
    def append_iq_device(self, d_iq_per_channel: Sequence[int], n_frames: int, frame_stride: Optional[int] = None, *,
                         wait: bool = True) -> None:
        """``d_iq_per_channel[c]``: device pointer (on ``devices[c]``) to ``n_frames`` frames of complex64 IQ for channel
        c.  ``wait=False`` only enqueues the transforms on each ring's stream (``WaterfallBuffer.append_iq_device``)."""
        if len(d_iq_per_channel) != len(self.rings):
            raise ValueError(f"expected {len(self.rings)} device pointers, got {len(d_iq_per_channel)}")
        if wait:
            self._per_channel(lambda c: self.rings[c].append_iq_device(d_iq_per_channel[c], n_frames, frame_stride))
        else:                       # enqueue-only calls return in microseconds: no threads needed
            for c, ring in enumerate(self.rings):
                ring.append_iq_device(d_iq_per_channel[c], n_frames, frame_stride, wait=False)

    def gather_begin(self, max_rows: int, *, decimate: int = 1, mode: str = "max", out: Optional[np.ndarray] = None) -> np.ndarray:
        """First half of a decimated gather of the newest ``max_rows`` rows of every channel into ONE host array
        ``(channels, max_rows, nfft // decimate)`` (``out``: ideally ``pinned_empty``): every channel's reduction and
        copy are enqueued, nothing is waited for.  Every channel must hold at least ``max_rows`` rows."""
        rows, bins = int(max_rows), self.nfft // decimate
        if min(len(r) for r in self.rings) < rows:
            raise ValueError(f"every channel needs at least {rows} rows")
        if out is None:
            out = np.empty((len(self.rings), rows, bins), dtype=np.float32)
        elif out.shape != (len(self.rings), rows, bins) or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous float32 array of shape {(len(self.rings), rows, bins)}")
        for c, ring in enumerate(self.rings):
            ring.gather_begin(max_rows=rows, decimate=decimate, mode=mode, out=out[c])
        self._gather = out
        return out

    def gather_end(self) -> Optional[np.ndarray]:
        """Second half: wait for every channel's copy; returns the array ``gather_begin`` filled."""
        for ring in self.rings:
            ring.gather_end()
        g, self._gather = getattr(self, "_gather", None), None
        return g

    def sync(self) -> None:
        for ring in self.rings:
            ring.sync()

    def gather(self, max_rows: Optional[int] = None, *, decimate: int = 1, mode: str = "max") -> np.ndarray:
        """Host gather: float32 ``(channels, rows, nfft // decimate)``; channels with fewer rows are
        NaN-padded at the top (oldest side)."""
        rows = max(len(r) for r in self.rings)
        if max_rows is not None:
            rows = min(rows, int(max_rows))
        out = np.full((len(self.rings), rows, self.nfft // decimate), np.nan, dtype=np.float32)

        def pull(c: int) -> None:
            a = self.rings[c].as_array(max_rows=rows, decimate=decimate, mode=mode)
            out[c, rows - a.shape[0]:] = a

        self._per_channel(pull)
        return out
