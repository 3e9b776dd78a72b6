"""Reader-loop shim: the reference's ``SDRDataStreamer`` interface around the GPU
spectrum path, with a pluggable sample source instead of a PlutoSDR.

Mirrors, method for method, what the dashboard and the chatbot call
(app/sdr/streamer.py):

    start_streaming (:53-60)   stop_streaming (:62-65)   is_connected (:49-51)
    _stream_data (:95-174)     _push drop-oldest (:186-194)
    get_latest_data FIFO pop (:196-200)                  get_status (:176-184)

The three numpy lines of the loop (:119-121) are the only thing replaced: each
iteration calls ``process_frame`` (GPU) and pushes the same ``plot_data`` dict
(:123-130).  Error handling keeps the reference's shape: any exception from the source
or the transform is logged and counted, the loop backs off 0.1 s → 1.6 s, and after 3
consecutive failures the stream stops (:157-174; the SDR-specific reconnect branches
have no counterpart without a radio).

A ``source`` is any object with ``rx() -> complex array`` (what ``adi.Pluto`` offers,
:114); ``SyntheticSource`` and ``SigMFSource`` are provided.
"""
from __future__ import annotations

import logging
import queue
import threading
import time
from typing import Callable, Optional

import numpy as np

from . import sigmf_io, synth
from .spectrum import process_frame

logger = logging.getLogger(__name__)


class SyntheticSource:
    """12-bit integer IQ frames from the package generator, optionally paced."""

    def __init__(self, nfft: int = 4096, seed: int = 1234, tone_bin: Optional[float] = 300.0,
                 tone_amp: float = 400.0, frame_period_s: float = 0.0):
        self.nfft, self.seed, self.frame = int(nfft), int(seed), 0
        self.tone = synth.tone(self.nfft, tone_bin, tone_amp) if tone_bin is not None else None
        self.frame_period_s = frame_period_s

    def rx(self) -> np.ndarray:
        x = synth.synth_iq(self.seed, self.frame, 1, self.nfft)[0]
        self.frame += 1
        if self.tone is not None:
            x = (x + self.tone).astype(np.complex64)
        if self.frame_period_s:
            time.sleep(self.frame_period_s)
        return x


class SigMFSource:
    """Frames cut from a cf32_le SigMF recording (BASELINE.json config 1), looping."""

    def __init__(self, path: str, nfft: int = 4096, loop: bool = True):
        self.samples, self.meta = sigmf_io.read_sigmf(path)
        self.nfft, self.loop, self.pos = int(nfft), loop, 0
        if self.samples.size < self.nfft:
            raise ValueError(f"recording has {self.samples.size} samples, need at least {self.nfft}")

    def rx(self) -> np.ndarray:
        if self.pos + self.nfft > self.samples.size:
            if not self.loop:
                raise EOFError("end of recording")
            self.pos = 0
        x = self.samples[self.pos: self.pos + self.nfft]
        self.pos += self.nfft
        return x


class SpectrumStreamer:
    def __init__(self, source, sample_rate: float = 1_000_000, center_freq: float = 2_400_000_000, *,
                 window=None, device: int = 0, queue_size: int = 100,
                 compute: Optional[Callable[..., dict]] = None):
        self.sdr = source                                   # same attribute name as the reference
        self.sample_rate, self.center_freq = sample_rate, center_freq
        self.data_queue: "queue.Queue[dict]" = queue.Queue(maxsize=queue_size)   # :18
        self.running = False
        self.thread: Optional[threading.Thread] = None
        self.connected = source is not None
        self.last_success_ts: Optional[float] = None
        self.total_frames = 0
        # per-frame transform; the default is the GPU path.  (CPU tests inject the oracle here to
        # exercise the queue / error logic without a device.)
        self._compute = compute or (lambda s, fs, fc: process_frame(s, fs, fc, window=window, device=device))

    def is_connected(self) -> bool:
        return self.sdr is not None and self.connected      # :49-51

    def start_streaming(self) -> bool:
        if not self.sdr:
            logger.error("source not connected")
            return False
        self.running = True
        self.thread = threading.Thread(target=self._stream_data, daemon=True)
        self.thread.start()
        return True

    def stop_streaming(self) -> None:
        self.running = False
        if self.thread:
            self.thread.join(timeout=2)

    def _stream_data(self) -> None:
        consecutive_errors, backoff, max_backoff = 0, 0.1, 1.6
        self.last_success_ts, self.total_frames = None, 0
        while self.running:
            try:
                samples = self.sdr.rx()
                plot_data = self._compute(samples, self.sample_rate, self.center_freq)
                consecutive_errors, backoff = 0, 0.1
                self._push(plot_data)
                self.last_success_ts = plot_data["time"]
                self.total_frames += 1
            except Exception as e:  # noqa: BLE001 - the reference catches everything here (:157)
                consecutive_errors += 1
                logger.error("stream iteration failed: %s", e)
                if consecutive_errors >= 3:
                    logger.error("3 consecutive errors; stopping stream.")
                    self.running = False
                    break
                time.sleep(backoff)
                backoff = min(backoff * 2, max_backoff)

    def get_status(self) -> dict:
        return {
            "connected": self.connected,
            "running": self.running,
            "queue_size": self.data_queue.qsize(),
            "last_success_age_ms": (time.time() - self.last_success_ts) * 1000 if self.last_success_ts else None,
            "total_frames": self.total_frames,
        }

    def _push(self, data: dict) -> None:
        try:
            self.data_queue.put_nowait(data)
        except queue.Full:                                   # drop the oldest (:186-194)
            try:
                self.data_queue.get_nowait()
                self.data_queue.put_nowait(data)
            except queue.Empty:
                pass

    def get_latest_data(self) -> Optional[dict]:
        try:
            return self.data_queue.get_nowait()              # FIFO: the OLDEST queued frame (:196-200)
        except queue.Empty:
            return None
