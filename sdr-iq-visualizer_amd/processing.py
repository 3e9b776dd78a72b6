"""``processing`` — the numpy-in / numpy-out module BASELINE.json's north-star
describes for the streamed-IQ FFT / PSD / waterfall path.

The reference's ``app/processing`` holds only the classifier (a consumer of
``power_db``); the spectrum itself is inline at app/sdr/streamer.py:119-121 and
the waterfall at app/dashboard/callbacks.py:176-182.  A maintainer switching to
this build imports these names there (INTEGRATION.md shows the three-line
change); the Dash callbacks keep reading the same ``plot_data`` dict.
"""
from .spectrum import fft_c64, freq_axis, process_frame, spectrum_db, stft_db, welch_psd  # noqa: F401
from .waterfall import WaterfallBuffer  # noqa: F401

__all__ = ["spectrum_db", "fft_c64", "freq_axis", "process_frame", "stft_db", "welch_psd", "WaterfallBuffer"]
