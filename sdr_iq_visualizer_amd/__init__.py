"""Import name of the MI355X-native IQ spectrum path.

The project layout keeps the package sources in ``sdr-iq-visualizer_amd/`` (the
repository's name for it, with a hyphen, which Python cannot import directly).
This shim package gives it an importable name by putting that directory on its
``__path__``; every submodule (``spectrum``, ``waterfall``, ``synth``,
``sharding``, ``processing``, ``_ffi``) lives there, next to ``csrc/`` (the HIP
kernels and the C ABI) and ``lib/libsdrk.so`` (the built library).
"""
import os as _os

_SRC_DIR = _os.path.join(
    _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "sdr-iq-visualizer_amd"
)
if not _os.path.isdir(_SRC_DIR):  # pragma: no cover - broken checkout
    raise ImportError(f"package sources not found at {_SRC_DIR}")
__path__.append(_SRC_DIR)

from .spectrum import (  # noqa: E402
    SpectrumPlan,
    fft_c64,
    freq_axis,
    process_frame,
    spectrum_db,
    stft_db,
    welch_psd,
)
from .waterfall import WaterfallBuffer  # noqa: E402
from .hostmem import is_pinned, pinned_empty, registered  # noqa: E402
from ._ffi import SdrkError, device_count, device_info, library_path  # noqa: E402

__all__ = [
    "SpectrumPlan",
    "WaterfallBuffer",
    "SdrkError",
    "device_count",
    "device_info",
    "fft_c64",
    "freq_axis",
    "is_pinned",
    "library_path",
    "pinned_empty",
    "process_frame",
    "registered",
    "spectrum_db",
    "stft_db",
    "welch_psd",
]
