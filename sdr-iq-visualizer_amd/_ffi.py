"""ctypes binding of the C ABI in ``include/sdrk.h`` (``lib/libsdrk.so``).

This is the only module that touches the shared library.  It declares every
exported symbol with its argument types, turns negative ``sdrk_status`` codes
into Python exceptions, and refuses to go on without the library: there is no
numpy fallback behind the product API (the reference's own producer loop
catches ordinary exceptions at ``app/sdr/streamer.py:157-159``, so raising is
the drop-in-compatible failure mode).

HIP runtime note: the library links ``libamdhip64.so.7``.  A process that also
uses PyTorch must ``import torch`` *before* this module so that both share the
single HIP runtime torch bundles; ``bench.py`` and ``__graft_entry__`` do so.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_size_t, c_uint32, c_uint64, c_void_p

SDRK_OK = 0
SDRK_ERR_INVALID = -1
SDRK_ERR_NO_DEVICE = -2
SDRK_ERR_HIP = -3
SDRK_ERR_NOMEM = -4
SDRK_ERR_UNSUPPORTED = -5

WINDOW_RECT = 0
WINDOW_HANN = 1
WINDOW_CUSTOM = 2

MAX_LOG2_NFFT = 22
ABI_VERSION = 500                 # SDRK_VERSION of include/sdrk.h this binding was written against
FEAT_PLANES = 19                  # SDRK_FEAT_* plane numbers of include/sdrk.h
(FEAT_MAX_DB, FEAT_NOISE_FLOOR_DB, FEAT_SNR_DB, FEAT_FLATNESS, FEAT_KURTOSIS, FEAT_THRESHOLD_DB, FEAT_PEAK_SPACING_STD_HZ,
 FEAT_PEAK_DENSITY, FEAT_BANDWIDTH_HZ) = range(9)
FEAT_ARGMAX, FEAT_PEAK_COUNT, FEAT_OCCUPIED_BINS = 11, 12, 13
PLAN_FUSED64K = 0x1
PLAN_OVERLAP_PASSES = 0x2
PLAN_TUNE_STAGING = 0x4
PLAN_TILED64K = 0x8

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# SDRK_LIB: developer override (A/B builds made by tools/variant.sh); the product loads lib/libsdrk.so
_LIB_PATH = os.environ.get("SDRK_LIB") or os.path.join(_PKG_DIR, "lib", "libsdrk.so")


class SdrkError(RuntimeError):
    """A call into libsdrk failed (HIP error, no device, out of memory...)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"libsdrk status {status}: {message}")
        self.status = status


# (name, restype, argtypes) — must list every symbol include/sdrk.h declares;
# tests/test_abi.py checks this table against the header and the built library.
SYMBOLS = [
    ("sdrk_version", c_int, []),
    ("sdrk_last_error", c_char_p, []),
    ("sdrk_device_count", c_int, []),
    ("sdrk_device_info", c_int, [c_int, c_char_p, c_size_t]),
    ("sdrk_dev_mem_info", c_int, [c_int, POINTER(c_size_t), POINTER(c_size_t)]),
    ("sdrk_dev_alloc", c_int, [c_int, c_size_t, POINTER(c_void_p)]),
    ("sdrk_dev_free", c_int, [c_int, c_void_p]),
    ("sdrk_dev_alloc_stream_pair", c_int, [c_int, c_size_t, c_size_t, c_int, c_void_p, POINTER(c_void_p),
                                           POINTER(c_void_p), POINTER(c_float), POINTER(c_int)]),
    ("sdrk_memcpy_h2d", c_int, [c_int, c_void_p, c_void_p, c_size_t]),
    ("sdrk_memcpy_d2h", c_int, [c_int, c_void_p, c_void_p, c_size_t]),
    ("sdrk_plan_create", c_int,
     [c_int, c_int, c_size_t, c_int, c_void_p, c_float, c_int, POINTER(c_void_p)]),
    ("sdrk_plan_create_ex", c_int,
     [c_int, c_int, c_size_t, c_int, c_void_p, c_float, c_int, c_uint32, POINTER(c_void_p)]),
    ("sdrk_plan_staging_probe", c_int, [c_void_p, POINTER(c_float), c_int, POINTER(c_int)]),
    ("sdrk_plan_fused_status", c_int, [c_void_p, POINTER(c_uint32), POINTER(c_int)]),
    ("sdrk_plan_tune_scratch", c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_int, POINTER(c_float),
                                       POINTER(c_int)]),
    ("sdrk_placement_report", c_int, [POINTER(c_float), POINTER(c_int), POINTER(c_float), POINTER(c_float), POINTER(c_float)]),
    ("sdrk_plan_destroy", c_int, [c_void_p]),
    ("sdrk_plan_nfft", c_int, [c_void_p]),
    ("sdrk_plan_device", c_int, [c_void_p]),
    ("sdrk_exec_host", c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p]),
    ("sdrk_exec_device", c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p]),
    ("sdrk_exec_fft_host", c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p]),
    ("sdrk_welch_psd_host", c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_float, c_void_p]),
    ("sdrk_plan_sync", c_int, [c_void_p]),
    ("sdrk_exec_device_timed", c_int,
     [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_int, POINTER(c_float)]),
    ("sdrk_exec_device_timed_each", c_int,
     [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_int, POINTER(c_float)]),
    ("sdrk_stream_ceiling_probe", c_int, [c_int, c_void_p, c_void_p, c_size_t, c_int, POINTER(c_float)]),
    ("sdrk_host_alloc", c_int, [c_size_t, POINTER(c_void_p)]),
    ("sdrk_host_free", c_int, [c_void_p]),
    ("sdrk_host_register", c_int, [c_void_p, c_size_t]),
    ("sdrk_host_unregister", c_int, [c_void_p]),
    ("sdrk_host_is_pinned", c_int, [c_void_p, c_size_t]),
    ("sdrk_copy_probe", c_int, [c_int, c_void_p, c_void_p, c_size_t, c_int, POINTER(c_float)]),
    ("sdrk_host_link_probe", c_int, [c_int, c_size_t, POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    ("sdrk_host_threads", c_int, []),
    ("sdrk_synth_fill", c_int, [c_int, c_uint32, c_uint64, c_size_t, c_int, c_void_p, c_void_p]),
    ("sdrk_row_features", c_int, [c_int, c_void_p, c_int, c_size_t, c_int, c_int, c_float, c_int, c_int,
                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    ("sdrk_frame_features_device", c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_int, c_float, c_int, c_int,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("sdrk_frame_features_host", c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_int, c_float, c_int, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("sdrk_row_features_planes", c_int, [c_int, c_void_p, c_int, c_size_t, c_int, c_int, c_float, c_int, c_int,
                                         c_void_p, c_void_p, c_void_p]),
    ("sdrk_frame_features_host_planes", c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_int, c_float, c_int, c_int,
                                                c_void_p, c_void_p, c_void_p, c_void_p]),
    ("sdrk_row_stats", c_int, [c_int, c_void_p, c_int, c_size_t, c_int, c_int, c_void_p]),
    ("sdrk_row_peaks", c_int, [c_int, c_void_p, c_int, c_size_t, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    ("sdrk_waterfall_create", c_int, [c_int, c_int, c_int, POINTER(c_void_p)]),
    ("sdrk_waterfall_destroy", c_int, [c_void_p]),
    ("sdrk_waterfall_append_rows", c_int, [c_void_p, c_void_p, c_size_t]),
    ("sdrk_waterfall_append_iq", c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t]),
    ("sdrk_waterfall_append_iq_device", c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t]),
    ("sdrk_waterfall_append_iq_device_async", c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t]),
    ("sdrk_waterfall_sync", c_int, [c_void_p, c_void_p]),
    ("sdrk_waterfall_read_decimated_begin", c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, POINTER(c_size_t)]),
    ("sdrk_waterfall_read_decimated_end", c_int, [c_void_p]),
    ("sdrk_waterfall_rows", c_int, [c_void_p]),
    ("sdrk_waterfall_read", c_int, [c_void_p, c_void_p, c_size_t, POINTER(c_size_t)]),
    ("sdrk_waterfall_read_decimated", c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, POINTER(c_size_t)]),
    ("sdrk_waterfall_maxhold16_rows", c_int, [c_void_p]),
    ("sdrk_waterfall_clear", c_int, [c_void_p]),
]

_lib = None
_lib_lock = threading.Lock()


def library_path() -> str:
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    """Load (once) and return the shared library; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise ImportError(
                f"{_LIB_PATH} is missing: build it with `make -C {os.path.join(_PKG_DIR, 'csrc')}` "
                "(or python -c 'import __graft_entry__ as g; g.build()'). "
                "There is no CPU fallback for the spectrum path."
            )
        handle = ctypes.CDLL(_LIB_PATH)
        handle.sdrk_version.restype = c_int
        built = int(handle.sdrk_version())
        if built != ABI_VERSION:         # a stale build: say so, instead of an AttributeError on some newer symbol
            raise ImportError(f"{_LIB_PATH} reports ABI version {built}, this package expects {ABI_VERSION}: rebuild it "
                              f"with `make -C {os.path.join(_PKG_DIR, 'csrc')}`")
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(handle, name)  # AttributeError if the build lacks a symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
        return _lib


def check(status: int) -> int:
    """Raise for a negative sdrk_status; pass non-negative values through."""
    if status >= 0:
        return status
    msg = lib().sdrk_last_error()
    text = msg.decode("utf-8", "replace") if msg else "unknown error"
    if status == SDRK_ERR_INVALID:
        raise ValueError(f"libsdrk: {text}")
    if status == SDRK_ERR_NOMEM:
        raise MemoryError(f"libsdrk: {text}")
    raise SdrkError(status, text)


def device_count() -> int:
    return int(lib().sdrk_device_count())


def device_info(device: int = 0) -> str:
    buf = ctypes.create_string_buffer(256)
    check(lib().sdrk_device_info(device, buf, len(buf)))
    return buf.value.decode()


def require_device(device: int = 0) -> None:
    """Fail loudly when there is no usable GPU (no silent host path)."""
    n = device_count()
    if n <= 0:
        raise SdrkError(SDRK_ERR_NO_DEVICE, "no HIP device visible to this process")
    if not 0 <= device < n:
        raise SdrkError(SDRK_ERR_NO_DEVICE, f"device {device} out of range ({n} visible)")


__all__ = [
    "SdrkError", "SYMBOLS", "lib", "check", "device_count", "device_info", "require_device",
    "library_path", "byref", "c_void_p", "c_size_t", "c_float", "c_int",
    "WINDOW_RECT", "WINDOW_HANN", "WINDOW_CUSTOM", "MAX_LOG2_NFFT",
]
