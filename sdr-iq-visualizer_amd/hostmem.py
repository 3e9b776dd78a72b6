"""Pinned host arrays for the numpy boundary.

``spectrum_db`` / ``fft_c64`` / ``stft_db`` accept any numpy array; pageable ones are staged through the library's
pinned slots by a pool of copy threads (about 100 GB/s per process, however many GPUs it drives).  Arrays that live
in pinned memory skip the staging: the copy engines read and write them directly, no host thread touches the bytes,
and the boundary scales with the number of GPUs — SURVEY.md §8(e)'s "host gather via per-GPU D2H into slices of one
pinned array" is ``spectrum_db(batch, devices=[...], out=pinned_empty(...))``.

    x = pinned_empty((B, 4096), np.complex64); x[...] = samples     # or let the producer write into it
    rows = pinned_empty((B, 4096), np.float32)
    spectrum_db(x, out=rows)

``registered(array)`` pins an existing array for the duration of a ``with`` block (page-locking costs about as
much as copying the array once, so it pays for buffers that are reused).
"""
from __future__ import annotations

import contextlib
import ctypes
import weakref

import numpy as np

from . import _ffi
from ._ffi import byref, c_size_t, c_void_p, check, lib


def pinned_empty(shape, dtype=np.float32) -> np.ndarray:
    """An uninitialised C-contiguous array in page-locked host memory (``sdrk_host_alloc``), visible to every GPU.
    The memory is released when the array and every view of it are gone."""
    _ffi.require_device(0)
    dt = np.dtype(dtype)
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(v) for v in shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    ptr = c_void_p()
    check(lib().sdrk_host_alloc(c_size_t(max(nbytes, 1)), byref(ptr)))
    buf = (ctypes.c_ubyte * max(nbytes, 1)).from_address(ptr.value)
    weakref.finalize(buf, lib().sdrk_host_free, c_void_p(ptr.value))
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)


def is_pinned(array: np.ndarray) -> bool:
    """True if the array's bytes lie in memory the library knows to be pinned (C-contiguous arrays only)."""
    a = np.asarray(array)
    if not a.flags.c_contiguous or a.nbytes == 0:
        return False
    return bool(lib().sdrk_host_is_pinned(c_void_p(a.ctypes.data), c_size_t(a.nbytes)))


@contextlib.contextmanager
def registered(array: np.ndarray):
    """Page-lock an existing C-contiguous array for the duration of the block (``sdrk_host_register``); the array
    must not be freed or resized inside it."""
    a = np.asarray(array)
    if not a.flags.c_contiguous or a.nbytes == 0:
        raise ValueError("registered() needs a non-empty C-contiguous array")
    _ffi.require_device(0)
    check(lib().sdrk_host_register(c_void_p(a.ctypes.data), c_size_t(a.nbytes)))
    try:
        yield a
    finally:
        check(lib().sdrk_host_unregister(c_void_p(a.ctypes.data)))
