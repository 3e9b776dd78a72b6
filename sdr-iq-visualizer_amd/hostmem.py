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

``pin="auto"`` (the default of ``spectrum_db(..., devices=[...])``) applies that arithmetic for the caller
(``plan_pinning``): a pageable array is staged while staging is what costs least, page-locked for the rest of its
life once the calls that have reused it would have paid for the page-locking (the ski-rental rule; the registration is
undone by a finaliser on the array that owns the memory), and a batch that the host CPUs — not the links — bound on
several GPUs gets ONE warning that names ``pinned_empty``.
"""
from __future__ import annotations

import contextlib
import ctypes
import dataclasses
import os
import threading
import warnings
import weakref
from typing import Optional

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


# ---- pin="auto": when is page-locking a caller's array worth it? -------------------------------------------------------

GIB = float(1 << 30)


@dataclasses.dataclass(frozen=True)
class HostCosts:
    """Measured on the MI355X box (profiles/r03/pcie_probe.log, bench.py secondary.numpy_boundary); per GiB of INPUT
    (complex64 frames; the float32 rows coming back are half as many bytes and included)."""
    link_GBps: float = 53.0              # one GPU's upstream rate with pinned arrays, rows coming down meanwhile
    staged_link_GBps: float = 45.0       # the same through the pinned staging slots when the host keeps up
    stage_cpu_ms_per_GiB: float = 160.0  # CPU time of the staging copies in + out (125-190 measured), any thread
    register_ms_per_GiB: float = 87.0    # hipHostRegister of the input AND its rows (1.5 GiB): 86-89 ms, one thread
    pool_threads: int = 8                # copy threads of the process-wide pool (7 helpers + the caller)


@dataclasses.dataclass(frozen=True)
class PinDecision:
    mode: str                 # "as-is" (nothing pageable), "stage", "register"
    staged_ms: float          # estimate of this call with staging
    pinned_ms: float          # estimate of this call from pinned arrays
    register_ms: float        # one-time cost of page-locking what is pageable
    host_bound: bool          # staging limited by the host CPUs rather than by the links
    warn: bool                # tell the caller about pinned_empty (several GPUs, large batch, host-bound)
    reason: str


def usable_cpus() -> int:
    """CPUs this process may run on at once (affinity mask capped by the cgroup quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def plan_pinning(n_devices: int, in_bytes: int, pageable_fraction: float, sightings: int, cores: int,
                 costs: HostCosts = HostCosts()) -> PinDecision:
    """The decision behind ``pin="auto"``, as arithmetic on measured rates (no GPU needed to evaluate it).

    ``pageable_fraction``: share of the call's host bytes (frames + rows) that is NOT already pinned;
    ``sightings``: how many auto-mode calls have handed in these same buffers so far, this one included.

      staged   = max( in / n_devices / staged_link ,  stage_cpu * in * pageable / min(cores, pool_threads) )
      pinned   = in / n_devices / link
      register = register_cost * in * pageable                       (once; undone when the array dies)

    With a few copy threads one call does not pay for its own registration (87 ms per GiB against 160 / threads), so a
    first sighting is staged (on a single core it is registered at once).  Reused buffers are registered once the time staging has cost beyond the pinned form, summed over the
    sightings, reaches the registration's price — after that every further call is pure gain, and the total is at
    most twice what the best choice in hindsight would have cost."""
    gib = in_bytes / GIB
    workers = max(1, min(cores, costs.pool_threads))
    transfer_staged = in_bytes / max(n_devices, 1) / (costs.staged_link_GBps * 1e9) * 1e3
    cpu_ms = costs.stage_cpu_ms_per_GiB * gib * pageable_fraction / workers
    staged = max(transfer_staged, cpu_ms)
    pinned = in_bytes / max(n_devices, 1) / (costs.link_GBps * 1e9) * 1e3
    register = costs.register_ms_per_GiB * gib * pageable_fraction
    host_bound = cpu_ms > 1.25 * transfer_staged
    if pageable_fraction <= 0.0 or in_bytes == 0:
        return PinDecision("as-is", pinned, pinned, 0.0, False, False, "every array of the call is already pinned")
    gain = staged - pinned
    if gain > 0 and sightings * gain >= register:
        return PinDecision("register", staged, pinned, register, host_bound, False,
                           f"seen {sightings} times: {sightings} x {gain:.1f} ms of staging overhead >= {register:.0f} ms of page-locking")
    warn = host_bound and n_devices >= 4 and in_bytes >= (1 << 30)
    return PinDecision("stage", staged, pinned, register, host_bound, warn,
                       f"staging costs {gain:.1f} ms more than pinned arrays per call, page-locking {register:.0f} ms once")


# key of a call's arrays -> (count, weak references to the arrays that OWN the memory).  An entry counts only while those
# very owners are alive: a fresh array that the allocator happens to put at a recycled address starts from zero.
_sightings: dict = {}
_auto_registered: dict = {}          # base address -> bytes, page-locked by _register_for_life
# base address -> weak reference to the owning array whose registration hipHostRegister refused (overlap with a user registration
# ...).  Tied to that very array: another array the allocator later puts at the same address is tried afresh.
_not_registrable: dict = {}
# Re-entrant: a finaliser (undo below) that the cyclic collector runs while this thread is inside a locked region must
# not deadlock; the finaliser itself does not take the lock at all (dict.pop is atomic under the GIL).
_auto_lock = threading.RLock()
_warned = False


def _owner(a: np.ndarray):
    """The object that owns the array's memory (end of the .base chain)."""
    o = a
    while isinstance(o, np.ndarray) and o.base is not None:
        o = o.base
    return o


def _count_sighting(arrays) -> int:
    """How many auto-mode calls, this one included, have handed in THESE arrays (same address, same size, same owning
    objects still alive).  Arrays whose owner cannot be weakly referenced never accumulate."""
    key = tuple((a.ctypes.data, a.nbytes) for a in arrays)
    owners = [_owner(a) for a in arrays]
    try:
        refs = tuple(weakref.ref(o) for o in owners)
    except TypeError:
        return 1
    with _auto_lock:
        if len(_sightings) > 64:
            _sightings.clear()
        count, old = _sightings.get(key, (0, ()))
        if len(old) != len(owners) or any(r() is not o for r, o in zip(old, owners)):
            count = 0                                   # another array at a recycled address, or the first sighting
        _sightings[key] = (count + 1, refs)
        return count + 1


def _register_for_life(a: np.ndarray) -> bool:
    """Page-lock the whole allocation ``a`` lives in and undo it when the owning array is collected.  False (nothing
    done, the caller stages) when the owner is not a numpy array that owns its data, cannot carry a finaliser, or the
    registration is refused (e.g. the range overlaps one the user registered)."""
    root = _owner(a)
    if not isinstance(root, np.ndarray) or not root.flags.owndata or not root.flags.c_contiguous or root.nbytes == 0:
        return False
    if root.nbytes > 2 * a.nbytes:          # a small window of a large allocation: page-locking all of it is not what was asked
        return False
    ptr, nbytes = root.ctypes.data, root.nbytes
    with _auto_lock:
        if ptr in _auto_registered:
            return True
        refused = _not_registrable.get(ptr)
        if refused is not None:
            if refused() is root:
                return False
            del _not_registrable[ptr]           # the refused array is gone; this is another one at a recycled address
        try:
            check(lib().sdrk_host_register(c_void_p(ptr), c_size_t(nbytes)))
        except (ValueError, MemoryError, _ffi.SdrkError):
            if len(_not_registrable) > 256:
                _not_registrable.clear()
            try:
                _not_registrable[ptr] = weakref.ref(root)
            except TypeError:                   # (cannot happen for an ndarray; without a reference nothing is remembered)
                pass
            return False
        _auto_registered[ptr] = nbytes

    def undo(p=ptr):
        _auto_registered.pop(p, None)        # no lock: may run from the collector inside a locked region of any thread
        _not_registrable.pop(p, None)
        try:
            lib().sdrk_host_unregister(c_void_p(p))
        except Exception:                   # noqa: BLE001 - interpreter shutdown
            pass

    weakref.finalize(root, undo)             # runs before numpy frees the memory (weak references are cleared first)
    return True


COSTS = HostCosts()          # what pin="auto" reckons with (a deployment on other hardware may replace it)


def auto_pin(arrays, n_devices: int, in_bytes: int, costs: Optional[HostCosts] = None, cores: Optional[int] = None,
             temporaries=()) -> PinDecision:
    """Apply ``plan_pinning`` to the host arrays of one call (frames and rows): count the sighting, register what the
    decision says, warn once where the host bounds a multi-GPU call.  Returns the decision (tests read it).

    ``temporaries``: arrays of the call that the library made itself (a result for ``out=None``, a complex64 copy of
    other input).  They weigh in the call's pageable share but are never counted as seen again and never page-locked:
    a fresh array per call would pay ``hipHostRegister`` (and the page faults it forces) every time, for nothing."""
    global _warned
    arrays = [a for a in arrays if a is not None and a.nbytes]
    temporaries = [a for a in temporaries if a is not None and a.nbytes]
    total = sum(a.nbytes for a in arrays) + sum(a.nbytes for a in temporaries)
    pageable = [a for a in arrays if not is_pinned(a)]
    frac = ((sum(a.nbytes for a in pageable) + sum(a.nbytes for a in temporaries)) / total) if total else 0.0
    n = _count_sighting(arrays) if arrays and not temporaries else 1
    d = plan_pinning(n_devices, in_bytes, frac, n, usable_cpus() if cores is None else cores, COSTS if costs is None else costs)
    if d.mode == "register" and temporaries:
        d = dataclasses.replace(d, mode="stage", reason=d.reason + "; but the call owns a temporary array (fresh every call): staged")
    if d.mode == "register":
        done = [_register_for_life(a) for a in pageable]
        if not all(done):
            d = dataclasses.replace(d, mode="stage", reason=d.reason + "; but an array's memory cannot be page-locked "
                                    "(not owned by a numpy array, or the registration was refused): staged")
    elif d.warn and not _warned:
        _warned = True
        warnings.warn(
            f"spectrum_db over {n_devices} GPUs from pageable arrays is bound by the host's staging copies "
            f"(~{d.staged_ms:.0f} ms per call here against ~{d.pinned_ms:.0f} ms from pinned arrays): allocate the frames and "
            "the result with sdr_iq_visualizer_amd.pinned_empty(...), or keep handing in the same arrays (they are "
            "page-locked automatically once that has paid for itself)", ResourceWarning, stacklevel=3)
    return d
