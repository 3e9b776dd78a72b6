"""Frame-range sharding of a batch across the GPUs of one node — no collectives.

Every frame is independent at the reference's call site (app/sdr/streamer.py:
119-121: one FFT per hardware buffer, no state carried between frames), so a
batch partitions into contiguous frame ranges, one per GPU, and the only
"exchange" is the host gathering each range's rows into one array.  Two ways to
drive it:

* ``spectrum_db_sharded`` — one process, one Python thread per device (ctypes
  releases the GIL during the C call), each thread writing its slice of a shared
  output array;
* ``distributed_spectrum_db`` — one process per GPU under ``torch.distributed``
  (launched by torchrun; backend "nccl" is RCCL on ROCm, "gloo" on CPU hosts):
  each rank transforms its own range and rank ``dst`` gathers the rows.

RCCL over xGMI would only enter if a cross-frame reduction (Welch mean, max-hold)
were added; the path as the reference defines it has none.
"""
from __future__ import annotations

import threading
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


def shard_ranges(n_frames: int, n_shards: int) -> List[Tuple[int, int]]:
    """Contiguous ``[start, stop)`` ranges covering ``n_frames``; the first
    ``n_frames % n_shards`` shards get one extra frame; empty shards are allowed."""
    if n_shards < 1:
        raise ValueError("n_shards must be >= 1")
    if n_frames < 0:
        raise ValueError("n_frames must be >= 0")
    base, extra = divmod(n_frames, n_shards)
    out, start = [], 0
    for s in range(n_shards):
        stop = start + base + (1 if s < extra else 0)
        out.append((start, stop))
        start = stop
    return out


def rank_range(n_frames: int, rank: int, world_size: int) -> Tuple[int, int]:
    return shard_ranges(n_frames, world_size)[rank]


def spectrum_db_sharded(samples, devices: Sequence[int], *, window=None, eps: float = 1e-12,
                        shift: bool = True, out: Optional[np.ndarray] = None, pin="auto") -> np.ndarray:
    """``spectrum_db`` of a ``(B, N)`` batch split over ``devices`` by frame range.  Each device's thread
    drives its own plan — its own pinned pipeline (sdrk_exec_host) — and writes its rows straight into its
    slice of the one result array (``out`` if given).

    ``pin``: what to do with pageable caller arrays (frames and ``out``).  ``"auto"`` (default): staged through the
    library's pinned slots until reuse of the same arrays has paid for page-locking them, then page-locked for the
    rest of their life (``hostmem.plan_pinning`` holds the arithmetic; a host-bound multi-GPU batch gets one
    ``ResourceWarning`` naming ``pinned_empty``); ``True``: page-lock them for this call only
    (``hostmem.registered``); ``False``: always stage."""
    from . import hostmem
    from .spectrum import SpectrumPlan, _as_c64, _cached_plan

    x = _as_c64(samples)
    if x.ndim != 2:
        raise ValueError("sharding needs a (B, N) batch")
    own = [] if x is samples or (isinstance(samples, np.ndarray) and np.may_share_memory(x, samples)) else [x]
    if out is None:
        out = SpectrumPlan._out_array(None, x.shape, np.float32)
        own.append(out)                      # arrays this call made itself: never counted, never page-locked
    else:
        out = SpectrumPlan._out_array(out, x.shape, np.float32)
    if pin not in ("auto", True, False):
        raise ValueError("pin must be 'auto', True or False")
    if pin is True and x.nbytes:
        import contextlib
        with contextlib.ExitStack() as stack:
            for a in (x, out):
                if not hostmem.is_pinned(a):
                    stack.enter_context(hostmem.registered(a))
            return spectrum_db_sharded(x, devices, window=window, eps=eps, shift=shift, out=out, pin=False)
    if pin == "auto" and x.nbytes:
        hostmem.auto_pin([a for a in (x, out) if not any(a is t for t in own)], len(devices), x.nbytes, temporaries=own)
    ranges = shard_ranges(x.shape[0], len(devices))
    errors: List[BaseException] = []

    def work(dev: int, lo: int, hi: int) -> None:
        try:
            if hi > lo:
                _cached_plan(x.shape[1], window, eps, shift, dev).spectrum_db(x[lo:hi], out=out[lo:hi])
        except BaseException as e:  # surfaced on the calling thread below
            errors.append(e)

    threads = [threading.Thread(target=work, args=(d, lo, hi)) for d, (lo, hi) in zip(devices, ranges)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return out


def stft_db_sharded(iq, nfft: int, hop: int, devices: Sequence[int], *, window=None, eps: float = 1e-12,
                    shift: bool = True) -> np.ndarray:
    """Spectrogram of ONE long stream split by row ranges over ``devices``: device d transforms rows
    ``[lo, hi)`` from the samples ``[lo*hop, (hi-1)*hop + nfft)`` — its range plus an ``nfft - hop``
    halo read from the shared host stream, nothing exchanged between devices."""
    from .spectrum import _as_c64, _cached_plan

    x = _as_c64(iq).reshape(-1)
    rows = 0 if x.shape[0] < nfft else 1 + (x.shape[0] - nfft) // hop
    out = np.empty((rows, nfft), dtype=np.float32)
    errors: List[BaseException] = []

    def work(dev: int, lo: int, hi: int) -> None:
        try:
            if hi > lo:
                seg = x[lo * hop: (hi - 1) * hop + nfft]
                out[lo:hi] = _cached_plan(nfft, window, eps, shift, dev).stft_db(seg, hop)
        except BaseException as e:
            errors.append(e)

    threads = [threading.Thread(target=work, args=(d, lo, hi))
               for d, (lo, hi) in zip(devices, shard_ranges(rows, len(devices)))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return out


def _default_compute(window, eps, shift, device) -> Callable[[np.ndarray], np.ndarray]:
    from .spectrum import _cached_plan

    def compute(frames: np.ndarray) -> np.ndarray:
        return _cached_plan(frames.shape[1], window, eps, shift, device).spectrum_db(frames)

    return compute


def _local_device(rank: int) -> int:
    """The GPU this rank owns: LOCAL_RANK (set by torchrun), else the rank, modulo the visible devices."""
    import os

    import torch
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    d = int(os.environ.get("LOCAL_RANK", rank))
    return d % n if n > 0 else d


def distributed_spectrum_db(samples=None, *, window=None, eps: float = 1e-12, shift: bool = True,
                            device: Optional[int] = None, dst: int = 0, group=None,
                            compute: Optional[Callable[[np.ndarray], np.ndarray]] = None,
                            local_frames=None, n_frames_total: Optional[int] = None):
    """One-process-per-GPU form.  Rank r transforms frames ``rank_range(B, r, W)`` on its own GPU and rank ``dst``
    returns the gathered ``(B, N)`` float32 array (other ranks return ``None``).  Two ways to hand the frames over:

    * ``samples``: every rank passes the same ``(B, N)`` batch (or a lazily-indexable view of it) and takes its
      range from it — convenient for tests and small batches;
    * ``local_frames`` + ``n_frames_total``: every rank passes ONLY its own range, ``(hi - lo, N)`` with
      ``(lo, hi) = rank_range(n_frames_total, rank, world)`` — what a job at BASELINE config 4's size has to do
      (256 GiB of IQ in aggregate: no rank can hold the whole batch).

    The rank's GPU is ``device``, else ``LOCAL_RANK``.  The gather moves host arrays and therefore runs over a CPU
    (gloo) group, also when the job's backend is nccl.  ``compute`` replaces the per-rank transform — the CPU
    test-suite injects the oracle there to exercise the sharding and the gather under gloo without a GPU; the
    product default is the HIP path."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if (samples is None) == (local_frames is None):
        raise ValueError("pass either samples (the whole batch) or local_frames + n_frames_total (this rank's range)")
    if local_frames is not None:
        if n_frames_total is None:
            raise ValueError("local_frames needs n_frames_total")
        n_frames, nfft = int(n_frames_total), int(local_frames.shape[1])
        lo, hi = rank_range(n_frames, rank, world)
        if int(local_frames.shape[0]) != hi - lo:
            raise ValueError(f"rank {rank} of {world} owns frames [{lo}, {hi}) of {n_frames}: expected {hi - lo} local "
                             f"frames, got {int(local_frames.shape[0])}")
    else:
        n_frames, nfft = int(samples.shape[0]), int(samples.shape[1])
        lo, hi = rank_range(n_frames, rank, world)
    use_cuda = dist.get_backend(group) == "nccl"
    if device is None and (compute is None or use_cuda):
        device = _local_device(rank)
    if compute is None:
        compute = _default_compute(window, eps, shift, device)
    mine = np.ascontiguousarray(local_frames if local_frames is not None else samples[lo:hi])
    rows = compute(mine) if hi > lo else np.empty((0, nfft), dtype=np.float32)
    rows = np.ascontiguousarray(rows, dtype=np.float32)

    # Host gather: pad every shard to the longest range so one gather suffices.  The rows are host arrays and
    # the destination is a host array, so the exchange runs over a CPU (gloo) group even when the job's default
    # backend is nccl/RCCL — staging them through HBM only to copy them straight back would add two PCIe trips.
    longest = max(b - a for a, b in shard_ranges(n_frames, world))
    padded = np.zeros((longest, nfft), dtype=np.float32)
    padded[: hi - lo] = rows
    t = torch.from_numpy(padded)
    hgroup = _host_group(group) if use_cuda else group
    if use_cuda and hgroup is None:
        # no CPU group could be made: fall back to device tensors on THIS rank's GPU (never torch's current
        # device, which is cuda:0 in every rank unless the caller ran torch.cuda.set_device)
        t = t.to(torch.device("cuda", int(device)))
        hgroup = group
    bucket = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dst_global = dst if group is None else dist.get_global_rank(group, dst)   # `dst` is a rank of `group`
    dist.gather(t, bucket, dst=dst_global, group=hgroup)
    if rank != dst:
        return None
    out = np.empty((n_frames, nfft), dtype=np.float32)
    for r, (a, b) in enumerate(shard_ranges(n_frames, world)):
        out[a:b] = bucket[r][: b - a].cpu().numpy()
    return out


def welch_piece(total_samples: int, nfft: int, hop: int, rank: int, world: int) -> Tuple[int, int, int]:
    """Which samples rank ``rank`` of ``world`` needs so that the ranks' segments are exactly the segments of the whole
    stream: ``(first_sample, end_sample, n_segments)``.  The ``1 + (total - nfft) // hop`` segments are split into
    contiguous ranges (``shard_ranges``); a rank's piece starts at its first segment and carries the ``nfft - hop``
    samples of overlap its last segment needs ("an N−hop sample halo read, no exchange")."""
    segs = 0 if total_samples < nfft else 1 + (total_samples - nfft) // hop
    lo, hi = shard_ranges(segs, world)[rank]
    if hi <= lo:
        return 0, 0, 0
    return lo * hop, (hi - 1) * hop + nfft, hi - lo


def distributed_welch_psd(piece, nfft: int, sample_rate: float, *, hop: Optional[int] = None, window="hann",
                          shift: bool = True, device: Optional[int] = None, group=None,
                          compute: Optional[Callable[[np.ndarray], np.ndarray]] = None) -> np.ndarray:
    """Averaged periodogram (``welch_psd``: scripts/process_sigmf_data.py:188-189) of one long recording whose samples
    are spread over the ranks — the one place on this path with a real exchange step: every rank averages the
    segments of its own piece on its GPU, then the per-rank sums and segment counts are **all-reduced** (RCCL when
    the job's backend is nccl: ``nfft + 1`` float64 values, latency-bound) and every rank returns the same
    ``(nfft,)`` float32 PSD.  ``piece`` is this rank's samples as ``welch_piece`` cuts them (a rank with no segment
    passes an empty array).  ``compute(piece) -> (psd_of_piece float (nfft,), segments)`` replaces the per-rank
    average — the CPU suite injects the oracle there to exercise the reduction under gloo."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    rank = dist.get_rank(group)
    nfft = int(nfft)
    hop = nfft if hop is None else int(hop)
    if hop < 1:
        raise ValueError("hop must be >= 1")
    x = np.ascontiguousarray(np.asarray(piece).reshape(-1))
    use_cuda = dist.get_backend(group) == "nccl"
    if device is None and (compute is None or use_cuda):
        device = _local_device(rank)
    if compute is None:
        from .spectrum import _cached_plan

        def compute(p):
            rows = 1 + (p.shape[0] - nfft) // hop
            return _cached_plan(nfft, window, 1e-12, shift, device).welch_psd(p, sample_rate, hop), rows
    acc = np.zeros(nfft + 1, dtype=np.float64)
    if x.shape[0] >= nfft:
        psd, rows = compute(x)
        acc[:nfft] = np.asarray(psd, dtype=np.float64) * rows         # back to the sum over this rank's segments
        acc[nfft] = rows
    t = torch.from_numpy(acc)
    if use_cuda:
        t = t.to(torch.device("cuda", int(device)))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    total = t.cpu().numpy()
    if total[nfft] < 1:
        raise ValueError("no rank holds a full segment")
    return (total[:nfft] / total[nfft]).astype(np.float32)


_host_groups: dict = {}


def _host_group(group):
    """A gloo process group with the same ranks as ``group`` (created once, collectively), or None."""
    import torch.distributed as dist
    key = id(group) if group is not None else None
    if key not in _host_groups:
        try:
            ranks = None if group is None else dist.get_process_group_ranks(group)
            _host_groups[key] = dist.new_group(ranks=ranks, backend="gloo")
        except Exception:  # pragma: no cover - backend not available
            _host_groups[key] = None
    return _host_groups[key]
