"""CPU, world_size 2 over gloo: the one-process-per-GPU sharding path
(sharding.distributed_spectrum_db).  There is no GPU here, so the per-rank
transform is replaced by the oracle through the documented ``compute`` hook; what
is under test is the frame-range partition, the ragged padding and the host gather."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, nfft, q, local_only=False):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from oracle import cpu_ref
    from sdr_iq_visualizer_amd import sharding, synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = synth.synth_iq(77, 0, n_frames, nfft)
        seen = []

        def compute(frames):
            seen.append(frames.shape[0])
            return cpu_ref.spectrum_db(frames)

        lo, hi = sharding.rank_range(n_frames, rank, world)
        if local_only:
            # the rank holds ONLY its own frames (BASELINE config 4's shape: nobody can hold the whole batch)
            mine = synth.synth_iq(77, lo, hi - lo, nfft)
            out = sharding.distributed_spectrum_db(local_frames=mine, n_frames_total=n_frames, compute=compute, dst=0)
            with pytest.raises(ValueError):
                sharding.distributed_spectrum_db(local_frames=mine[:0] if hi - lo else x[:1], n_frames_total=n_frames + 7,
                                                 compute=compute, dst=0)
        else:
            out = sharding.distributed_spectrum_db(x, compute=compute, dst=0)
        assert seen == ([hi - lo] if hi > lo else [])
        if rank == 0:
            q.put(out)
        else:
            assert out is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,local_only", [(6, False), (5, False), (1, False), (5, True), (8, True)])
def test_two_rank_frame_range_sharding_and_gather(n_frames, local_only):
    import torch.multiprocessing as mp
    from oracle import cpu_ref
    from sdr_iq_visualizer_amd import synth
    nfft, world, port = 256, 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, nfft, q, local_only)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref = cpu_ref.spectrum_db(synth.synth_iq(77, 0, n_frames, nfft))
    assert out.shape == (n_frames, nfft) and out.dtype == np.float32
    assert np.array_equal(out, ref)


def _welch_worker(rank, world, port, total, nfft, hop, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from oracle import cpu_ref
    from sdr_iq_visualizer_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = _welch_stream(total)
        a, b, segs = sharding.welch_piece(total, nfft, hop, rank, world)

        def compute(piece):
            rows = 1 + (piece.shape[0] - nfft) // hop
            assert rows == segs
            return cpu_ref.welch_psd(piece, nfft, 2.4e6, hop=hop), rows

        out = sharding.distributed_welch_psd(x[a:b], nfft, 2.4e6, hop=hop, compute=compute)
        q.put((rank, segs, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _welch_stream(total):
    rng = np.random.default_rng(total)
    return ((rng.standard_normal(total) + 1j * rng.standard_normal(total)) * 3).astype(np.complex64)


@pytest.mark.parametrize("world,total,nfft,hop", [(2, 5000, 256, 256), (2, 5000, 256, 100), (3, 4096, 1024, 512),
                                                  (3, 300, 256, 256), (2, 1000, 64, 200)])
def test_welch_over_ranks_all_reduce_equals_one_stream(world, total, nfft, hop):
    """sharding.distributed_welch_psd over gloo (the per-rank average replaced by the oracle): the ranks' pieces
    (`welch_piece`: contiguous segment ranges, nfft - hop samples of halo) cover every segment of the stream exactly
    once — also when a rank gets none — and the all-reduced PSD, the same on every rank, equals the one-stream
    oracle to float32 rounding."""
    import torch.multiprocessing as mp
    from oracle import cpu_ref
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_welch_worker, args=(r, world, port, total, nfft, hop, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref = cpu_ref.welch_psd(_welch_stream(total), nfft, 2.4e6, hop=hop)
    assert sum(s for _, s, _ in got) == 1 + (total - nfft) // hop
    for _, _, out in got:
        assert out.dtype == np.float32 and out.shape == (nfft,)
        assert np.abs(out - ref).max() <= 2e-7 * ref.max()
        assert np.array_equal(out, got[0][2])
