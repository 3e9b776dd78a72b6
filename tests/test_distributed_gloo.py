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
