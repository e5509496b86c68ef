"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding plan and every collective the GPU
bench uses (setup broadcast, digest all-gather, max/sum reductions).  No GPU, no compute kernels."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_ranges_partition():
    from dsp_amd.shard import stream_range, owner_of
    for n in (1, 2, 7, 256, 1024):
        for w in (1, 2, 3, 4, 8):
            if n < w:
                continue
            seen = []
            for r in range(w):
                lo, hi = stream_range(n, r, w)
                assert hi > lo
                seen += list(range(lo, hi))
            assert seen == list(range(n))
            assert owner_of(n - 1, n, w) == w - 1
    assert [stream_range(256, r, 8) for r in range(8)] == [(32 * r, 32 * r + 32) for r in range(8)]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    from dsp_amd.shard import Job, stream_range, taps_fingerprint
    job = Job(backend="gloo")
    taps = np.random.default_rng(7).standard_normal(65536) if rank == 0 else None
    chain, got = job.broadcast_setup("lowpass 1k 0.707 fir_p -t pcm -e double -c 1 filt.raw" if rank == 0 else "", taps)
    n_streams = 7
    lo, hi = stream_range(n_streams, rank, world)
    local = torch.tensor([[s, s * s, 0.5 * s] for s in range(lo, hi)], dtype=torch.float64).reshape(-1, 3)
    dig = job.gather_digests(local, n_streams)
    t = job.max_time(1.0 + rank)
    c = job.sum_count(hi - lo)
    job.barrier()
    q.put((rank, chain, taps_fingerprint(got), dig.tolist(), t, c))
    job.close()


def test_two_process_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from dsp_amd.shard import taps_fingerprint
    fp = taps_fingerprint(np.random.default_rng(7).standard_normal(65536))
    for rank, chain, tfp, dig, t, c in res:
        assert chain.startswith("lowpass 1k 0.707 fir_p")
        assert tfp == fp
        assert dig == [[float(s), float(s * s), 0.5 * s] for s in range(7)]
        assert t == 2.0 and c == 7
