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


def _bench(args, env_extra=None, drop=("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DSP_AMD_BENCH_BACKEND")):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_bench_refuses_more_ranks_than_gpus():
    """`bench.py --gpus N` without a launcher starts its own N ranks -- or refuses (exit 2, no JSON line) when fewer than N GPUs are
    visible: it never prints a line for another number of ranks than it claims."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a box with two GPUs runs the two ranks")
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and "refusing" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_refuses_world_size_mismatch():
    """under a launcher the world must be what --gpus says"""
    r = _bench(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
