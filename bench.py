#!/usr/bin/env python3
"""bench.py -- headline benchmark of the effects_chain hot path on MI355X.

Workload (BASELINE.json `metric` / north_star): 256 independent streams x 8 channels @ 48 kHz, chain =
10 biquads (SURVEY.md section 8(d) config-2 argv) + fir_p with a 65536-tap filter (config-3 recipe).
A "step" is one block of --block frames pushed through the chain for every stream, with input and output
resident in HBM ([stream][frame][channel] fp64).  Streams are sharded contiguously over the ranks (one
process per GPU, no data-plane collective), the total number of streams is fixed: strong scaling.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line: metric/value/... + "roofline" (dominant kernel, HIP events on the launch
stream) + "cpu_baseline" (the reference's own CPU path on this box's cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BIQUADS = ("lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 "
           "eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707")
HBM_PEAK = 8.0e12      # B/s, MI355X spec (MI355X_MICROARCH.md); measured copy ceiling reported next to it
B_ALG = 16.0           # algorithmic bytes per input channel-sample for same-rate chains


def make_filter(taps, seed=7):
    # N(0,1) * exp(-n/8000), L2-normalised then / 4, numpy default_rng(7)
    rng = np.random.default_rng(seed)
    h = rng.standard_normal(taps) * np.exp(-np.arange(taps) / 8000.0)
    return h / np.sqrt(np.sum(h * h)) / 4.0


def measured_traffic(kernel, workload_key):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_traffic.json, produced by
    scripts/profile_round.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this same command).
    bench.py cannot run a PMC pass on itself; the figure is only reported when the summary was taken on the same
    workload (streams x channels x block x taps), otherwise null."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload_key") != workload_key or kernel not in d.get("kernels", {}):
            continue
        k = d["kernels"][kernel]
        return k["traffic"], {"file": os.path.relpath(f, ROOT), "fetch_raw": k["fetch_raw"], "fetch_corrected": k["fetch_corrected"],
                              "write": k["write"], "bytes_the_kernel_is_designed_to_move": k["algorithmic_bytes_of_kernel"]}
    return None, None


def cpu_baseline(chain, filt_dir, fs, channels, seconds_target=6.0):
    """The reference's own CPU path (oracle/_ref, built from /root/reference) timed on this box's cores on a bounded
    sample of the same workload: every build that is present -- the reference's own flags (-Os) and -O2, with the
    repository's FFT behind the FFTW3 ABI and, where the box has it, MKL's FFTW3 wrapper -- and the reference's
    non-partitioned `fir` (fir.c:109-149) beside `fir_p`.  `value` is the fastest fir_p build (the strongest baseline);
    falls back to the scalar oracle port when _ref is absent."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        from oracle_api import RefChain, Oracle
    except Exception as e:  # pragma: no cover
        return {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "unavailable", "sample": str(e)}
    block = 2048
    rng = np.random.Generator(np.random.PCG64(1234))
    x = rng.uniform(-0.5, 0.5, size=(block, channels))
    builds = (("_mkl", "-O2", "MKL FFTW3 wrapper (sequential)"), ("_o2", "-O2", "own fp64 FFT (oracle/fftw3_abi)"), ("", "-Os (reference flags)", "own fp64 FFT (oracle/fftw3_abi)"))
    variants = []

    def time_one(L, ch, cores, target, unit, streams=None):
        # calibrate with a short run, then size the sample for ~target seconds; `unit` blocks = one period of the effect's
        # work (the non-partitioned fir transforms once per 65536 frames = 32 blocks)
        ns = streams or cores
        t = L.refh_bench(ch.encode(), filt_dir.encode(), fs, channels, ns, ns, block, 2 * unit, x.ctypes.data)
        if t <= 0:
            return None
        n_blocks = int(max(2 * unit, min(20000, 2 * unit * target / t))) // unit * unit
        t = L.refh_bench(ch.encode(), filt_dir.encode(), fs, channels, ns, ns, block, n_blocks, x.ctypes.data)
        return (ns * n_blocks * block * channels / t / 1e6, n_blocks, t) if t > 0 else None

    for variant, flags, fft in builds:
        if not RefChain.available(variant):
            continue
        try:
            L = RefChain.lib(variant)
        except OSError:
            continue
        cores = L.refh_ncpu()
        for eff in ("fir_p", "fir"):
            if eff == "fir" and variant == "_o2":
                continue          # (one -O2 and the -Os build are enough for the side figure)
            ch = chain if eff == "fir_p" else chain.replace("fir_p ", "fir ")
            if eff == "fir" and ch == chain:
                continue
            r = time_one(L, ch, cores, seconds_target if eff == "fir_p" else seconds_target / 2, 4 if eff == "fir_p" else 32)
            if r:
                variants.append({"effect": eff, "build": f"gcc {flags}", "fft": fft, "value": r[0], "cores": cores,
                                 "sample": f"{cores} streams x {channels} ch x {r[1]} blocks of {block} frames, {r[2]:.1f} s"})
            if eff == "fir_p" and not any("one per three cores" in v["sample"] for v in variants) and cores >= 6:
                # fir_p runs two worker threads beside every chain (fir_p.c:407): one chain per core oversubscribes the box three
                # times; a third as many chains gives every thread a core of its own
                ns = cores // 3
                r = time_one(L, ch, cores, seconds_target / 2, 4, streams=ns)
                if r:
                    variants.append({"effect": eff, "build": f"gcc {flags}", "fft": fft, "value": r[0], "cores": cores,
                                     "sample": f"{ns} streams (one per three cores) x {channels} ch x {r[1]} blocks of {block} frames, {r[2]:.1f} s"})
    main = [v for v in variants if v["effect"] == "fir_p"]
    if main:
        best = max(main, key=lambda v: v["value"])
        return {"value": best["value"], "unit": "Msamples/s", "cores": best["cores"], "kind": "reference",
                "sample": f"{best['sample']} (fir_p runs 2 worker threads beside every chain), reference sources {best['build']}, FFT = {best['fft']}",
                "variants": variants}
    if Oracle.available():
        import oracle_chain
        n = 48000
        xs = rng.uniform(-0.5, 0.5, size=(n, 1))
        filt = np.fromfile(os.path.join(filt_dir, "filt.raw"))
        t0 = time.time()
        oracle_chain.run(chain.replace("filt.raw", "{F}"), xs, fs, filt=filt)
        t = time.time() - t0
        return {"value": n / t / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
                "sample": f"oracle/dsp_oracle.c restatement, 1 stream x 1 ch x {n} frames, scalar, {t:.1f} s"}
    return {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "unavailable", "sample": "oracle not built"}


def parity_of_first_step(chain, filt_dir, fs, picks):
    """Self-check inside the bench run (checker only, like cpu_baseline): the first step's output of a few (stream, channel)
    picks against the reference itself (oracle/_ref, built from /root/reference) fed the same device-generated input, over
    the WHOLE step -- every row and column of the step's transforms.  The chains benched here act on every channel alike,
    so the reference runs them as one-channel chains.  picks: [(stream, channel, x[frames], y[frames_out])]."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        from oracle_api import RefChain
        if not RefChain.available():
            return {"checked": False, "why": "oracle/_ref not built"}
        t0 = time.time()
        worst, worst_rel, n = 0.0, 0.0, 0
        against = "oracle/_ref: the reference's own sources, same input, whole first step"
        for s, c, xin, y in picks:
            if "zita_convolver" in chain:
                # the reference build here has no libzita-convolver (PARITY UNPINNED): the restated contract, hilbert in fp64 first
                from oracle_api import Oracle, zita_contract
                from scipy.signal import fftconvolve
                pre = xin
                if chain.startswith("hilbert -p 4095 "):
                    pre = fftconvolve(xin, Oracle.hilbert_taps(4095))[:xin.shape[0]]
                r = zita_contract(pre.reshape(-1, 1), np.fromfile(os.path.join(filt_dir, "filt.raw")))[:xin.shape[0], 0]
                against = "the restated zita_convolver contract (float32 in / exact convolution / float32 out; parity unpinned: libzita-convolver absent), tolerance 1e-6 of the signal"
            else:
                ref = RefChain(chain, fs, 1, directory=filt_dir)
                parts = [ref.run(xin[p:p + 2048].reshape(-1, 1)) for p in range(0, xin.shape[0], 2048)]
                r = np.concatenate([q for q in parts if q.shape[0]])[:, 0]
                ref.close()
            m = min(r.shape[0], y.shape[0])      # (a rate changer hands frames over in other portions than the reference: common prefix)
            d = r[:m] - y[:m]
            e = float(np.sqrt(np.mean(d * d)))
            worst = max(worst, e)
            worst_rel = max(worst_rel, e / max(float(np.sqrt(np.mean(r[:m] * r[:m]))), 1e-300))
            n = m
        return {"checked": True, "rms": worst, "rms_rel_to_signal": worst_rel, "frames_compared": n,
                "picks": [[int(s), int(c)] for s, c, _, _ in picks], "against": against,
                "seconds": time.time() - t0}
    except Exception as e:  # pragma: no cover
        return {"checked": False, "why": str(e)[:300]}


# BASELINE.json's other configs as presets (parity-test cases; the bench line of record is the default run)
CONFIGS = {
    "2": dict(streams=1, channels=8, block=1 << 20, chain=BIQUADS),                    # 1 stream x 8 ch, 10 biquads
    "3": dict(streams=256, channels=8, block=983040, taps=65536, chain="fir_p -t pcm -e double -c 1 {F}"),
    "4": dict(streams=256, channels=8, block=978944, taps=65536, chain=BIQUADS + " fir_p -t pcm -e double -c 1 {F} resample 96k"),
    # config 5 as BASELINE.json states it: hilbert + a zita_convolver-contract convolution of 131072 taps (float32 in / transforms / out,
    # zita_convolver.cpp:44,53,110; PARITY UNPINNED -- libzita-convolver is absent); "5f" = the same shape with the fp64 fir_p in its place
    "5": dict(streams=1024, channels=2, block=917504, taps=131072, chain="hilbert -p 4095 zita_convolver -t pcm -e double -c 1 {F}"),
    "5f": dict(streams=1024, channels=2, block=917504, taps=131072, chain="hilbert -p 4095 fir_p -t pcm -e double -c 1 {F}"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--block", type=int, default=983040, help="frames per step per stream (default: the hop of a 2^20-point transform for 65536 taps, 15 x 65536: the valid fraction of every transform is 15/16)")
    ap.add_argument("--taps", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-runs", action="store_true", help="skip the side figures (merged-IIR plan) reported next to the headline")
    ap.add_argument("--slab-pad", type=int, default=68, help="frames of padding between the slabs of two streams (input and output buffers): "
                    "at exactly block x C x 8 bytes apart -- a multiple of 4 MiB at the default block -- every stream's frame t sits on the same "
                    "memory channels (0 = contiguous [S][block][C] tensors)")
    ap.add_argument("--chain", default=None, help="override the chain (use {F} for the filter file)")
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="one of BASELINE.json's other configs (sets streams / channels / block / taps / chain); the default run is the headline workload")
    args = ap.parse_args()
    if args.config:
        for k, v in CONFIGS[args.config].items():
            setattr(args, k, v)

    # `python bench.py --gpus N` with no launcher around it starts its own N ranks: the same torch.distributed.run command the driver
    # uses (one process per GPU, rendezvous on 127.0.0.1), with this process only waiting for them.  Under a launcher (WORLD_SIZE set)
    # the world must be what --gpus says: a line that claims N GPUs is never printed by another number of ranks.
    backend = os.environ.get("DSP_AMD_BENCH_BACKEND", "nccl")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket
        import subprocess
        import torch
        n_dev = torch.cuda.device_count()
        if backend == "nccl" and n_dev < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but {n_dev} GPU(s) visible: refusing (one rank per GPU; DSP_AMD_BENCH_BACKEND=gloo is the test switch that lets ranks share a GPU)\n")
            sys.exit(2)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: refusing to print a line for another number of ranks than it claims\n")
        sys.exit(2)

    # every DSP_AMD_* switch that is set is printed with the result (they select between equivalent kernels / plans; the
    # library has no switch that skips work: round 1's DSP_AMD_CASCADE_DEBUG was removed from the kernels)
    env_set = {k: v for k, v in sorted(os.environ.items()) if k.startswith("DSP_AMD_")}

    import torch
    import dsp_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # one rank per GPU; DSP_AMD_BENCH_BACKEND=gloo (a test switch) lets several ranks share the GPUs there are -- the sharding, setup
    # broadcast, digest gathering and timing reductions run as they do over RCCL, on a box with one GPU
    if backend == "nccl" and torch.cuda.device_count() < world:
        sys.stderr.write(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPU(s) visible: one rank per GPU\n")
        sys.exit(2)
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    L = dsp_amd.load_library()
    L.dspamd_set_device(dev_index)
    from dsp_amd.shard import Job, stream_range
    job = Job(backend=backend, device=torch.device("cuda", dev_index))   # "nccl" is RCCL over xGMI on ROCm

    fs, C = 48000, args.channels
    S_total = args.streams
    s_lo, s_hi = stream_range(S_total, rank, world)
    S = s_hi - s_lo
    assert S >= 1, "fewer streams than ranks: replicas only"

    filt_dir = f"/tmp/dsp_amd_bench_{os.getpid()}"
    os.makedirs(filt_dir, exist_ok=True)
    chain_t = args.chain or (BIQUADS + " fir_p -t pcm -e double -c 1 {F}")
    # setup broadcast (the only payload that ever crosses xGMI besides digests and timings): chain text + taps
    chain_t, taps = job.broadcast_setup(chain_t, make_filter(args.taps) if rank == 0 else None)
    np.asarray(taps, dtype="<f8").tofile(os.path.join(filt_dir, "filt.raw"))
    chain = chain_t.replace("{F}", "filt.raw")

    batch = dsp_amd.BatchChain(chain, fs, C, S, args.block, directory=filt_dir)
    plan = batch.plan()
    stream = torch.cuda.current_stream().cuda_stream
    # synthetic sgen input resident in HBM: stream i = sine at (100 + 90 i) Hz, amplitude 1
    xc = [torch.empty((S, args.block, C), dtype=torch.float64, device="cuda") for _ in range(2)]
    for k in range(2):
        L.dspamd_sgen_sine(xc[k].data_ptr(), S, args.block, C, fs, 100.0 + 90.0 * s_lo, 90.0, k * args.block, stream)
    if args.slab_pad > 0:
        # the same slabs, args.slab_pad frames apart: [S][block + pad][C] buffers of which the engine sees [:, :block, :]
        x = []
        for k in range(2):
            buf = torch.zeros((S, args.block + args.slab_pad, C), dtype=torch.float64, device="cuda")
            buf[:, :args.block, :] = xc[k]
            x.append(buf[:, :args.block, :])
        del xc
        out = torch.empty((S, batch.max_out_frames(args.block) + args.slab_pad, batch.ochannels), dtype=torch.float64, device="cuda")
    else:
        x = xc
        out = torch.empty((S, batch.max_out_frames(args.block), batch.ochannels), dtype=torch.float64, device="cuda")

    def barrier():
        torch.cuda.synchronize()
        job.barrier()
        torch.cuda.synchronize()

    # ---- step 0 of the stream, before anything is timed: the position every run shares whatever --warmup / --steps are.
    # The digest is taken here (a reproducible checksum), and two (stream, channel) picks are kept for the parity check
    # against the reference (rank 0, with the CPU baseline leg).
    y0 = batch.run(x[0], out)
    f0 = y0.shape[1]
    dig = torch.empty((S, 3), dtype=torch.float64, device="cuda")
    L.dspamd_digest(out.data_ptr(), S, f0, out.shape[1], batch.ochannels, dig.data_ptr(), stream)
    torch.cuda.synchronize()
    picks = []
    if rank == 0 and world == 1 and not args.no_cpu_baseline and ":" not in chain and "remix" not in chain:
        # four (stream, channel) picks drawn from a seed that is printed with the result (a fixed pair would let a stream-indexing
        # error that spares the corners through); the first and last stream stay among them
        pick_seed = int(os.environ.get("DSP_AMD_BENCH_PICK_SEED", str(int(time.time()) & 0xffff)))
        prng = np.random.default_rng(pick_seed)
        pick_set = {(0, int(prng.integers(C))), (S - 1, int(prng.integers(C)))}
        while len(pick_set) < min(4, S * C):
            pick_set.add((int(prng.integers(S)), int(prng.integers(C))))
        for s_, c_ in sorted(pick_set):
            picks.append((s_lo + s_, c_, x[0][s_, :, c_].cpu().numpy().copy(), y0[s_, :, min(c_, batch.ochannels - 1)].cpu().numpy().copy()))

    for w in range(args.warmup):
        batch.run(x[(w + 1) & 1], out)

    def timed_region():
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            batch.run(x[(k + 1) & 1], out)
        barrier()
        return job.max_time(time.perf_counter() - t0)

    # EXACTLY --steps steps per timed region, barrier + synchronize on both sides, max over ranks.  A region of a few
    # hundred milliseconds sits inside the GPU's clock ramp: regions are repeated until a second of work has gone by (at
    # most 12); the measurement is the MEDIAN region once the first (the ramp) is set aside; all of them are reported.
    regions = [timed_region()]
    while sum(regions) < 1.0 and len(regions) < 12:
        regions.append(timed_region())
    settled = sorted(regions[1:]) if len(regions) > 1 else regions
    elapsed = settled[len(settled) // 2]

    # ---- per-kernel averages with HIP events on the launch stream (same K steps, second pass) ----
    L.dspamd_profile_enable(1)
    for k in range(args.steps):
        batch.run(x[k & 1], out)
    prof = {}
    for line in L.dspamd_profile_collect().decode().splitlines():
        name, ms, cnt = line.split()
        prof[name] = {"total_ms": float(ms), "launches": int(cnt), "avg_ms": float(ms) / max(int(cnt), 1)}
    L.dspamd_profile_enable(0)

    # measured copy ceiling (read + write of 1 GiB)
    nbytes = 1 << 30
    a = torch.empty(nbytes // 8, dtype=torch.float64, device="cuda")
    b = torch.empty_like(a)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    L.dspamd_copy_probe(a.data_ptr(), b.data_ptr(), nbytes, stream)
    ev0.record()
    for _ in range(5):
        L.dspamd_copy_probe(a.data_ptr(), b.data_ptr(), nbytes, stream)
    ev1.record()
    torch.cuda.synchronize()
    copy_gbps = 5 * 2 * nbytes / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
    del a, b

    # digest of step 0 (taken above): keeps the output observable and is the same number for every --warmup / --steps
    dig = job.gather_digests(dig, S_total)       # [S_total, 3]: sum, sum of squares, peak per stream
    # the last timed output too must be finite (observed, so nothing of the timed region can be skipped)
    dlast = torch.empty((S, 3), dtype=torch.float64, device="cuda")
    L.dspamd_digest(out.data_ptr(), S, min(args.block, out.shape[1]), out.shape[1], batch.ochannels, dlast.data_ptr(), stream)
    torch.cuda.synchronize()
    finite = bool(torch.isfinite(dig).all().item()) and bool(torch.isfinite(dlast).all().item())
    total_streams = job.sum_count(S)
    assert total_streams == S_total

    # end of stream (the reference's timed run includes the drain): drain_frames of silence pushed through
    # the chain + the rate changers' hand-over, reported next to the steps, never inside ms_per_step
    barrier()
    t0 = time.perf_counter()
    drained = 0
    while True:
        o = batch.drain(args.block, out)
        if o is None:
            break
        drained += o.shape[1]
    barrier()
    ms_drain = job.max_time(time.perf_counter() - t0) * 1e3

    if rank == 0:
        samples_per_step_total = S_total * C * args.block          # input channel-samples, all ranks
        value = samples_per_step_total * args.steps / elapsed / 1e6
        # dominant kernel = largest total time among the per-launch kernels
        dom = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
        launches_per_step = dom[1]["launches"] / args.steps
        samples_per_launch = S * C * args.block / launches_per_step  # this rank's samples handled by one launch
        b_alg = 24.0 if args.config == "4" else B_ALG      # config 4 writes two output samples per input sample (48k -> 96k)
        achieved = samples_per_launch * b_alg / (dom[1]["avg_ms"] * 1e-3) / 1e9
        chain_frac = (value * 1e6 / world) * b_alg / HBM_PEAK
        traffic, traffic_src = (None, None)
        if args.chain is None:
            traffic, traffic_src = measured_traffic(dom[0], f"{S}x{C}x{args.block}x{args.taps}")
        res = {
            "metric": "Msamples/s (all streams), 256x8ch biquadx10 + fir_p(65536)" if not (args.config or args.chain) else f"Msamples/s (all streams), side run: {'config ' + args.config if args.config else 'custom chain'}",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "inner_repeats": len(regions), "ms_per_step_of_each_region": [r / args.steps * 1e3 for r in regions],
            "ms_drain": ms_drain, "drain_frames": drained, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": ("f32 (float32 transforms: the zita_convolver contract) + f64" if "f32-spectrum" in plan else "f64"), "data": "synthetic (device sgen sine, 100+90*i Hz per stream; seeded random 65536-tap filter)",
            "config": {"workload": (f"{S_total} streams x {C} ch @ {fs} Hz, chain = 10 biquads + fir_p({args.taps} taps), {args.block} frames/step/stream" if not (args.config or args.chain)
                                    else f"{S_total} streams x {C} ch @ {fs} Hz, chain = {chain_t}, {args.block} frames/step/stream"),
                       "streams": S_total, "channels": C, "block_frames": args.block, "taps": args.taps, "slab_pad_frames": args.slab_pad,
                       "parallelism": f"streams sharded {S_total // world}/GPU, no data-plane collective",
                       "plan": plan},
            "roofline": {"bound": "hbm", "kernel": dom[0], "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved * 1e9 / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": dom[1]["avg_ms"], "launches_per_step": launches_per_step,
                         "algorithmic_bytes_per_launch": samples_per_launch * b_alg,
                         "whole_chain_frac_per_gpu": chain_frac, "measured_copy_GBps": copy_gbps,
                         "kernels": {k: {"avg_ms": v["avg_ms"], "launches_per_step": v["launches"] / args.steps} for k, v in prof.items()}},
            "output_finite": finite, "env": env_set,
            "digest": {"of": "step 0 of the stream (same for every --warmup / --steps)", "streams": int(dig.shape[0]), "frames": int(f0),
                       "sum_of_squares": float(dig[:, 1].sum().item()), "peak": float(dig[:, 2].max().item())},
        }
        if world > 1:
            # (N = 1 prints the line it always printed; a multi-rank line also says who ran what and over which communicator)
            res["config"]["ranks"] = [{"rank": r, "streams": list(stream_range(S_total, r, world))} for r in range(world)]
            res["config"]["communicator"] = {"backend": "RCCL (torch.distributed nccl)" if backend == "nccl" else backend, "ranks": world, "communicators": 1,
                                             "collectives": "setup broadcast, barriers around the timed region, digest all-gather, max / sum of timings and counts"}
        if picks:
            res["parity"] = parity_of_first_step(chain, filt_dir, fs, picks)
            res["parity"]["pick_seed"] = pick_seed
        if world == 1 and not (args.config or args.chain) and not args.no_side_runs and not args.no_cpu_baseline:
            # SIDE FIGURE, never `value`: the same chain with the planner's opt-in LTI merge of the sections into the filter
            # (DSP_AMD_MERGE_IIR=1, docs/history.md section 6): no cascade pass at all, exact to the decay criterion (2^-70).  The headline
            # above is measured with the cascade kernel in place, as the workload is defined.
            res["side_runs"] = {}
            try:
                # SIDE FIGURE, never `value`: the same step from wire format to wire format (the reference's file -> file run: s16 in,
                # the chain, TPDF dither at 16 bits + clip + s16 out, dsp.c:685-699) -- conversions in the first / last kernel
                # (dspamd_batch_run_wire) against the same conversions as passes of their own around the fp64 step
                del batch
                wb = dsp_amd.BatchChain(chain, fs, C, S, args.block, directory=filt_dir)
                x16 = torch.zeros((S, args.block + args.slab_pad, C), dtype=torch.int16, device="cuda")
                x16[:, :args.block, :] = (x[0] * 20000.0).round().to(torch.int16)
                och = wb.ochannels
                o16 = torch.empty((S, out.shape[1], och), dtype=torch.int16, device="cuda")
                wstats = torch.zeros((S, 2), dtype=torch.float64, device="cuda")

                def t_of(fn, n=5):
                    for _ in range(2):
                        fn()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(n):
                        fn()
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t0) / n

                dt_f = t_of(lambda: wb.run_wire(x16[:, :args.block, :], "s16", "s16", 16, wstats, o16))
                bits = wb.wire_fused()
                xd = x[1]
                written = [0]

                def separate():
                    # (x16 and x[1] are views of buffers with the same padded layout: one flat conversion over the whole buffer)
                    L.dspamd_pcm_read(2, x16.data_ptr(), xd.data_ptr(), S * (args.block + args.slab_pad) * C, stream)
                    f = wb.run(xd, out).shape[1]
                    L.dspamd_pcm_write(2, out.data_ptr(), out.shape[1], o16.data_ptr(), S, f, och, 16, written[0], wstats.data_ptr(), stream)
                    written[0] += f

                dt_s = t_of(separate)
                res["side_runs"]["file_to_file"] = {
                    "what": "s16 -> chain -> dither(16) + clip + s16, per step; fused = conversions in the first / last kernel (dspamd_batch_run_wire), "
                            "separate = dspamd_pcm_read + the fp64 step + dspamd_pcm_write; not the headline",
                    "fused_ms_per_step": dt_f * 1e3, "separate_ms_per_step": dt_s * 1e3, "fused_bits": bits,
                    "value_fused": S * C * args.block / dt_f / 1e6, "value_separate": S * C * args.block / dt_s / 1e6, "unit": "Msamples/s"}
                del wb, x16, o16
            except Exception as e:  # pragma: no cover
                res["side_runs"]["file_to_file"] = {"error": str(e)[:300]}
            try:
                del x, out
                torch.cuda.empty_cache()
                os.environ["DSP_AMD_MERGE_IIR"] = "1"
                sb = 954368                                        # N = 2^20 less 23 whole rows of history for the merged filter (65536 + 24832 - 1 taps): whole-hop calls take the two-pair first pass
                mb = dsp_amd.BatchChain(chain, fs, C, S, sb, directory=filt_dir)
                os.environ.pop("DSP_AMD_MERGE_IIR")
                mx = torch.zeros((S, sb + args.slab_pad, C), dtype=torch.float64, device="cuda")
                L.dspamd_sgen_sine(mx.data_ptr(), S, sb + args.slab_pad, C, fs, 100.0 + 90.0 * s_lo, 90.0, 0, stream)
                mo = torch.empty((S, sb + args.slab_pad, C), dtype=torch.float64, device="cuda")
                for _ in range(2):
                    mb.run(mx[:, :sb, :], mo)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(6):
                    mb.run(mx[:, :sb, :], mo)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 6
                res["side_runs"]["merged_iir"] = {"what": "sections folded into the filter by the planner (opt-in DSP_AMD_MERGE_IIR=1); not the headline",
                                                   "value": S * C * sb / dt / 1e6, "unit": "Msamples/s", "ms_per_step": dt * 1e3, "block_frames": sb, "plan": mb.plan()}
                del mb, mx, mo
            except Exception as e:  # pragma: no cover
                os.environ.pop("DSP_AMD_MERGE_IIR", None)
                res["side_runs"]["merged_iir"] = {"error": str(e)[:300]}
            # SIDE FIGURES, never `value`: BASELINE.json's other configurations and the headline chain at the reference's own block
            # (2048 frames, dsp.h:38) and at round 1's step (196608), a few steps each, so that these numbers too come from the
            # driver's run of this file (parity of every one of them: tests/test_gpu_conv.py, test_gpu_parity.py, test_gpu_smallcalls.py)
            res["side_runs"]["other_configs"] = {}
            presets = [("headline_block_2048", dict(streams=S_total, channels=C, block=2048, taps=args.taps, chain=None, steps=200)),
                       ("headline_block_16384", dict(streams=S_total, channels=C, block=16384, taps=args.taps, chain=None, steps=60)),
                       ("headline_block_196608", dict(streams=S_total, channels=C, block=196608, taps=args.taps, chain=None, steps=10))]
            presets += [("config_" + k, dict(v, steps=6)) for k, v in sorted(CONFIGS.items()) if k in ("2", "3", "4", "5")]
            for name, cfg in presets:
                try:
                    torch.cuda.empty_cache()
                    ctext = (cfg["chain"] or (BIQUADS + " fir_p -t pcm -e double -c 1 {F}"))
                    if cfg.get("taps") and cfg["taps"] != args.taps:
                        np.asarray(make_filter(cfg["taps"]), dtype="<f8").tofile(os.path.join(filt_dir, f"filt{cfg['taps']}.raw"))
                        ctext = ctext.replace("{F}", f"filt{cfg['taps']}.raw")
                    ctext = ctext.replace("{F}", "filt.raw")
                    Sx, Cx, Bx = cfg["streams"], cfg["channels"], cfg["block"]
                    sb2 = dsp_amd.BatchChain(ctext, fs, Cx, Sx, Bx, directory=filt_dir)
                    xs = torch.zeros((Sx, Bx + args.slab_pad, Cx), dtype=torch.float64, device="cuda")
                    L.dspamd_sgen_sine(xs.data_ptr(), Sx, Bx + args.slab_pad, Cx, fs, 100.0, 90.0, 0, stream)
                    os_ = torch.empty((Sx, sb2.max_out_frames(Bx) + args.slab_pad, sb2.ochannels), dtype=torch.float64, device="cuda")
                    for _ in range(2):
                        sb2.run(xs[:, :Bx, :], os_)
                    torch.cuda.synchronize()
                    # a measurement, not a glance: regions of at least 0.25 s of work each (the step count follows from a first estimate), the
                    # median of three, and the kernels that ran -- so that a number that looks odd says which plan produced it
                    t0 = time.perf_counter()
                    sb2.run(xs[:, :Bx, :], os_)
                    torch.cuda.synchronize()
                    n_steps = max(cfg["steps"], int(0.25 / max(time.perf_counter() - t0, 1e-6)) + 1)
                    reg = []
                    for _ in range(3):
                        t0 = time.perf_counter()
                        for _ in range(n_steps):
                            sb2.run(xs[:, :Bx, :], os_)
                        torch.cuda.synchronize()
                        reg.append((time.perf_counter() - t0) / n_steps)
                    dt = sorted(reg)[1]
                    L.dspamd_profile_enable(1)
                    for _ in range(3):
                        sb2.run(xs[:, :Bx, :], os_)
                    kern = {}
                    for line in L.dspamd_profile_collect().decode().splitlines():
                        kn, ms, cnt = line.split()
                        kern[kn] = {"ms_per_step": float(ms) / 3, "launches_per_step": int(cnt) / 3}
                    L.dspamd_profile_enable(0)
                    b_alg2 = 24.0 if name == "config_4" else B_ALG
                    res["side_runs"]["other_configs"][name] = {
                        "value": Sx * Cx * Bx / dt / 1e6, "unit": "Msamples/s", "ms_per_step": dt * 1e3, "streams": Sx, "channels": Cx, "block_frames": Bx,
                        "steps_per_region": n_steps, "ms_per_step_of_each_region": [r * 1e3 for r in reg], "kernels": kern,
                        "whole_chain_frac": Sx * Cx * Bx / dt * b_alg2 / HBM_PEAK, "finite": bool(torch.isfinite(os_[:, :min(Bx, os_.shape[1] - args.slab_pad), :]).all().item()), "plan": sb2.plan()}
                    del sb2, xs, os_
                except Exception as e:  # pragma: no cover
                    res["side_runs"]["other_configs"][name] = {"error": str(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            # (the reference build on this box has no libzita-convolver: its CPU baseline for a zita_convolver chain is the same chain with
            # the reference's own fp64 fir_p in the convolver's place)
            res["cpu_baseline"] = cpu_baseline(chain.replace("zita_convolver ", "fir_p "), filt_dir, fs, C)
            if "zita_convolver " in chain and res["cpu_baseline"].get("sample"):
                res["cpu_baseline"]["sample"] += "; zita_convolver replaced by the reference's fir_p (libzita-convolver is absent)"
            if res["cpu_baseline"].get("value"):
                res["gpu_over_cpu"] = value / res["cpu_baseline"]["value"]
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    job.close()


if __name__ == "__main__":
    main()
