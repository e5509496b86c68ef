"""Drop-in check: the UNMODIFIED reference host (CLI + chain runtime + registry, compiled by oracle/Makefile
from /root/reference into oracle/_ref/dsp_gpu) linked against libdsp_amd.so, run side by side with the stock
reference CLI (oracle/_ref/dsp_ref) on the same files.  Exercises the plugin ABI end to end: init / merge /
channel_offsets / drain_samples / run / drain2 / plot / destroy as the reference host calls them."""
import os
import subprocess

import numpy as np
import pytest

from oracle_api import REF_DIR, rms

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.path.exists(os.path.join(REF_DIR, "dsp_gpu")) and os.path.exists(os.path.join(REF_DIR, "dsp_ref"))),
                                 reason="oracle/_ref/dsp_gpu or dsp_ref not built")]

REF = os.path.join(REF_DIR, "dsp_ref")
GPU = os.path.join(REF_DIR, "dsp_gpu")
BIQ = ("lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 "
       "eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707")


def run_cli(exe, args, **kw):
    r = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, **kw)
    assert r.returncode == 0, f"{exe} {' '.join(args)}\n{r.stderr[-2000:]}"
    return r


def both(tmp_path, in_args, chain, channels_out):
    outs = []
    for exe, tag in ((REF, "ref"), (GPU, "gpu")):
        o = os.path.join(str(tmp_path), f"out_{tag}.raw")
        run_cli(exe, ["-q"] + in_args + ["-o", "-t", "pcm", "-e", "double", o] + chain.split())
        outs.append(np.fromfile(o).reshape(-1, channels_out))
    return outs


def test_config1_sgen_cli(tmp_path):
    # BASELINE config 1: sgen sine 48 kHz 2 ch -> gain -6 -> lowpass 1k 0.707
    ref, gpu = both(tmp_path, ["-t", "sgen", "-r", "48k", "-c", "2", "sine+1"], "gain -6 lowpass 1k 0.707", 2)
    assert ref.shape == gpu.shape == (48000, 2)
    assert rms(ref - gpu) < 1e-13


def test_file_chain_cli(tmp_path):
    rng = np.random.Generator(np.random.PCG64(77))
    x = rng.uniform(-0.3, 0.3, size=(30000, 2))
    xin = os.path.join(str(tmp_path), "in.raw"); x.astype("<f8").tofile(xin)
    h = rng.standard_normal(5000) * np.exp(-np.arange(5000) / 600.0); h = h / np.sqrt(np.sum(h * h)) / 4
    hf = os.path.join(str(tmp_path), "h.raw"); h.astype("<f8").tofile(hf)
    in_args = ["-t", "pcm", "-e", "double", "-r", "48k", "-c", "2", xin]
    for chain, och, tol in [
        (f"gain -3 {BIQ}", 2, 1e-12),
        (f"{BIQ} fir_p -t pcm -e double -c 1 {hf} resample 96k", 2, 1e-11),
        (f"fir -t pcm -e double -c 1 {hf} :0 delay 11S : remix 0,1 1 0", 3, 1e-12),
        ("hilbert -p 1023 :1 gain -2 : resample 44.1k", 2, 1e-11),
        ("lowpass 2k 0.707 lowpass -r 2k 0.707 :0 highpass -r60 30 0.707", 2, 1e-12),   # time-reversed IIR, merged + per channel
        ("st2ms :1 delay -f 0.37S : highshelf 6k 0.7 -2 ms2st crossfeed 700 4.5", 2, 1e-12),   # channel-pair effects + fractional delay
    ]:
        ref, gpu = both(tmp_path, in_args, chain, och)
        assert ref.shape == gpu.shape, (chain, ref.shape, gpu.shape)
        assert rms(ref - gpu) < tol, (chain, rms(ref - gpu))


def test_consecutive_effects_share_one_device_segment(tmp_path):
    # the effects of this library that sit next to each other in the host's chain run as ONE fused device pipeline
    # (one H2D / D2H pair per block): the verbose log says so, and the output is unchanged
    rng = np.random.Generator(np.random.PCG64(79))
    x = rng.uniform(-0.3, 0.3, size=(20000, 2))
    xin = os.path.join(str(tmp_path), "in.raw"); x.astype("<f8").tofile(xin)
    o = os.path.join(str(tmp_path), "o.raw")
    chain = f"gain -3 {BIQ} remix 1 0"
    r = run_cli(GPU, ["-q", "-t", "pcm", "-e", "double", "-r", "48k", "-c", "2", xin, "-o", "-t", "pcm", "-e", "double", o] + chain.split(),
                env=dict(os.environ, DSP_AMD_LOGLEVEL="4"))
    assert "fused into one device segment" in r.stderr, r.stderr[-1500:]
    fused = np.fromfile(o).reshape(-1, 2)
    r2 = run_cli(GPU, ["-q", "-t", "pcm", "-e", "double", "-r", "48k", "-c", "2", xin, "-o", "-t", "pcm", "-e", "double", o] + chain.split(),
                 env=dict(os.environ, DSP_AMD_PLUGIN_NO_FUSE="1"))
    unfused = np.fromfile(o).reshape(-1, 2)
    assert fused.shape == unfused.shape and rms(fused - unfused) < 1e-13
    ro = os.path.join(str(tmp_path), "r.raw")
    run_cli(REF, ["-q", "-t", "pcm", "-e", "double", "-r", "48k", "-c", "2", xin, "-o", "-t", "pcm", "-e", "double", ro] + chain.split())
    assert rms(np.fromfile(ro).reshape(-1, 2) - fused) < 1e-12


def test_bit_exact_class_cli(tmp_path):
    rng = np.random.Generator(np.random.PCG64(78))
    x = rng.uniform(-0.3, 0.3, size=(5000, 4))
    xin = os.path.join(str(tmp_path), "in.raw"); x.astype("<f8").tofile(xin)
    ref, gpu = both(tmp_path, ["-t", "pcm", "-e", "double", "-r", "48k", "-c", "4", xin],
                    "gain -6 :1,3 mult 0.3 : add 0.001 remix 0,1 2 . 1,2,3 :0 delay 37S :2,3 st2ms", 4)
    assert ref.shape == gpu.shape and np.array_equal(ref, gpu)


def test_plot_coefficients_identical():
    # `dsp -p` prints every biquad with %.15e (biquad.h:94-95): a coefficient known-answer test through the CLI
    args = ["-p", "-r", "48k", "-c", "1", "-n"] + "gain -6 lowpass 1k 0.707 eq 400 2.0 1.5 highshelf 8k 6d -3 linkwitz_transform 80 0.9 40 0.5".split()
    a = run_cli(REF, args).stdout
    b = run_cli(GPU, args).stdout
    assert "H0_1(w)=" in a and a == b


def test_host_buffers_through_the_copy_path(tmp_path):
    # no mapped staging: every block of the unmodified CLI goes through copy commands on the host's own (pageable, never registered: round 6) block
    # buffers -- in place, out of place with a rate change, and a drain at the end
    rng = np.random.Generator(np.random.PCG64(79))
    x = rng.uniform(-0.3, 0.3, size=(40000, 2))
    xin = os.path.join(str(tmp_path), "in.raw"); x.astype("<f8").tofile(xin)
    env = dict(os.environ, DSP_AMD_PLUGIN_MAPPED_KB="0", DSP_AMD_LOGLEVEL="4")
    for chain, tol in ((f"gain -3 {BIQ}", 1e-12), ("hilbert -p 1023 :1 gain -2 : resample 44.1k", 1e-11)):
        ro, go = os.path.join(str(tmp_path), "r.raw"), os.path.join(str(tmp_path), "g.raw")
        args = ["-q", "-t", "pcm", "-e", "double", "-r", "48k", "-c", "2", xin, "-o", "-t", "pcm", "-e", "double"]
        run_cli(REF, args + [ro] + chain.split())
        r = run_cli(GPU, args + [go] + chain.split(), env=env)
        assert "registered for DMA" not in r.stderr
        ref, gpu = np.fromfile(ro).reshape(-1, 2), np.fromfile(go).reshape(-1, 2)
        assert ref.shape == gpu.shape, (chain, ref.shape, gpu.shape)
        assert rms(ref - gpu) < tol, (chain, rms(ref - gpu))


@pytest.mark.parametrize("seed", range(24))
def test_random_chains_cli(tmp_path, seed):
    # the generator of tests/test_gpu_fuzz.py through the PLUGIN path: the reference's own chain runtime (parser, merge /
    # optimise, auto-inserted align effects, drain) driving this library's effects, against the stock CLI
    from test_gpu_fuzz import gen_chain
    rng = np.random.Generator(np.random.PCG64(9000 + seed))
    channels = int(rng.choice([1, 2, 2, 3, 4, 6]))
    chain = gen_chain(rng, channels)
    x = rng.uniform(-0.5, 0.5, size=(int(rng.integers(3000, 20000)), channels))
    xin = os.path.join(str(tmp_path), "in.raw"); x.astype("<f8").tofile(xin)
    in_args = ["-b", str(int(rng.choice([512, 2048, 4096]))), "-t", "pcm", "-e", "double", "-r", "48k", "-c", str(channels), xin]
    outs = []
    for exe, tag in ((REF, "ref"), (GPU, "gpu")):
        o = os.path.join(str(tmp_path), f"out_{tag}.raw")
        r = subprocess.run([exe, "-q"] + in_args + ["-o", "-t", "pcm", "-e", "double", o] + chain.split(),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        outs.append((r.returncode, np.fromfile(o) if r.returncode == 0 else None))
    assert outs[0][0] == outs[1][0], (chain, outs[0][0], outs[1][0])
    if outs[0][0] == 0:
        ref, gpu = outs[0][1], outs[1][1]
        assert ref.shape == gpu.shape, (chain, ref.shape, gpu.shape)
        assert rms(ref - gpu) <= 1e-10 * max(rms(ref), 1e-3), (chain, rms(ref - gpu))
