"""Second drop-in consumer: the reference's LADSPA frontend, ladspa_dsp.c, compiled UNMODIFIED
(oracle/Makefile, <ladspa.h> restated in oracle/ladspa_abi/) twice -- with the reference's own effects
(oracle/_ref/ladspa_dsp_ref.so) and with this repo's effects from libdsp_amd.so (oracle/_ref/ladspa_dsp_gpu.so) -- and driven
by tests/ladspa_host.py the way a LADSPA host does: float32 ports, small run() sizes (64 ... 1024 frames), config files
found through LADSPA_DSP_CONFIG_PATH (ladspa_dsp.c:221-248, :316-355)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle_api import REF_DIR

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(REF_DIR, "ladspa_dsp_ref.so")
GPU = os.path.join(REF_DIR, "ladspa_dsp_gpu.so")
needs_builds = pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(GPU)), reason="oracle/_ref/ladspa_dsp_*.so not built")

BIQ = ("lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 "
       "eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707")


def write_configs(d):
    rng = np.random.Generator(np.random.PCG64(4242))
    # the LADSPA build of the reference has no pcm codec (codec.c:76-79, :121-126): the filter comes as a coefs: literal
    h = rng.standard_normal(400) * np.exp(-np.arange(400) / 80.0)
    coefs = ",".join("%.17g" % v for v in h / np.sqrt(np.sum(h * h)) / 4)
    with open(os.path.join(d, "config"), "w") as f:           # label "ladspa_dsp"
        f.write(f"# stereo EQ\ninput_channels=2\noutput_channels=2\nLC_NUMERIC=C\neffects_chain=gain -3 {BIQ}\n")
    with open(os.path.join(d, "config_xover"), "w") as f:     # 2 -> 4 crossover with a delayed low band
        f.write("input_channels=2\noutput_channels=4\n[effects_chain]\n"
                "remix 0 1 0 1\n:0,1 lowpass 2k 0.707 lowpass 2k 0.707 delay 7S\n:2,3 highpass 2k 0.707 highpass 2k 0.707 gain -1.5\n")
    with open(os.path.join(d, "config_conv"), "w") as f:
        f.write("input_channels=2\noutput_channels=2\n"
                f"effects_chain=fir_p coefs:{coefs} :0 delay -f 0.4S : st2ms highshelf 6k 0.7 -2 ms2st\n")
    with open(os.path.join(d, "config_mono"), "w") as f:      # defaults: 1 in, 1 out
        f.write("effects_chain=hilbert -p 255 eq 500 1 2\n")


def host(lib, cfg, label, blocks, fin, fout, fs=48000):
    r = subprocess.run([sys.executable, os.path.join(HERE, "ladspa_host.py"), lib, cfg, label, str(fs), blocks, fin, fout],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, f"{lib} {label}\n{r.stderr[-2000:]}"
    return json.loads(r.stdout.strip().splitlines()[-1]), np.load(fout)


@needs_builds
def test_descriptors_follow_config_files(tmp_path):
    # no GPU needed: the library reads its config files at load time; effects are only built by instantiate()
    cfg = str(tmp_path); write_configs(cfg)
    code = ("import sys, json; sys.path.insert(0, %r); import ladspa_host as L; lib = L.load(sys.argv[1], sys.argv[2]);"
            "print(json.dumps(sorted((L.describe(d) for d in L.descriptors(lib)), key=lambda x: x['label'])))" % HERE)
    got = []
    for lib in (REF, GPU):
        r = subprocess.run([sys.executable, "-c", code, lib, cfg], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        got.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert got[0] == got[1]
    by = {d["label"]: d for d in got[1]}
    assert sorted(by) == ["ladspa_dsp", "ladspa_dsp:conv", "ladspa_dsp:mono", "ladspa_dsp:xover"]
    assert (by["ladspa_dsp:xover"]["inputs"], by["ladspa_dsp:xover"]["outputs"]) == (2, 4)
    assert by["ladspa_dsp:xover"]["names"] == ["Input0", "Input1", "Output0", "Output1", "Output2", "Output3"]
    assert (by["ladspa_dsp:mono"]["inputs"], by["ladspa_dsp:mono"]["outputs"]) == (1, 1)


@needs_builds
@pytest.mark.gpu
@pytest.mark.parametrize("label,n_in,n_out,blocks", [
    ("ladspa_dsp", 2, 2, "256"),
    ("ladspa_dsp", 2, 2, "64,1024,100,1"),
    ("ladspa_dsp:xover", 2, 4, "128,1000"),
    ("ladspa_dsp:conv", 2, 2, "512,64"),
    ("ladspa_dsp:mono", 1, 1, "1024,333"),
])
def test_ladspa_frontend_matches_reference_build(tmp_path, label, n_in, n_out, blocks):
    cfg = str(tmp_path); write_configs(cfg)
    rng = np.random.Generator(np.random.PCG64(99))
    x = rng.uniform(-0.5, 0.5, size=(40000, n_in)).astype(np.float32)
    fin = os.path.join(cfg, "in.npy"); np.save(fin, x)
    info_r, ref = host(REF, cfg, label, blocks, fin, os.path.join(cfg, "ref.npy"))
    info_g, gpu = host(GPU, cfg, label, blocks, fin, os.path.join(cfg, "gpu.npy"))
    assert info_r["plugin"] == info_g["plugin"]
    assert ref.shape == gpu.shape == (40000, n_out) and ref.dtype == gpu.dtype == np.float32
    assert float(np.max(np.abs(ref))) > 1e-3
    # the ports are float32: the two double-precision results (<= 1e-12 apart) may round to neighbouring floats now and then
    diff = np.abs(ref.astype(np.float64) - gpu.astype(np.float64))
    assert float(np.max(diff)) <= 1.5 * np.spacing(np.float32(np.max(np.abs(ref)))), float(np.max(diff))
    assert np.count_nonzero(diff) <= 1e-3 * diff.size, np.count_nonzero(diff)


@needs_builds
@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16))
def test_ladspa_random_chains(tmp_path, seed):
    # the generator of tests/test_gpu_fuzz.py (without rate changes: the LADSPA build has no resample) as plugin configs
    from oracle_api import RefChain
    from test_gpu_fuzz import gen_chain
    rng = np.random.Generator(np.random.PCG64(23000 + seed))
    channels = int(rng.choice([1, 2, 2, 3, 4]))
    while True:
        chain = gen_chain(rng, channels)
        if "resample" not in chain: break
    try:
        n_out = RefChain(chain, 48000, channels).ochannels
    except ValueError:
        pytest.skip("chain refused by the reference")
    cfg = str(tmp_path)
    with open(os.path.join(cfg, "config_fuzz"), "w") as f:
        f.write(f"input_channels={channels}\noutput_channels={n_out}\neffects_chain={chain}\n")
    x = rng.uniform(-0.5, 0.5, size=(int(rng.integers(5000, 20000)), channels)).astype(np.float32)
    fin = os.path.join(cfg, "in.npy"); np.save(fin, x)
    blocks = str(rng.choice(["64", "256,1000", "1024", "128,1,512"]))
    _, ref = host(REF, cfg, "ladspa_dsp:fuzz", blocks, fin, os.path.join(cfg, "ref.npy"))
    _, gpu = host(GPU, cfg, "ladspa_dsp:fuzz", blocks, fin, os.path.join(cfg, "gpu.npy"))
    assert ref.shape == gpu.shape == (x.shape[0], n_out)
    diff = np.abs(ref.astype(np.float64) - gpu.astype(np.float64))
    assert float(np.max(diff)) <= 1.5 * np.spacing(np.float32(max(float(np.max(np.abs(ref))), 1e-3))), (chain[:200], float(np.max(diff)))
    # (no count of differing floats here: in the quiet tails of time-reversed sections 1e-13 absolute is more than half an ulp)
    ref_rms = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    assert float(np.sqrt(np.mean(diff ** 2))) <= 2e-8 * max(ref_rms, 1e-3), (chain[:200], float(np.sqrt(np.mean(diff ** 2))), ref_rms)
