"""Small plugin blocks through the resident wave (kernels_resident.hip, round 5): a device segment that is one cascade serves blocks of up to 1024 frames
without a launch per block.  Through the plugin vtable as a host drives it (the reference's chain runtime linked over this library: oracle/_ref/
libdspref_gpu.so), against the all-reference runtime on the same input: mixed block sizes (resident and ordinary paths alternate on the same states),
gains and adds among the sections, selectors (ops that skip channels), eight channels (two waves), a pause longer than the wave's lifetime."""
import time

import numpy as np
import pytest

from oracle_api import RefChain, rms

pytestmark = pytest.mark.gpu

BIQ = ("lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 "
       "eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707")


@pytest.fixture(scope="module", autouse=True)
def hip_runtime():
    # libdspref_gpu.so pulls in libdsp_amd.so, which leaves the choice of the HIP runtime to its host: loaded here
    import dsp_amd
    assert dsp_amd.load_library().dspamd_device_count() >= 1


def stream(chain, C, x, blocks, variant, pause_at=None):
    """the chain built and run by the reference's own effects_chain runtime; variant "_gpu": its effects come from libdsp_amd.so"""
    r = RefChain(chain, 48000, C, variant=variant)
    outs, pos, k = [], 0, 0
    while pos < x.shape[0]:
        n = blocks[k % len(blocks)]
        if pause_at is not None and k in pause_at:
            time.sleep(0.02)             # (the wave has left by now: the next block starts another one, which finds the request waiting)
        outs.append(r.run(x[pos:pos + n]))
        pos += n
        k += 1
    r.close()
    return np.concatenate([o for o in outs if o.shape[0]])


@pytest.mark.skipif(not RefChain.available("_gpu"), reason="oracle/_ref/libdspref_gpu.so not present")
@pytest.mark.parametrize("chain,C,blocks,pause", [
    ("gain -3 " + BIQ, 2, (64,), (5, 300)),                              # the LADSPA shape: every block through the resident wave; two pauses; 3000 blocks back to back outlast a wave's 20 ms
    ("gain -3 " + BIQ, 2, (64, 1, 1024, 17, 4096, 256, 3), None),        # resident and ordinary kernels in turn, on the same states
    ("lowpass 2k 0.707 :0 eq 300 1.5 4 gain -2 : add 0.001 highshelf 6k 0.7 2 mult 0.5", 3, (128, 64, 500), None),   # gains, an add, an op that skips channels
    (BIQ, 8, (256, 64), (3,)),                                           # eight channels: two waves
    ("gain -6 mult 1.5 add 0.25", 2, (64, 200), None),                   # no section at all: bit-exact ops only
    # the crossover shape (examples/crossover_lr4_2kHz): a remix 2 -> 4 in front of per-band sections -- the remix inside the wave too
    ("remix 0 1 0 1 :0,1 lowpass 2k 0.707 lowpass 2k 0.707 :2,3 highpass 2k 0.707 highpass 2k 0.707 : gain -1", 2, (64, 128, 1000), (4,)),
    ("remix 0,1 0 1 gain -3 add 0.001", 2, (64, 96), None),              # a mono sum beside the two channels, then bit-exact ops only
    # round 6: direct FIRs (<= 32 taps, fir.c:43-62 / fir_p.c:131-148: bit-exact, history carried in device memory across blocks AND across the two paths)
    ("fir_p coefs:0.9,0.05,-0.02,0.01,0.003 gain -2", 2, (64, 3, 33, 128, 2048, 64), None),
    ("highpass 30 0.707 fir_p coefs:0.9,0.05,-0.02,0.01,0.003 gain -1", 2, (64, 17, 128, 5), (2,)),
    ("fir coefs:0.5,0.25,0.125 :0 eq 300 1.5 4", 2, (64, 96), None),
    # ... the crossover with a correction FIR behind the sections, and FIRs either side of the cascade
    ("remix 0 1 0 1 :0,1 lowpass 2k 0.707 lowpass 2k 0.707 :2,3 highpass 2k 0.707 highpass 2k 0.707 : fir_p coefs:1.0,0.1,-0.05", 2, (64, 128, 700), (3,)),
    ("fir_p coefs:0.8,0.1 eq 1k 1.0 3 fir_p coefs:1.0,-0.2,0.04", 2, (64, 20), None),
    # ... and weighted mixes: mid / side around an equaliser, the two ends of a crossfeed (st2ms.c:28-54, crossfeed.c:41-46)
    ("st2ms eq 1k 1.0 3 ms2st", 2, (64, 100), None),
    ("crossfeed 700 4.5", 2, (64, 128), None),
])
def test_small_blocks_through_the_resident_wave(chain, C, blocks, pause):
    rng = np.random.Generator(np.random.PCG64(99))
    x = rng.uniform(-0.5, 0.5, size=(200000 if blocks == (64,) else 40000, C))
    x[100:110] = 0.0                                                     # (and some exact zeros of both signs: gains keep the sign of zero)
    x[105] = -0.0
    from dsp_amd.lib import plugin_counters
    c0 = plugin_counters()
    got = stream(chain, C, x, blocks, "_gpu", pause)
    c1 = plugin_counters()
    ref = stream(chain, C, x, blocks, "")
    # the wave really served every block it is meant for (a dead wave falls back to launches and gives the same samples: outputs alone prove nothing)
    n_blocks, small, pos, k = 0, 0, 0, 0
    while pos < x.shape[0]:
        n = min(blocks[k % len(blocks)], x.shape[0] - pos)
        small += n <= (128 if "fir" in chain else 256)        # (the wave's limit: 256 frames for one systolic pass and nothing per tap, else 128 -- plugin.cpp Resident::init)
        n_blocks += 1
        pos += n
        k += 1
    d = {key: c1[key] - c0[key] for key in c1}
    assert d["wave_blocks"] == small, (d, small, n_blocks)
    assert d["wave_timeouts"] == 0 and d["wave_off"] == 0, d
    assert d["wave_blocks"] + d["mapped_blocks"] + d["copied_blocks"] == n_blocks, (d, n_blocks)
    assert 1 <= d["wave_launches"] <= small, d
    if pause:
        assert d["wave_launches"] >= 1 + len(pause), d           # (every pause outlasts the wave: the block behind it started another one)
    assert got.shape == ref.shape
    assert got.shape[1] == (3 if chain.startswith("remix 0,1") else 4 if chain.startswith("remix") else C)
    if not any(w in chain for w in ("eq", "pass", "shelf", "crossfeed")):          # no section anywhere: gains, adds, remixes and direct FIRs are bit-exact
        assert np.array_equal(got, ref) and np.array_equal(np.signbit(got), np.signbit(ref))
    else:
        assert rms(got - ref) < 1e-12, rms(got - ref)


_SWITCH_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import dsp_amd
from dsp_amd.lib import plugin_counters
dsp_amd.load_library()
from oracle_api import RefChain, rms
chain = {chain!r}
rng = np.random.Generator(np.random.PCG64(5))
x = rng.uniform(-0.5, 0.5, size=(64 * 300, 2))
def stream(variant):
    r = RefChain(chain, 48000, 2, variant=variant)
    y = np.concatenate([r.run(x[p:p + 64]) for p in range(0, x.shape[0], 64)])
    r.close()
    return y
got, c, ref = stream("_gpu"), plugin_counters(), stream("")
assert got.shape == ref.shape and rms(got - ref) < 1e-12, rms(got - ref)
print("COUNTERS", c["wave_blocks"], c["mapped_blocks"], c["wave_timeouts"], c["wave_off"])
"""


@pytest.mark.skipif(not RefChain.available("_gpu"), reason="oracle/_ref/libdspref_gpu.so not present")
@pytest.mark.parametrize("env,wave", [({"DSP_AMD_PLUGIN_MAILBOX": "host"}, True), ({"DSP_AMD_PLUGIN_RESIDENT": "0"}, False), ({}, True)])
def test_the_mailbox_in_host_memory_and_the_path_without_a_wave(env, wave):
    """the two ways round the default (the switches are read once per process: a process each): the request mailbox in page-locked host memory -- what a
    device without a large BAR gets -- and no resident wave at all (a launch per block); both against the reference, and the counters say which ran"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _SWITCH_SCRIPT.format(root=root, tests=os.path.join(root, "tests"), chain="gain -3 " + BIQ)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DSP_AMD_LOGLEVEL="4", **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    wb, mb, to, off = [int(v) for v in [ln for ln in r.stdout.splitlines() if ln.startswith("COUNTERS")][0].split()[1:]]
    assert to == 0 and off == 0
    if wave:
        assert wb == 300 and mb == 0
        assert ("request mailbox in page-locked host memory" if env else "request mailbox in device memory") in r.stderr, r.stderr[-1500:]
    else:
        assert wb == 0 and mb == 300
