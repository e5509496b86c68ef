"""GPU parity: the HIP path (through the C ABI) vs the committed golden vectors (outputs of the real
reference) and vs the oracle / the real reference on seeded inputs, with block-size sweeps."""
import json
import os

import numpy as np
import pytest

from oracle_api import Oracle, RefChain, rms
import oracle_chain

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "golden.npz"))
META = json.load(open(os.path.join(HERE, "golden", "golden.json")))
CASES = {c["name"]: c for c in META["cases"]}

# bit-exact class (gain / remix / integer delay); everything else <= 1e-6 RMS by contract,
# and in practice ~1e-15 (fp64 throughout)
BITEXACT = {"gain_sel", "remix", "remix_up", "delay", "midside", "fir_direct"}
TOL = 1e-12


def noise(frames, ch, seed, amp=0.5):
    return np.random.Generator(np.random.PCG64(seed)).uniform(-amp, amp, size=(frames, ch))


def write_filter(tmp_path, name):
    key = name + "__filter"
    if key not in G:
        return None
    p = os.path.join(str(tmp_path), name + ".raw")
    np.asarray(G[key], dtype="<f8").tofile(p)
    return p


@pytest.fixture(scope="module")
def amd():
    import dsp_amd
    assert dsp_amd.load_library().dspamd_device_count() >= 1, "no HIP device: GPU tests must fail loudly, not fall back"
    return dsp_amd


@pytest.mark.parametrize("name", list(CASES))
def test_golden_host_chain(amd, tmp_path, name):
    c = CASES[name]
    x = noise(c["frames"], c["channels"], c["seed"], c["amp"])
    f = write_filter(tmp_path, name)
    chain = c["chain"].replace("{F}", f) if f else c["chain"]
    ec = amd.EffectsChain(chain, c["fs"], c["channels"])
    y = ec.process(x, block=c["block"])
    ref = G[name + "__out"]
    assert (ec.ofs, ec.ochannels) == (c["ofs"], c["ochannels"])
    assert y.shape == ref.shape, (y.shape, ref.shape)
    if name in BITEXACT:
        assert np.array_equal(y, ref)
    else:
        assert rms(y - ref) < TOL, rms(y - ref)


def test_biquad_impulse_responses(amd):
    imp = np.zeros((48, 1)); imp[0] = 1.0
    for i, b in enumerate(META["biquads"]):
        y = amd.EffectsChain(b, 48000, 1).process(imp, block=48)
        assert y.shape == (48, 1)
        assert np.abs(y[:, 0] - G["biquad_ir"][i]).max() < 1e-14, b


@pytest.mark.parametrize("block", [1, 63, 64, 65, 1000, 1024, 1025, 2048, 5000, 70000])
def test_block_size_invariance_biquads(amd, block):
    # consecutive run() calls of ANY size form one stream
    chain = CASES["config2"]["chain"]
    n = 9000 if block < 64 else 140001
    if block == 1:
        n = 300
    x = noise(n, 8, 77)
    ref, _ = oracle_chain.run(chain, x, 48000)
    y = amd.EffectsChain(chain, 48000, 8).process(x, block=block)
    assert y.shape == ref.shape
    assert rms(y - ref) < TOL, rms(y - ref)


def test_batch_matches_per_stream(amd):
    import torch
    chain = "gain -6 " + CASES["config2"]["chain"] + " :0,3 add 0.01"
    S, C, N = 5, 8, 4100
    x = np.stack([noise(N, C, 100 + s) for s in range(S)])
    b = amd.BatchChain(chain, 48000, C, S, 2048)
    y = b.process(torch.from_numpy(x).cuda(), 2048).cpu().numpy()
    for s in range(S):
        ref, _ = oracle_chain.run(chain, x[s], 48000)
        assert y[s].shape == ref.shape
        assert rms(y[s] - ref) < TOL


def test_wide_stream_channel_groups(amd):
    # more than 16 channels: the cascade kernel splits a stream into channel groups
    chain = "lowpass 1k 0.707 :3,17,40 eq 400 2.0 6 : highpass 20 0.707"
    x = noise(3000, 41, 5)
    ref, _ = oracle_chain.run(chain, x, 48000)
    y = amd.EffectsChain(chain, 48000, 41).process(x, block=1100)
    assert rms(y - ref) < TOL


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
def test_against_real_reference_chain(amd):
    chain = "gain -4 lowshelf 100 0.8s 6 :1 delay 10S : remix 0,1 1 0 highpass 30 bw4.1"
    x = noise(5000, 2, 9)
    ref = RefChain(chain, 48000, 2).process(x, block=700)
    y = amd.EffectsChain(chain, 48000, 2).process(x, block=700)
    assert y.shape == ref.shape
    assert rms(y - ref) < TOL


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
def test_host_buffer_of_many_pipeline_calls(amd):
    """dspamd_chain_run with a host buffer of several 65536-frame pipeline calls and a ragged rest: page-locked staging buffers
    filled and emptied by the caller and three helper threads (engine.h: PinnedStage, crew_memcpy).  Same samples as the stream
    fed in small blocks, and the real reference's (dsp.c:1295-1454 runs any block size through the same chain)."""
    chain = "lowshelf 100 0.8s 6 eq 1k 1.2 -3 delay 7S gain -2"
    frames, ch = 5 * 65536 + 12345, 8
    x = noise(frames, ch, 21)
    ec = amd.EffectsChain(chain, 48000, ch)
    y = ec.run(x)
    y2 = ec.run(x)                                           # (a second buffer continues the stream)
    small = amd.EffectsChain(chain, 48000, ch)
    z = np.concatenate([small.run(np.concatenate([x, x])[p:p + 50000]) for p in range(0, 2 * frames, 50000)])
    assert y.shape == (frames, ch) and np.max(np.abs(np.concatenate([y, y2]) - z)) < 1e-13           # (other call sizes, other cascade kernels: not bit for bit)
    ref = RefChain(chain, 48000, ch).run(np.concatenate([x, x]))
    assert rms(np.concatenate([y, y2]) - ref) < TOL


def test_host_buffers_of_many_calls_from_two_threads(amd):
    """two host threads, a chain each, buffers of several pipeline calls at the same time: the staging copies' helper threads serve one block at a
    time (the other thread copies on its own, engine.cpp: CopyCrew::copy) -- same samples as each chain run alone"""
    import threading
    chain = "lowshelf 100 0.8s 6 eq 1k 1.2 -3 gain -2"
    frames, ch = 6 * 65536 + 777, 8
    xs = [noise(frames, ch, 50 + i) for i in range(2)]
    alone = []
    for x in xs:
        ec = amd.EffectsChain(chain, 48000, ch)
        alone.append([ec.run(x) for _ in range(3)])
    got = [None, None]
    def work(i):
        ec = amd.EffectsChain(chain, 48000, ch)
        got[i] = [ec.run(xs[i]) for _ in range(3)]
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    for i in range(2):
        for a, b in zip(alone[i], got[i]):
            assert np.array_equal(a, b)


def test_plugin_abi_run(amd):
    """Drive one effect through the reference's plugin surface: init -> run -> destroy (effect.h:24-59)."""
    import ctypes as C
    from dsp_amd.lib import StreamInfo, ssize_t
    L = amd.load_library()
    ei = L.dspamd_get_effect_info(b"lowpass")
    assert ei and ei.contents.effect_number == 7
    si = StreamInfo(48000, 2)
    sel = (C.c_char * 2)(1, 1)
    argv = (C.c_char_p * 3)(b"lowpass", b"1k", b"0.707")
    e = L.biquad_effect_init(ei, C.byref(si), sel, None, 3, argv)
    assert e and e.contents.name == b"lowpass" and e.contents.flags == (1 << 1 | 1 << 3)
    x = noise(3000, 2, 21)
    ibuf = x.copy()
    obuf = np.zeros_like(ibuf)
    fr = ssize_t(1000)
    outs = []
    for p in range(0, 3000, 1000):
        blk = np.ascontiguousarray(ibuf[p:p + 1000])
        r = e.contents.run(e, C.byref(fr), blk.ctypes.data, obuf.ctypes.data)
        assert r == blk.ctypes.data and fr.value == 1000   # in place, returns ibuf (biquad.c:296-315)
        outs.append(blk)
    y = np.concatenate(outs)
    ref, _ = oracle_chain.run("lowpass 1k 0.707", x, 48000)
    assert rms(y - ref) < TOL
    e.contents.destroy(e)
    C.CDLL(None).free(e)


REVERSE_CHAINS = [
    "lowpass -r 1k 0.707",                                    # one conjugate pole pair
    "highpass -r 40 0.707",                                   # poles close to the unit circle: 2^12 comb stages' worth of taps
    "lowpass_1 -r 2k",                                        # a single real pole
    "highshelf -r60 8k 0.7 -3",                               # zeros -> FIR part, explicit threshold
    "lowpass -r 1k 0.707 highpass -r 100 0.707",              # two effects merged into one partial-fraction expansion
    "lowpass -r 1k 0.5 lowpass -r 1k 0.5",                    # repeated poles -> series states
    "lowpass -r 1k 0.5",                                      # Q = 0.5: a double real pole inside one section
    "allpass -r 500 1.0 eq -r 2k 1.5 4",
]


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
@pytest.mark.parametrize("chain", REVERSE_CHAINS)
def test_reverse_iir_vs_real_reference(amd, chain):
    # `biquad -r` (reverse_iir.c): the backend designs the equivalent FIR on the host and runs it on the FFT convolver;
    # the real reference evaluates its comb cascades.  Same stream (host-side alignment included), fp64 rounding apart.
    x = noise(30000, 2, 91, 0.4)
    ref = RefChain(chain, 48000, 2).process(x, block=2048)
    ec = amd.EffectsChain(chain, 48000, 2)
    y = ec.process(x, block=1777)
    assert y.shape == ref.shape, (y.shape, ref.shape)
    assert rms(y - ref) < 1e-12, rms(y - ref)


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
def test_reverse_iir_channel_subset_and_mixed_chain(amd):
    # linear-phase crossover style use: forward + reversed section on one channel only; the host delays the other
    chain = ":0 lowpass 2k 0.707 lowpass -r 2k 0.707 : gain -3"
    x = noise(20000, 3, 92, 0.4)
    ref = RefChain(chain, 48000, 3).process(x, block=4096)
    y = amd.EffectsChain(chain, 48000, 3).process(x, block=1000)
    assert y.shape == ref.shape
    assert np.array_equal(y[:, 1:], ref[:, 1:])          # untouched channels: delayed copies, bit-exact
    assert rms(y - ref) < 1e-12, rms(y - ref)


FRAC_DELAY_CHAINS = [
    ("delay -f 0.3S", 2),                                       # default order 2 on every channel
    ("delay -f1 2.7S", 2),                                      # first-order Thiran all-pass + integer part
    (":0 delay -f 1.25m", 3),                                   # one channel only: 60 samples, exactly integer -> host delay line
    (":0 delay -f 1.26m", 3),                                   # ... and 60.48 samples: all-pass on channel 0, nothing on the others
    ("delay 3S delay -f 0.4S", 2),                              # integer + fractional effects merge (delay.c:127-141)
    (":0 delay -f1 0.2S :1 delay -f2 5.5S : delay -f 0.25S", 2),  # per-channel amounts and orders add up / take the maximum
    ("lowpass 2k 0.7 delay -f 10.5S gain -2", 2),               # the all-pass fuses into the biquad cascade
    (":1 delay -f -2.5S", 2),                                   # negative delay: the host delays the other channel
    ("delay -f3 0.4S", 2),                                      # orders above 2: the reference's Thiran ladder (allpass.h:83-118) ...
    (":0 delay -f5 2.3S : lowpass 3k 0.7", 3),                  # ... here factored into second-order all-pass sections
    ("delay -f8 0.77S", 2),
    (":0 delay -f4 1.5S :1 delay -f7 0.2S", 2),                 # different orders per channel: different numbers of sections
    ("delay -f10 12.5S", 2),
    ("delay -f24 0.3S", 2),
    (":1 delay -f32 7.77S", 2),
    ("delay -f34 0.3S", 2),                                      # round 4: the poles in 113-bit arithmetic (thiran_roots.cpp): every order the reference's parser takes
    ("delay -f41 12.77S", 2),
    ("delay -f50 0.5S", 2),
    (":0 delay -f50 30.05S : gain -1", 3),
]


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
@pytest.mark.parametrize("chain,ch", FRAC_DELAY_CHAINS)
def test_fractional_delay_vs_real_reference(amd, chain, ch):
    # `delay -f[order]` (delay.c:149-204, allpass.h): orders 1 and 2 are biquad-shaped all-pass sections on the device,
    # the integer remainder is the host's alignment delay -- same stream as the reference incl. drain
    x = noise(9000, ch, 93, 0.4)
    r = RefChain(chain, 48000, ch)
    ref_drain = r.drain_frames()
    ref = r.process(x, block=2048)
    ec = amd.EffectsChain(chain, 48000, ch)
    assert ec.drain_frames() == ref_drain
    y = ec.process(x, block=1500)
    assert y.shape == ref.shape, (y.shape, ref.shape)
    assert rms(y - ref) < 1e-13, rms(y - ref)


def test_fractional_delay_orders_beyond_the_parser_refused(amd):
    # delay.c:700-701: fd_ap_n > 0 && fd_ap_n <= 50 -- the parser's refusal, in its words
    with pytest.raises(ValueError, match="parameter out of range: order"):
        amd.EffectsChain("delay -f51 0.5S", 48000, 2)


PAIR_CHAINS = [
    ("st2ms", 2, True),
    ("ms2st", 2, True),
    ("st2ms gain -3 ms2st", 2, True),                            # mid/side processing round trip
    (":1,3 st2ms", 5, True),                                     # a pair inside a wider stream, other channels untouched
    ("st2ms :0 lowpass 3k 0.7 : ms2st", 2, False),               # filtered mid channel: cascade between the two mixes
    ("crossfeed 700 4.5", 2, False),                             # crossfeed.c: direct + low-passed opposite + high-passed own
    (":0,2 crossfeed 1k 3 : gain -1", 4, False),
    ("crossfeed 700 6 crossfeed 500 9", 2, False),
]


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
@pytest.mark.parametrize("chain,ch,exact", PAIR_CHAINS)
def test_channel_pair_effects_vs_real_reference(amd, chain, ch, exact):
    # st2ms / ms2st (st2ms.c:28-54) are single-rounded sums and products: bit-exact.  crossfeed (crossfeed.c:33-50) runs
    # its four first-order sections in the fused cascade kernel between a spreading and a combining mix: fp64 rounding apart.
    x = noise(7000, ch, 94, 0.4)
    x[5, :] = -0.0                                               # signed zeros survive (no `0.0 +` start of the sums)
    ref = RefChain(chain, 48000, ch).process(x, block=2048)
    y = amd.EffectsChain(chain, 48000, ch).process(x, block=1300)
    assert y.shape == ref.shape
    if exact:
        assert np.array_equal(y, ref) and np.array_equal(np.signbit(y), np.signbit(ref))
    else:
        assert rms(y - ref) < 1e-13, rms(y - ref)


def test_channel_pair_effects_need_two_channels(amd):
    with pytest.raises(ValueError, match="input channels must be 2"):
        amd.EffectsChain("st2ms", 48000, 3)
    with pytest.raises(ValueError, match="input channels must be 2"):
        amd.EffectsChain(":0 crossfeed 700 4", 48000, 2)


@pytest.mark.parametrize("S", [80, 136])          # 640 channels: two per wave (cascade_rows<2>); 1088: four per wave (<4>)
@pytest.mark.parametrize("gains", [False, True])
@pytest.mark.parametrize("tail", ["", " fir_p coefs:0.5,0.25,-0.125,0.0625"])
def test_many_streams_rows_cascade(amd, tail, gains, S):
    # > 1024 channels with identical biquad sections on all channels: cascade_rows (a wave = 4 channels, one per DPP row,
    # 32 frames per lane).  Block sizes exercise whole tiles (512 frames), the generic remainder kernel and state carried
    # from call to call; with a convolver behind it the results leave through the pair ring (row-ordered stores).
    import torch
    # gains among the sections: folded into the next section's b coefficients / applied to the finished tile
    chain = ("gain -3 lowpass 3k 0.707 highshelf 8k 0.7 -3 mult 1.25 eq 300 1.5 4 eq 1200 2.0 -2.5 highpass 30 0.707 gain 2 mult -0.9" if gains else
             "lowpass 3k 0.707 highshelf 8k 0.7 -3 eq 300 1.5 4 eq 1200 2.0 -2.5 highpass 30 0.707") + tail
    C, N = 8, 11000
    rng = np.random.Generator(np.random.PCG64(4242))
    x = rng.uniform(-0.5, 0.5, size=(S, N, C))
    b = amd.BatchChain(chain, 48000, C, S, 8192)            # 16 tiles per call: 5 waves per group, 3-4 tiles each; then 2808 frames
    assert ("T=4" in b.plan()) == bool(tail)
    y = b.process(torch.from_numpy(x).cuda(), 8192).cpu().numpy()
    for s in (0, 1, 63, 64, S - 1):
        ref, _ = oracle_chain.run(chain, x[s], 48000)
        assert y[s].shape == ref.shape
        assert rms(y[s] - ref) < TOL, (s, rms(y[s] - ref))
    # identical runs must agree bit for bit (regression: 128-bit buffer stores whose data registers were re-used right
    # behind them lost the low dword of single samples, differently from run to run -- see rw_store_b128)
    y2 = amd.BatchChain(chain, 48000, C, S, 8192).process(torch.from_numpy(x).cuda(), 8192).cpu().numpy()
    assert np.array_equal(y, y2)


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
def test_effects_files_vs_real_reference(amd, tmp_path):
    # `@file` sources (effects_chain.c:336-372): comments, quoting, a nested file in a sub-directory with paths relative to
    # ITS directory, %r / %c substitution in file names (util.c:276-343), the active channel selection carried into the file
    d = tmp_path / "fx"
    (d / "sub").mkdir(parents=True)
    h = np.array([0.5, -0.25, 0.125, 0.0625, -0.03125] * 8)
    h.astype("<f8").tofile(str(d / "sub" / "h_48000.raw"))
    (d / "sub" / "inner.fx").write_text("# inner file\nfir_p -t pcm -e double -c 1 h_%r.raw   # relative to sub/\n\"gain\" -1.5\n")
    (d / "main_2.fx").write_text("lowpass 2k 0.7\n:0 @sub/inner.fx\n: highshelf 5k 0.7 -2 # trailing comment\n")
    chain = "gain -3 :0,1 @main_%c.fx : delay 3S"
    x = noise(6000, 3, 95, 0.4)
    ref = RefChain(chain, 48000, 3, directory=str(d)).process(x, block=2048)
    y = amd.EffectsChain(chain, 48000, 3, directory=str(d)).process(x, block=1500)
    assert y.shape == ref.shape
    assert rms(y - ref) < 1e-13, rms(y - ref)
    with pytest.raises(ValueError, match="failed to load effects file"):
        amd.EffectsChain("@nope.fx", 48000, 2, directory=str(d))


@pytest.mark.parametrize("S,C,N,gains", [(1, 8, 196608, False), (1, 8, 98304, True), (3, 2, 65536, False), (1, 1, 40960, True)])
def test_few_channels_chunked_cascade(amd, S, C, N, gains):
    # few channels, long calls (BASELINE config 2: ONE 8-channel stream): the time axis is cut into K chunks that run as
    # independent zero-state streams, then cascade_chunk_carry / cascade_chunk_fix put the carried states back
    # (kernels_chunk.hip).  Two calls: the state at the end of the first is the start of the second.
    import torch
    chain = ("gain -3 lowpass 3k 0.707 highshelf 8k 0.7 -3 mult 1.25 eq 300 1.5 4 eq 60 2.0 -2.5 highpass 30 0.707 gain 2" if gains else
             "lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707")
    rng = np.random.Generator(np.random.PCG64(515))
    x = rng.uniform(-0.5, 0.5, size=(S, 2 * N, C))
    b = amd.BatchChain(chain, 48000, C, S, N)
    L = amd.load_library()
    L.dspamd_profile_enable(1)
    y = b.process(torch.from_numpy(x).cuda(), N).cpu().numpy()
    names = {ln.split()[0] for ln in L.dspamd_profile_collect().decode().splitlines()}
    L.dspamd_profile_enable(0)
    assert "cascade_chunk_fix" in names and "cascade_chunk_carry" in names, names
    for s in range(S):
        ref, _ = oracle_chain.run(chain, x[s], 48000)
        assert y[s].shape == ref.shape
        assert rms(y[s] - ref) < TOL, (s, rms(y[s] - ref))


def test_chunked_cascade_not_used_with_add(amd):
    # `add` is not linear in the state: the ordinary kernels run
    import torch
    chain = "lowpass 1k 0.707 add 0.001 eq 300 1.5 4"
    x = np.random.Generator(np.random.PCG64(516)).uniform(-0.5, 0.5, size=(1, 131072, 2))
    b = amd.BatchChain(chain, 48000, 2, 1, 65536)
    L = amd.load_library()
    L.dspamd_profile_enable(1)
    y = b.process(torch.from_numpy(x).cuda(), 65536).cpu().numpy()
    names = {ln.split()[0] for ln in L.dspamd_profile_collect().decode().splitlines()}
    L.dspamd_profile_enable(0)
    assert "cascade_chunk_fix" not in names, names
    ref, _ = oracle_chain.run(chain, x[0], 48000)
    assert rms(y[0] - ref) < TOL
