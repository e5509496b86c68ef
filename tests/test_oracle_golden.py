"""Oracle restatement vs the committed golden vectors (outputs of the real
reference, tests/golden/make_golden.py).  CPU-only; needs no /root/reference."""
import json
import os

import numpy as np
import pytest

from oracle_api import Oracle, rms
import oracle_chain

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "golden.npz"))
META = json.load(open(os.path.join(HERE, "golden", "golden.json")))

pytestmark = pytest.mark.skipif(not Oracle.available(), reason="oracle/liboracle.so not built")

# bit-exact class vs FFT-based class (SURVEY.md section 8(d) parity bar)
EXACT = {"config1", "config2", "gain_sel", "remix", "remix_up", "delay", "fir_direct", "midside", "crossfeed", "delay_frac"}


def noise(frames, ch, seed, amp):
    return np.random.Generator(np.random.PCG64(seed)).uniform(-amp, amp, size=(frames, ch))


def test_biquad_impulse_responses():
    imp = np.zeros((48, 1)); imp[0] = 1.0
    for i, b in enumerate(META["biquads"]):
        y, _ = oracle_chain.run(b, imp, 48000)
        assert np.array_equal(y[:, 0], G["biquad_ir"][i]), b


ORACLE_CASES = [c for c in META["cases"] if c.get("oracle", True)]     # (the 8(f) rows are pinned on the real reference only)


@pytest.mark.parametrize("case", ORACLE_CASES, ids=[c["name"] for c in ORACLE_CASES])
def test_case(case):
    x = noise(case["frames"], case["channels"], case["seed"], case["amp"])
    filt = G[case["name"] + "__filter"] if case["name"] + "__filter" in G else None
    y, ofs = oracle_chain.run(case["chain"], x, case["fs"], filt)
    ref = G[case["name"] + "__out"]
    assert ofs == case["ofs"]
    assert y.shape == ref.shape, (y.shape, ref.shape)
    if case["name"] in EXACT:
        assert np.array_equal(y, ref)
    else:
        assert rms(y - ref) < 1e-15, rms(y - ref)
