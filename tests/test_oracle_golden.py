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


def test_zita_contract_restatements_agree():
    # two restatements of the zita_convolver contract (PARITY UNPINNED: the library is absent): the oracle's direct form in
    # extended precision (oracle/dsp_oracle.c, orc_zita_equiv_*) and the FFT form the full-size GPU test uses
    # (oracle_api.zita_contract): the same float32 values but for a rounding tie in a handful of samples
    import numpy as np
    from oracle_api import Oracle, zita_contract
    if not Oracle.available():
        import pytest
        pytest.skip("liboracle.so not built")
    rng = np.random.Generator(np.random.PCG64(3))
    h = rng.standard_normal(700) * np.exp(-np.arange(700) / 90.0) / 8
    x = rng.uniform(-0.5, 0.5, size=(3000, 2))
    a = Oracle.per_channel("zita_equiv", h, np.vstack([x, np.zeros((699 + 64, 2))]), 64)[64:]
    b = zita_contract(x, h)
    assert a.shape == b.shape
    d = np.abs(a - b)
    assert d.max() <= 1.2e-7 and np.count_nonzero(d) <= 0.01 * d.size, (d.max(), np.count_nonzero(d))
