"""Every fast path has a switch (DSP_AMD_*), and every slower path behind a switch is a kernel of its own: this module runs the
battery of tests/fallback_probe.py once per switch, each in a process of its own (the switches are read once per process), and
holds every run to the REAL reference's outputs on the same inputs (oracle/_ref) -- so the one-shot conv_row, resample_kernel,
cascade_fast / cascade_kernel, the de-interleaving pass, copy-command staging ... have driver-visible parity, not just the kernels
the default plan happens to select.  (Round 6 removed 16 switches together with the kernels that had lost their A/B for two rounds:
conv_row_big, cascade_wave, the one-shot K2 at long rows -- what is left here is what a plan can still fall back to.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

import fallback_probe as fp
from oracle_api import RefChain, rms

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")]

SWITCHES = [
    "",                                                       # the default plan
    "DSP_AMD_CASCADE_ROWS=0", "DSP_AMD_CASCADE_ROWS=0 DSP_AMD_CASCADE_FAST=0",
    "DSP_AMD_CASCADE_CHUNKS=0",
    "DSP_AMD_CONV_NO_DIRECT=1", "DSP_AMD_NO_LTI_MERGE=1", "DSP_AMD_NO_FEED=1", "DSP_AMD_CONV_FDL=0",
    "DSP_AMD_RESAMPLE_NO_GEMM=1", "DSP_AMD_RESAMPLE_DIRECT=1",
    "DSP_AMD_NO_WIRE_FUSION=1", "DSP_AMD_PLUGIN_MAPPED_KB=0", "DSP_AMD_ZITA_F64=1", "DSP_AMD_CONV_UPC=0",
    "DSP_AMD_PLUGIN_STAGE=0", "DSP_AMD_PLUGIN_STAGE=0 DSP_AMD_PLUGIN_MAPPED_KB=0",
    "DSP_AMD_FUSE=0", "DSP_AMD_CONV_SHORT=0", "DSP_AMD_CONV_SHORT=14",   # (14: the one-trip convolver's 16384-point window wherever the 8192-point one would do)
]


@pytest.fixture(scope="module")
def reference(tmp_path_factory):
    """the reference's outputs for every (case, stream) the probe keeps"""
    d = tmp_path_factory.mktemp("probe_ref")
    ref = {}
    for name, c in list(fp.CASES.items()) + list(fp.HOST_CASES.items()):
        chain = c["chain"]
        h = fp.filter_of(c)
        if h is not None:
            f = os.path.join(str(d), f"{name}.raw")
            np.asarray(h, dtype="<f8").tofile(f)
            chain = chain.replace("{F}", f)
        x = fp.inputs(name, c)
        if "zita_convolver" in chain:
            # not in the reference build here (libzita-convolver is absent; parity unpinned): the restated contract
            from oracle_api import zita_contract
            for s in c["pick"]:
                ref[f"{name}/{s}"] = zita_contract(x[s], h)
            continue
        if name in fp.HOST_CASES:
            ref[f"{name}/0"] = RefChain(chain, 48000, c["C"]).process(x, block=c["block"])
        else:
            for s in c["pick"]:
                ref[f"{name}/{s}"] = RefChain(chain, 48000, c["C"]).process(x[s], block=2048)
    return ref


_baseline = {}


@pytest.mark.parametrize("switch", SWITCHES)
def test_every_switch_gives_the_reference_outputs(reference, tmp_path, switch):
    env = dict(os.environ)
    for kv in switch.split():
        k, v = kv.split("=")
        env[k] = v
    out = os.path.join(str(tmp_path), "probe.npz")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fallback_probe.py"), out], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, (switch, r.stderr[-3000:])
    got = np.load(out)
    for key, want in reference.items():
        y = got[key]
        name = key.split("/")[0]
        if name == "zita":
            assert y.shape == want.shape and rms(y - want) <= 1e-6 * rms(want), (switch, key, rms(y - want) / rms(want))      # the contract's tolerance
            continue
        if name.startswith("rs") or name == "host_conv":
            # a rate changer hands frames over in other portions than the reference mid-stream; after the drain the totals agree
            assert y.shape == want.shape, (switch, key, y.shape, want.shape)
            tol = 1e-11
        else:
            assert y.shape == want.shape, (switch, key, y.shape, want.shape)
            tol = 1e-12
        if name == "remix":
            assert np.array_equal(y, want), (switch, key)
        else:
            assert rms(y - want) <= tol, (switch, key, rms(y - want))
    # wire formats: every byte and both statistics must be the same whichever kernels did the conversions
    if not switch:
        for k in ("wire/bytes", "wire/stats"):
            _baseline[k] = got[k]
        assert int(got["wire/fused"]) == 3
    elif _baseline:
        if "NO_WIRE_FUSION" in switch or "CASCADE" not in switch:
            # the same cascade kernel did the arithmetic: every byte and both statistics are identical
            assert np.array_equal(got["wire/bytes"], _baseline["wire/bytes"]), switch
            assert np.array_equal(got["wire/stats"], _baseline["wire/stats"]), switch
        else:
            # another cascade kernel: sums rounded in another order (1e-15) -- a sample on a rounding boundary may land one step away
            d = np.abs(got["wire/bytes"].astype(np.int64) - _baseline["wire/bytes"].astype(np.int64))
            assert d.max() <= 1 and np.count_nonzero(d) <= 1e-4 * d.size, (switch, d.max(), np.count_nonzero(d))
            assert np.array_equal(got["wire/stats"][:, 0], _baseline["wire/stats"][:, 0]), switch                  # clip counts
            assert np.allclose(got["wire/stats"][:, 1], _baseline["wire/stats"][:, 1], rtol=1e-9, atol=0), switch   # peaks (fp64, before the conversion: the kernels agree to rounding noise)
        if "NO_WIRE_FUSION" in switch:
            assert int(got["wire/fused"]) == 0
    # the switch did switch something: the plans name the path where the plan text shows it
    plans = {k.split("/")[0]: str(got[k]) for k in got.files if k.endswith("/plan")}
    if "NO_LTI_MERGE" in switch:
        assert "+" not in plans["two_conv"].split("conv[")[1].split(" ")[0]
    assert ("cascade-fused" in plans["fused"]) == ("DSP_AMD_FUSE=0" not in switch and "NO_FEED" not in switch), plans["fused"]
    if "CONV_NO_DIRECT" in switch:
        assert "slab-direct" not in plans["conv1024"]
