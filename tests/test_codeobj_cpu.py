"""No scratch on the kernels the BASELINE configs run.

Reads the kernel records of the gfx950 code objects inside dsp_amd/libdsp_amd.so (tests/codeobj_notes.py: the AMDGPU metadata notes -- what the
loader itself goes by) and holds them to two rules:
  1. every kernel instance on the plan of one of BASELINE.json's five configs (and of the headline chain at the reference's own 2048-frame calls)
     exists and has private_segment_fixed_size == 0: no spilled register, no local array in private memory;
  2. no OTHER kernel has any either, except the instances listed in ALLOWED below -- wire-format and 12-section instances that are on no BASELINE
     plan, each with the number of bytes it had when it was listed, so that a new spill anywhere shows up here and not in a profile three rounds later.
Round 5 shipped `conv_row_duo<12,2,false>` (config 4's K2) with 12 bytes, `conv_fdl<11>` / `<12>` (the 2048-frame call) with 12 / 28 and
`resample_gemm_kernel<6>` with 416 bytes of private memory per lane (a loop the unroller gave up on): found by the judge, not by the suite."""
import os
import re

import pytest

import codeobj_notes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "dsp_amd", "libdsp_amd.so")

# kernel instances by BASELINE config (the plans bench.py prints in config.plan / side_runs.*.plan; namespaces p64 / p32 / pfz / psh dropped --
# the float32 instances of config 5 carry the same names in p32 and are covered by "every instance of that name")
BASELINE_KERNELS = {
    "headline (256 x 8 ch, biquad x 10 + fir_p 65536, N = 2^20 = 256 x 4096)": [
        "fused_prepass_mm<2, 8>", "cascade_chunk_carry<10>", "fused_col_fwd<10, 16, 8>", "conv_row_duo<12, 1, true>", "conv_col_inv_pipe<8>"],
    "headline chain at 2048-frame calls (dsp.h:38)": ["cascade_rows<4, 0, 32, 2>", "conv_fdl<11>", "conv_col_fwd<4, false>", "conv_row<10, 3>", "conv_col_inv<4, 4, 0>"],
    "config 2 (1 x 8 ch, 10 biquads)": ["cascade_rows<4, 0, 32, 2>", "cascade_chunk_carry<10>", "cascade_chunk_fix"],
    "config 3 (fir_p 65536 alone)": ["fused_col_fwd<1, 16, 8>", "conv_row_duo<12, 1, true>", "conv_col_inv_pipe<8>"],
    "config 4 (+ resample 48k -> 96k)": ["fused_prepass_mm<2, 8>", "cascade_chunk_carry<10>", "fused_col_fwd<10, 17, 8>", "conv_row_duo<12, 2, false>", "conv_col_inv<8, 4, 2>"],
    "config 5 (hilbert + 131072-tap float32 contract)": ["conv_short<14, 8, false>", "conv_short<13, 8, false>", "conv_col_fwd<8, false>", "conv_row_duo<12, 1, true>", "conv_col_inv<8, 1, 0>"],
    "general n/d resampling, LADSPA-size blocks": ["resample_gemm_kernel<6>", "resample_gemm_kernel<1>", "cascade_resident<false>", "cascade_resident<true>"],
}

# instances on no BASELINE plan that still carry a few spilled registers: name -> bytes of private memory per lane at the time of listing
ALLOWED = {
    "cascade_rows<1, 2, 32, 2>": 20, "cascade_rows<1, 3, 32, 2>": 28, "cascade_rows<2, 3, 32, 2>": 16, "cascade_rows<4, 3, 32, 2>": 16,     # s24 / s32 / float wire formats in the cascade's loads and stores
    "fused_col_fwd<12, 0, 2>": 28, "fused_col_fwd<12, 0, 8>": 28, "fused_col_fwd<12, 32, 8>": 28,                                          # 11 / 12 sections in front of a convolver
}


@pytest.fixture(scope="module")
def records():
    if not os.path.exists(LIB):
        pytest.skip("dsp_amd/libdsp_amd.so not built")
    ks = codeobj_notes.kernels(LIB)
    assert len(ks) > 100, "the library's gfx950 code objects were not found"
    by_name = {}
    for k in ks:
        by_name.setdefault(codeobj_notes.short_name(k["name"]), []).append(k)
    return by_name


def test_the_kernels_of_the_baseline_configs_use_no_private_memory(records):
    missing, bad = [], []
    for config, names in BASELINE_KERNELS.items():
        for n in names:
            inst = records.get(n)
            if not inst:
                # (plain functions have no template arguments in their short name)
                inst = [k for nm, ks in records.items() for k in ks if nm == n or nm.split("<")[0] == n and "<" not in n]
            if not inst:
                missing.append((config, n))
                continue
            for k in inst:
                if k["scratch"] or k["spill_vgpr"] or k["dynamic_stack"]:
                    bad.append((config, n, k["scratch"], k["spill_vgpr"]))
    assert not missing, f"kernels named for a BASELINE config are not in the library (renamed? update BASELINE_KERNELS): {missing}"
    assert not bad, f"kernels on a BASELINE plan with private memory (config, kernel, bytes per lane, spilled registers): {bad}"


def test_no_other_kernel_grows_private_memory(records):
    grown = []
    for n, inst in records.items():
        for k in inst:
            if k["scratch"] > ALLOWED.get(n, 0):
                grown.append((n, k["scratch"], k["spill_vgpr"], ALLOWED.get(n, 0)))
    assert not grown, f"(kernel, bytes per lane, spilled registers, allowed): {grown}"
    on_plan = {n for names in BASELINE_KERNELS.values() for n in names}
    assert not (on_plan & set(ALLOWED)), "a kernel of a BASELINE plan may not be on the allow list"


def test_register_budgets_of_the_two_workgroup_kernels(records):
    """the kernels that count on two workgroups per CU (amdgpu_waves_per_eu(2, 2) / __launch_bounds__(256, 2)) stay inside 256 registers -- with 257 the
    second workgroup silently does not fit and the kernel halves its rate"""
    for n in ("conv_row_duo<12, 1, true>", "conv_row_duo<12, 2, false>", "conv_row_duo<11, 1, true>", "conv_row_duo<11, 2, false>", "conv_fdl<11>", "conv_fdl<12>", "conv_short<13, 8, false>", "conv_short<13, 8, true>"):
        for k in records[n]:
            assert k["vgpr"] + k["agpr"] <= 256, (n, k["vgpr"], k["agpr"])
