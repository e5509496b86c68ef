#!/bin/bash
# experiments on kernels_fused.hip: the library rebuilt with extra defines ($FZ_DEFS) and the resource usage of the NSEC = 10 instance.
# FZ_DBG=1: the round-4 ablation instances (fused_col_fwd without its recurrence / transform / loads / stores / twiddles, selected at run
# time by DSP_AMD_FUSE_DBG through scripts/exp_fused.py dbg) are NOT part of the product source: they live in scripts/fused_dbg.patch
# (taken against the round-5 kernel; `git log -- scripts/fused_dbg.patch` names the kernel revision it applies to) and are compiled from a
# patched copy with -DFUSE_EXPERIMENTS.
cd /root/repo/dsp_amd/csrc || exit 1
SRC=kernels_fused.hip
if [ "${FZ_DBG:-0}" = 1 ]; then
	cp kernels_fused.hip /tmp/kernels_fused_dbg.hip && patch -s /tmp/kernels_fused_dbg.hip ../../scripts/fused_dbg.patch || exit 1
	cp /tmp/kernels_fused_dbg.hip ./_kernels_fused_dbg.hip; SRC=_kernels_fused_dbg.hip; FZ_DEFS="$FZ_DEFS -DFUSE_EXPERIMENTS"
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -I../../include -I. $FZ_DEFS -c $SRC -o build/kernels_fused.o -Rpass-analysis=kernel-resource-usage > /tmp/fz.log 2>&1
rm -f ./_kernels_fused_dbg.hip
grep -E "error" /tmp/fz.log | head
grep -E "Function Name|VGPRs:|AGPRs|VGPRs Spill|ScratchSize" /tmp/fz.log | sed 's/kernels_fused[_a-z]*.hip:[0-9]*:1: remark: //g; s/\[-Rpass.*//' | paste - - - - - | grep "col_fwdILi10ELi1\|prepassILi10" | cut -c1-200
g++ -shared -fPIC -Wl,-z,nodelete -o ../libdsp_amd.so build/*.o
