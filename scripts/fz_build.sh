#!/bin/bash
# experiments on kernels_fused.hip: the library with the debug instances compiled in (-DFUSE_EXPERIMENTS), resource usage of the NSEC = 10 instance
cd /root/repo/dsp_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -I../../include -I. $FZ_DEFS -c kernels_fused.hip -o build/kernels_fused.o -Rpass-analysis=kernel-resource-usage > /tmp/fz.log 2>&1
grep -E "error" /tmp/fz.log | head
grep -E "Function Name|VGPRs:|AGPRs|VGPRs Spill|ScratchSize" /tmp/fz.log | sed 's/kernels_fused.hip:[0-9]*:1: remark: //g; s/\[-Rpass.*//' | paste - - - - - | grep "col_fwdILi10ELi1ELi0\|prepassILi10" | cut -c1-200
g++ -shared -fPIC -o ../libdsp_amd.so build/*.o
