#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2b; mkdir -p $O; cd $R
timeout 400 scripts/ubench/hbmprobe 4 > $O/hbmprobe.txt 2>&1
tail -12 $O/hbmprobe.txt
