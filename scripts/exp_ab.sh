#!/bin/bash
# usage (GPU box): bash scripts/exp_ab.sh [bench args --] lib1.so lib2.so ...   A/B of experimental builds of the library (scripts/exp_libs/*.so,
# built by hand with -D switches; never committed): each is copied over dsp_amd/libdsp_amd.so of the box's scratch copy in turn
args=""; if echo " $* " | grep -q -- " -- "; then while [ "$1" != "--" ]; do args="$args $1"; shift; done; shift; fi
for lib in "$@"; do
	cp $lib dsp_amd/libdsp_amd.so
	python bench.py --steps 6 --warmup 2 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']
print('$lib'.split('/')[-1].ljust(24), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()})"
done
