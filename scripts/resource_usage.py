#!/usr/bin/env python3
"""Tabulate per-kernel register / LDS / occupancy figures of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys, os
src = sys.argv[1]
d = os.path.dirname(os.path.abspath(src))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{d}", f"-I{d}/../../include", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: (?:Function Name: (\S+)|\s*([A-Za-z \[\]/]+): (\d+))", line)
    if not m: continue
    if m.group(1):
        cur = {"name": subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[m.group(2).strip()] = int(m.group(3))
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'spill':>5s} {'scratch':>7s} {'occ':>4s} {'LDS':>7s}")
for r in rows:
    print(f"{r['name'][:70]:70s} {r.get('VGPRs',0):5d} {r.get('AGPRs',0):5d} {r.get('TotalSGPRs',0):5d} {r.get('VGPRs Spill',0):5d} {r.get('ScratchSize [bytes/lane]',0):7d} {r.get('Occupancy [waves/SIMD]',0):4d} {r.get('LDS Size [bytes/block]',0):7d}")
