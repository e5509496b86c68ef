#!/bin/bash
# round 6, second session, last call: smoke(), then the whole GPU suite as ONE process N times on HEAD (one line per run)
N=${1:-3}
out=gpurun_out/r06c_final; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ulimit -c 0
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; echo "smoke rc $? $(tail -1 $out/smoke.log)" | tee $out/summary.txt
for i in $(seq 1 $N); do
  now=$(date +%s)
  timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -x > $out/pytest$i.log 2>&1
  echo "run $i rc $? $(( $(date +%s) - now )) s: $(tail -1 $out/pytest$i.log)" | tee -a $out/summary.txt
done
