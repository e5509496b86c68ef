#!/bin/bash
# round 6, call 2: the registration reproducer with and without registrations; the BAR probe; the suite (a process per module) on the day's changes
mkdir -p gpurun_out/c2
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ulimit -c 0
scripts/ubench/barprobe > gpurun_out/c2/barprobe.txt 2>&1; echo "barprobe rc $?"; cat gpurun_out/c2/barprobe.txt
for rep in 1 2; do
  timeout 300 python scripts/r06_pin_repro.py 150 > gpurun_out/c2/pin1_$rep.txt 2>&1; echo "pin default rep $rep rc $?: $(tail -1 gpurun_out/c2/pin1_$rep.txt | cut -c1-300)"
  DSP_AMD_PLUGIN_PIN=0 timeout 300 python scripts/r06_pin_repro.py 150 > gpurun_out/c2/pin0_$rep.txt 2>&1; echo "pin 0 rep $rep rc $?: $(tail -1 gpurun_out/c2/pin0_$rep.txt | cut -c1-300)"
done
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c2/pytest.log 2>&1; echo "pytest rc $?: $(tail -3 gpurun_out/c2/pytest.log | cut -c1-300)"
