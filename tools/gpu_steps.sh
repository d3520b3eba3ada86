#!/bin/bash
# Runs the given steps (each one shell command line) in order on the GPU box, every one under its own timeout and
# with its output in gpurun_out/$OUT/<n>.log; stops at the first step that was killed by its timeout (a hung GPU
# step must not be followed by another), goes on after ordinary failures.  usage: OUT=dir tools/gpu_steps.sh "secs|cmd" ...
OUT=${OUT:-steps}
mkdir -p gpurun_out/$OUT
n=0
for step in "$@"; do
  n=$((n+1))
  secs=${step%%|*}
  cmd=${step#*|}
  echo "=== step $n (${secs}s): $cmd" | tee -a gpurun_out/$OUT/steps.log
  timeout -k 10 "$secs" bash -o pipefail -c "$cmd" > gpurun_out/$OUT/$n.log 2>&1
  rc=$?
  echo "=== step $n rc=$rc" | tee -a gpurun_out/$OUT/steps.log
  tail -5 gpurun_out/$OUT/$n.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $n timed out: stopping" | tee -a gpurun_out/$OUT/steps.log; exit 1; fi
done
exit 0
