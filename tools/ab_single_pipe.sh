#!/bin/bash
# A/B of tuning builds (tune_*.so in the repo root) on the headline workload, one pipe and two: ms per step and the
# traversal kernel's average launch duration (bench.py's HIP events).  Usage: bash tools/ab_single_pipe.sh [out]
out=${1:-gpurun_out/ab_single_pipe.txt}
: > $out
for lib in "" $(ls tune_*.so 2>/dev/null); do
  for pipes in 1 2; do
    CRT_LIB=${lib:+$PWD/$lib} python bench.py --steps 4 --warmup 1 --no-cpu-baseline --opt wf_pipes=$pipes 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('${lib:-default} pipes=$pipes ms_per_step', j['ms_per_step'], 'value', j['value'], 'trace avg ms', j['roofline']['avg_launch_ms'], 'launches', j['roofline']['launches'], 'frac', j['roofline']['frac'])" >> $out
  done
done
cat $out
