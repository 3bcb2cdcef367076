#!/bin/bash
# How the traversal kernel's duration responds to the waves in flight: one pipe, wf_waves_per_cu swept (the kernel is
# persistent: blocks = CUs x waves_per_cu; 20 fit per CU at 92 VGPRs / 8 KB of LDS).  Usage: bash tools/mlp_sweep.sh [out] [lib]
out=${1:-gpurun_out/mlp_sweep.txt}
: > $out
for w in ${WAVES:-8 12 16 20 24}; do
    CRT_LIB=${2:+$PWD/$2} python bench.py --steps 3 --warmup 1 --no-cpu-baseline --opt wf_pipes=1 --opt wf_waves_per_cu=$w 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('${2:-default} waves_per_cu=$w ms_per_step', j['ms_per_step'], 'trace avg ms', j['roofline']['avg_launch_ms'], 'launches', j['roofline']['launches'], 'frac', j['roofline']['frac'])" >> $out
done
cat $out
