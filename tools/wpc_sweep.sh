#!/bin/bash
# Two pipes, traversal waves per CU and pipe swept (headline workload).  Usage: bash tools/wpc_sweep.sh [out]
out=${1:-gpurun_out/wpc_sweep.txt}
: > $out
for w in ${WAVES:-10 12 14 16 20}; do
    python bench.py --steps 4 --warmup 1 --no-cpu-baseline --opt wf_waves_per_cu=$w 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('waves_per_cu=$w ms_per_step', j['ms_per_step'], 'value', j['value'], 'trace avg ms', j['roofline']['avg_launch_ms'], 'frac', j['roofline']['frac'])" >> $out
done
cat $out
