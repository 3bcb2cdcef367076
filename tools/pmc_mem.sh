#!/bin/bash
# usage: tools/pmc_mem.sh <outdir> <scene> <spp> <calls> [option=value ...] -- the memory path of the pass's kernels by PMC, two counters of
# one block per rocprofv3 pass (more than a block's few counters and the profile cannot be built) (never with other trace domains): texture addresser (TA), vector L1 (TCP), L2 (TCC), GRBM.
out=$1; scene=$2; spp=$3; calls=$4; shift 4
export TMPDIR=/tmp
mkdir -p $out
i=0
for grp in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_BUSY_avr TCC_TAG_STALL_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "pass $i: $grp" >> $out/progress.txt
  timeout -k 5 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python3 tools/prof_workload.py $scene $spp $calls "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed" >> $out/progress.txt; grep -m1 -i "exceeds\|error code" $out/p$i.log >> $out/progress.txt; break; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('$out/p*/*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:44]
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); n[k][row['Counter_Name']] += 1
for k, d in agg.items():
    if 'k_wf_trace' not in k and 'k_wf_shade' not in k and 'k_wf_gen' not in k: continue
    print(k)
    for c, v in sorted(d.items()): print('   %-40s %.5g%s' % (c, v / n[k][c] if c.endswith('_avr') else v, '  (mean over launches)' if c.endswith('_avr') else ''))
    if d.get('TCP_TCC_READ_REQ_sum'): print('   mean L1->L2 read latency: %.0f cycles' % (d['TCP_TCC_READ_REQ_LATENCY_sum'] / d['TCP_TCC_READ_REQ_sum']))
    if d.get('GRBM_GUI_ACTIVE') and d.get('TA_TA_BUSY_sum'): print('   TA busy / (GUI_ACTIVE per XCD x 256 TAs): %.3f' % (d['TA_TA_BUSY_sum'] / (d['GRBM_GUI_ACTIVE'] / 8.0 * 256.0)))
    if d.get('GRBM_GUI_ACTIVE') and d.get('TCC_REQ_sum'): print('   L2 requests per cycle and XCD: %.2f (16 channels)' % (d['TCC_REQ_sum'] / d['GRBM_GUI_ACTIVE']))
PY
