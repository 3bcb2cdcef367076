#!/bin/bash
# TA / TCP busy counters for the current workload
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python3 tools/prof_workload.py "$@" > $out/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('$out/p*/*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        agg[row['Kernel_Name'][:40]][row['Counter_Name']] += float(row['Counter_Value'])
for k, d in agg.items():
    if 'wf_' not in k: continue
    print(k)
    for c, v in sorted(d.items()): print('   %-40s %.6g' % (c, v))
PY
