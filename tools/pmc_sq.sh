#!/bin/bash
# usage: tools/pmc_sq.sh <outdir> <option=value> -- the two SQ counter groups only (instruction mix and wave states)
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "pass $i: $grp" >> $out/progress.txt; timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python3 tools/prof_workload.py atrium250k 8 2 "$@" > $out/p$i.log 2>&1 || echo "pass $i failed" >> $out/progress.txt
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('$out/p*/*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:44]
        agg[k][row['Counter_Name']] += float(row['Counter_Value'])
for k, d in agg.items():
    if 'k_wf_trace' not in k and 'k_wf_shade' not in k and 'k_wf_gen' not in k: continue
    print(k)
    for c, v in sorted(d.items()): print('   %-28s %.5g' % (c, v))
    if d.get('SQ_INSTS_VALU'):
        print('   lane utilisation (THREAD_CYCLES_VALU / 64 / ACTIVE_INST_VALU... per instr): %.3f' % (d['SQ_THREAD_CYCLES_VALU'] / 64.0 / max(d['SQ_ACTIVE_INST_VALU'], 1) / 1.0))
PY
