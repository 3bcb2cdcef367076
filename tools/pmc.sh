#!/bin/bash
# usage: tools/pmc.sh <outdir> <workload args...>   -- separate rocprofv3 --pmc passes (never mixed with sys-trace)
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "pass $i: $grp" >> $out/progress.txt; timeout -k 10 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python3 tools/prof_workload.py "$@" > $out/p$i.log 2>&1 || echo "pass $i failed" >> $out/progress.txt
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('$out/p*/*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:60]
        agg[k][row['Counter_Name']] += float(row['Counter_Value'])
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()): print('   %-36s %.6g' % (c, v))
PY
