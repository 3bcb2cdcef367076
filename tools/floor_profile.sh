#!/bin/bash
# usage: tools/floor_profile.sh <outdir> -- where a pipelined 64-spp step goes on a scene with (almost) nothing to traverse
# (cornell), on mesh10k and on S2: per-kernel launch durations from the middle of a pipelined bench run + the overlap states
out=$1; export TMPDIR=/tmp; mkdir -p $out
for sc in cornell mesh10k atrium250k; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$sc -o p -- python3 bench.py --scene $sc --steps 5 --warmup 1 --no-cpu-baseline > $out/$sc.json 2> $out/$sc.err || { echo "$sc failed" >> $out/progress.txt; exit 1; }
  echo "$sc done" >> $out/progress.txt
  echo "== $sc" >> $out/summary.txt
  python3 -c "
import json; j = json.loads(open('$out/$sc.json').readline()); print('   ms_per_step', j['ms_per_step'], 'Mrays/s', j['value'], 'rays/path', j.get('rays_per_path'), 'boxes/ray', j.get('boxes_per_ray'), 'prims/ray', j.get('prims_per_ray'))" >> $out/summary.txt
  python3 - $out/$sc/p_kernel_stats.csv >> $out/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_wf' in r['Name']: print('   %-46s calls %5s  avg %8.1f us  total %8.2f ms' % (r['Name'][:46], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
  python3 tools/timeline.py $out/$sc/p_kernel_trace.csv >> $out/summary.txt 2>&1
  rm -rf $out/$sc
done
cat $out/summary.txt
