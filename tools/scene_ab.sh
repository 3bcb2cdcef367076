#!/bin/bash
# The pass on scenes of growing size (cornell: a handful of nodes; mesh10k; S2), one pipe and two: where the time goes
# when there is (almost) nothing to traverse.  Usage: bash tools/scene_ab.sh [out] [lib]
out=${1:-gpurun_out/scene_ab.txt}
: > $out
for scene in cornell mesh10k atrium250k; do
  for pipes in 1 2; do
    CRT_LIB=${2:+$PWD/$2} python bench.py --scene $scene --steps 3 --warmup 1 --no-cpu-baseline --opt wf_pipes=$pipes 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
r = j['roofline']
print('$scene pipes=$pipes ms_per_step', j['ms_per_step'], 'Mrays/s', j['value'], 'walked', j['mrays_walked_per_s'], 'rays/path', j['rays_per_path'], 'trace avg ms', r['avg_launch_ms'], 'launches', r['launches'], 'share', r.get('kernel_share_of_pass'))" >> $out
  done
done
cat $out
