#!/bin/bash
# The round's bench lines on the final build (outputs under gpurun_out/final/).  Usage: bash tools/final_benches.sh
set -x
o=gpurun_out/final; mkdir -p $o
python bench.py > $o/bench_n1.json 2> $o/bench_n1.err
python bench.py --spp 1 --steps 320 --warmup 32 --no-cpu-baseline > $o/bench_spp1.json 2> $o/bench_spp1.err
python bench.py --width 3840 --height 2160 --spp 256 --steps 2 --warmup 1 --no-cpu-baseline > $o/bench_4k256.json 2> $o/bench_4k256.err
python bench.py --scene soup --spp 16 --steps 3 --no-cpu-baseline > $o/bench_soup10M.json 2> $o/bench_soup10M.err
python bench.py --scene mesh10k --spp 1 --steps 320 --warmup 32 --no-cpu-baseline > $o/bench_mesh10k_1spp.json 2> $o/bench_mesh10k_1spp.err
python bench.py --opt wf_pipes=1 --opt wf_waves_per_cu=20 --steps 4 --warmup 1 --no-cpu-baseline > $o/bench_single_pipe_w20.json 2> $o/bench_single_pipe_w20.err
python tools/steady.py > $o/steady_state.log 2>&1
python tools/lbvh_ab.py big > $o/lbvh_vs_sah.log 2>&1
tail -n 3 $o/steady_state.log
for f in $o/bench_*.json; do python -c "
import json,sys
j=json.loads(open('$f').readline()); r=j['roofline']
print('$f', j['value'], j['ms_per_step'], r['frac'], r.get('whole_pass',{}).get('frac') if isinstance(r.get('whole_pass'),dict) else '')"; done
