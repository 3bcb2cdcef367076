#!/bin/bash
# Round-3 evidence, second leg on the final build: final_r03_b.sh (SQ counters of both traversal forms, the traffic passes, steady-state
# shares, display loops, latency) and then the headline bench lines with roofline.traffic from those very passes.
o=gpurun_out/r03_final; mkdir -p $o
rm -f $o/traffic.json
bash tools/final_r03_b.sh > $o/leg_b.log 2>&1 || exit 1
[ -s $o/traffic.json ] || { echo "no traffic.json"; exit 1; }
cp $o/traffic.json profiles/r03_traffic.json
echo "== bench_n1_final" >> $o/progress.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > $o/bench_n1_final.json 2> $o/bench_n1_final.err || exit 1
timeout -k 10 400 python bench.py --scene soup --spp 16 --steps 4 --warmup 1 --no-cpu-baseline > $o/bench_soup10M.json 2> $o/bench_soup.err || exit 1
timeout -k 10 400 python bench.py --scene soup --spp 16 --steps 4 --warmup 1 --no-cpu-baseline --accel lbvh > $o/bench_soup10M_lbvh.json 2> $o/bench_soup_lbvh.err || exit 1
echo "== done_bc" >> $o/progress.txt
tail -4 $o/progress.txt
