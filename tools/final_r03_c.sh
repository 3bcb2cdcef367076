#!/bin/bash
# Round-3 evidence, last leg: the HBM traffic passes (S2, the 10 M soup with both builders) and the headline bench lines on the
# build they were measured on (bench.py reports roofline.traffic only for the kernel sources the passes ran on).
o=gpurun_out/r03_final_c; mkdir -p $o
export TMPDIR=/tmp
step() { echo "== $1" | tee -a $o/progress.txt; }
rm -f $o/traffic.json
step traffic_s2;    timeout -k 10 600 python3 tools/traffic.py $o/traffic.json atrium250k 64 2 bvh2 > $o/traffic_s2.txt 2>&1 || exit 1
step traffic_soup;  timeout -k 10 900 python3 tools/traffic.py $o/traffic.json soup 16 2 bvh2 > $o/traffic_soup.txt 2>&1 || exit 1
step traffic_soup_lbvh; timeout -k 10 900 python3 tools/traffic.py $o/traffic.json soup 16 2 lbvh > $o/traffic_soup_lbvh.txt 2>&1 || exit 1
cp $o/traffic.json profiles/r03_traffic.json
step bench_n1;      timeout -k 10 300 python bench.py --steps 20 --warmup 3 > $o/bench_n1.json 2> $o/bench_n1.err || exit 1
step bench_soup;    timeout -k 10 400 python bench.py --scene soup --spp 16 --steps 4 --warmup 1 --no-cpu-baseline > $o/bench_soup10M.json 2> $o/bench_soup.err || exit 1
step bench_soup_lbvh; timeout -k 10 400 python bench.py --scene soup --spp 16 --steps 4 --warmup 1 --no-cpu-baseline --accel lbvh > $o/bench_soup10M_lbvh.json 2> $o/bench_soup_lbvh.err || exit 1
step done
ls $o
