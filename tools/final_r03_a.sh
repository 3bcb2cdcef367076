#!/bin/bash
# Round-3 evidence on the final build: bench lines of every BASELINE configuration, rocprofv3 kernel stats + timeline,
# SQ counters (lane utilisation) of both traversal forms, HBM traffic by PMC (S2 and the 10 M soup, both builders), the
# single-pipe line, the scene triple, steady-state shares, the Node display loop.  Outputs under gpurun_out/r03_final/.
o=gpurun_out/r03_final; mkdir -p $o
export TMPDIR=/tmp
step() { echo "== $1" | tee -a $o/progress.txt; }
step bench_n1;      timeout -k 10 300 python bench.py --steps 10 --warmup 2 > $o/bench_n1.json 2> $o/bench_n1.err
step bench_spp1;    timeout -k 10 200 python bench.py --spp 1 --steps 320 --warmup 32 --no-cpu-baseline > $o/bench_spp1.json 2> $o/bench_spp1.err
step bench_mesh10k; timeout -k 10 200 python bench.py --scene mesh10k --spp 1 --steps 320 --warmup 32 --no-cpu-baseline > $o/bench_mesh10k_1spp.json 2> $o/bench_mesh10k.err
step bench_4k;      timeout -k 10 300 python bench.py --width 3840 --height 2160 --spp 256 --steps 2 --warmup 1 --no-cpu-baseline > $o/bench_4k256.json 2> $o/bench_4k256.err
step bench_soup;    timeout -k 10 400 python bench.py --scene soup --spp 16 --steps 4 --warmup 1 --no-cpu-baseline > $o/bench_soup10M.json 2> $o/bench_soup.err
step bench_soup_lbvh; timeout -k 10 400 python bench.py --scene soup --spp 16 --steps 4 --warmup 1 --no-cpu-baseline --accel lbvh > $o/bench_soup10M_lbvh.json 2> $o/bench_soup_lbvh.err
step bench_1pipe;   timeout -k 10 200 python bench.py --opt wf_pipes=1 --opt wf_waves_per_cu=20 --steps 4 --warmup 1 --no-cpu-baseline > $o/bench_single_pipe.json 2> $o/bench_single_pipe.err
step bench_form1;   timeout -k 10 200 python bench.py --opt wf_trace_form=1 --steps 6 --warmup 2 --no-cpu-baseline > $o/bench_form1.json 2> $o/bench_form1.err
step scene_ab;      bash tools/scene_ab.sh $o/scene_ab.txt > /dev/null 2>&1
step kernel_stats;  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $o/prof -o p --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $o/prof_bench.json 2> $o/prof_bench.err
python3 tools/timeline.py $o/prof/p_kernel_trace.csv > $o/timeline.txt 2>&1; cp $o/prof/p_kernel_stats.csv $o/kernel_stats.csv; rm -rf $o/prof
step kernel_stats_1pipe; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $o/prof1 -o p --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --opt wf_pipes=1 --opt wf_waves_per_cu=20 > $o/prof_bench_1pipe.json 2> $o/prof_bench_1pipe.err
cp $o/prof1/p_kernel_stats.csv $o/kernel_stats_single_pipe.csv; rm -rf $o/prof1
step done_a
ls $o
