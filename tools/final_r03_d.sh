#!/bin/bash
# Round-3 evidence after the adaptive cohort (host-side change): parity suite, the 1-spp loops, display loops, steady-state share.
o=gpurun_out/r03_final_d; mkdir -p $o
export TMPDIR=/tmp
step() { echo "== $1" | tee -a $o/progress.txt; }
step tests;         timeout -k 10 900 python -m pytest tests -x -q -m gpu > $o/tests.log 2>&1; rc=$?; tail -3 $o/tests.log; if grep -q "Memory access fault" $o/tests.log; then exit 9; fi; [ $rc -eq 0 ] || exit 1
step bench_spp1;    timeout -k 10 200 python bench.py --spp 1 --steps 640 --warmup 64 --no-cpu-baseline > $o/bench_spp1.json 2> $o/bench_spp1.err || exit 1
step bench_spp1_320; timeout -k 10 200 python bench.py --spp 1 --steps 320 --warmup 32 --no-cpu-baseline > $o/bench_spp1_320.json 2> $o/bench_spp1_320.err || exit 1
step bench_spp1_fixed16; timeout -k 10 200 python bench.py --spp 1 --steps 640 --warmup 64 --no-cpu-baseline --opt wf_cohort_max=16 > $o/bench_spp1_fixed16.json 2> $o/bench_spp1_fixed16.err || exit 1
step bench_mesh10k; timeout -k 10 200 python bench.py --scene mesh10k --spp 1 --steps 640 --warmup 64 --no-cpu-baseline > $o/bench_mesh10k_1spp.json 2> $o/bench_mesh10k.err || exit 1
step steady;        timeout -k 10 400 python tools/steady.py > $o/steady_state.log 2>&1 || exit 1
step display;       python tools/dump_packed.py atrium250k 1920 1080 /tmp/s2 > /dev/null 2>&1 && timeout -k 10 200 node host/display_loop.js --packed /tmp/s2 --frames 2000 --lag 64 --ring 128 > $o/display_loop_node.json 2> $o/display_loop_node.err || exit 1
step display_py;    timeout -k 10 200 python tools/display_loop.py 2000 64 128 > $o/display_loop_py.txt 2>&1 || exit 1
step done
ls $o
