#!/bin/bash
# Round-3 evidence on the final build: bench lines of every BASELINE configuration, rocprofv3 kernel stats + timeline,
# SQ counters (lane utilisation) of both traversal forms, HBM traffic by PMC (S2 and the 10 M soup, both builders), the
# single-pipe line, the scene triple, steady-state shares, the Node display loop.  Outputs under gpurun_out/r03_final/.
o=gpurun_out/r03_final; mkdir -p $o
export TMPDIR=/tmp
step() { echo "== $1" | tee -a $o/progress.txt; }
step pmc_sq_f2;     bash tools/pmc_sq.sh $o/pmc_f2 wf_trace_form=2 > $o/pmc_sq_form2.txt 2>&1; rm -rf $o/pmc_f2
step pmc_sq_f1;     bash tools/pmc_sq.sh $o/pmc_f1 wf_trace_form=1 > $o/pmc_sq_form1.txt 2>&1; rm -rf $o/pmc_f1
step traffic_s2;    timeout -k 10 600 python3 tools/traffic.py $o/traffic.json atrium250k 64 2 bvh2 > $o/traffic_s2.txt 2>&1
step traffic_soup;  timeout -k 10 900 python3 tools/traffic.py $o/traffic.json soup 16 2 bvh2 > $o/traffic_soup.txt 2>&1
step traffic_soup_lbvh; timeout -k 10 900 python3 tools/traffic.py $o/traffic.json soup 16 2 lbvh > $o/traffic_soup_lbvh.txt 2>&1
step steady;        timeout -k 10 400 python tools/steady.py > $o/steady_state.log 2>&1
step display;       python tools/dump_packed.py atrium250k 1920 1080 /tmp/s2 > /dev/null 2>&1 && timeout -k 10 200 node host/display_loop.js --packed /tmp/s2 --frames 2000 --lag 64 --ring 128 > $o/display_loop_node.json 2> $o/display_loop_node.err
step display_py;    timeout -k 10 200 python tools/display_loop.py 2000 64 128 > $o/display_loop_py.txt 2>&1
step latency;       timeout -k 10 200 python tools/latency.py > $o/latency_single_call.txt 2>&1
step done
ls $o
