"""Stress for the tail of the wavefront pipeline with two pipes on two streams: cornell 1000x1000 has
~1000 paths trapped in the glass sphere until MAXDEPTH, a 100-iteration tail.  Every frame must equal
the first one bit for bit.   python tools/tail_race.py [reps] [tail_walk 0|1]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, '.')

from computeraytracer_amd import Renderer, cornell

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
walk = int(sys.argv[2]) if len(sys.argv) > 2 else 1
r = Renderer(0)
ps = cornell()
ref = None
bad = []
t0 = time.time()
for finish_at, count in [(0, True), (4096, True), (4096, False), (0, False)]:
    r.set_option("wf_finish_at", finish_at).set_option("wf_pipes", 2).set_option("wf_tail_walk", walk)
    for i in range(reps):
        r.upload(ps).build_accel("bvh2").enable_counters(count).reset_counters()
        r.frame(2).sync()
        acc = r.read_accum().view(np.uint32)[..., :3].copy()
        if ref is None:
            ref = acc
        n = int((acc != ref).any(axis=-1).sum())
        if n:
            bad.append([finish_at, count, i, n])
    print(json.dumps({"finish_at": finish_at, "count": count, "reps": reps, "tail_walk": walk, "bad": bad, "s": round(time.time() - t0, 1)}), flush=True)
sys.exit(1 if bad else 0)
