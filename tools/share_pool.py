"""The 1/8 row-band share of the 1080p frame (what one rank of an 8-GPU run renders), 64 spp per call, pipelined:
ms per call by pool size and cohort."""
import sys, time
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
r.set_row_bands(8, 8, 3)
def t(calls=32, n=2):
    best = 1e9
    for _ in range(n):
        r.reset(); r.sync()
        t0 = time.perf_counter()
        for _ in range(calls): r.frame(64)
        r.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3 / calls)
    return best
for cohort in (16, 32):
    out = []
    for pool in (0, 4 << 20, 8 << 20, 12 << 20, 16 << 20):
        r.set_option('wf_pool', pool).set_option('wf_cohort', cohort)
        out.append('%dM: %.2f' % (pool >> 20, t()))
    print('cohort %d (x8 for this tile) |' % cohort, ' | '.join(out), flush=True)
