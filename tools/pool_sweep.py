"""Pool size in the steady state on the final build: S2 1080p, 16 pipelined 64-spp calls + one sync, ms per call."""
import sys, time
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
def t(calls=16, n=2):
    best = 1e9
    for _ in range(n):
        r.reset(); r.sync()
        t0 = time.perf_counter()
        for _ in range(calls): r.frame(64)
        r.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3 / calls)
    return best
out = []
for pool in [int(a) << 20 for a in (sys.argv[1:] or "4 6 8 12 16 24".split())]:
    r.set_option('wf_pool', pool)
    out.append('%dM: %.2f' % (pool >> 20, t()))
    print(' | '.join(out), flush=True)
