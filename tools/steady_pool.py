"""Pool size in the steady state (batches pipelined across calls): S2 1080p 64 spp, 16 calls + one sync, and one synced call."""
import sys, time
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
def t(calls, n=2):
    best = 1e9
    for _ in range(n):
        r.reset(); r.sync()
        t0 = time.perf_counter()
        for _ in range(calls): r.frame(64)
        r.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3 / calls)
    return best
for name, bands in (('frame', None), ('1/2 bands', (8, 2, 1)), ('1/4 bands', (8, 4, 1))):
    if bands: r.set_row_bands(*bands)
    out = []
    for rnd in range(2):
        for pool in (1 << 22, 6 << 20, 1 << 23):
            r.set_option('wf_pool', pool)
            out.append('%.0fM: %.2f (synced %.2f)' % (pool / 2**20, t(16), t(1, 3)))
    print(name, ' | '.join(out), flush=True)
