"""Pool-size sweep with batches parked across calls (wf_defer=1): S2 1080p 64 spp per call, 8 calls + sync."""
import sys, time
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
def t(calls=8, n=2):
    best = 1e9
    for _ in range(n):
        r.reset(); r.sync()
        t0 = time.perf_counter()
        for _ in range(calls): r.frame(64)
        r.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3 / calls)
    return best
for name, bands in (('frame', None), ('1/8 bands', (8, 8, 3))):
    if bands: r.set_row_bands(*bands)
    for pool in (0, 1 << 20, 3 << 19, 1 << 21, 3 << 20, 1 << 22, 1 << 23, 1 << 24):
        for pipes in (2, 3):
            r.set_option('wf_pool', pool).set_option('wf_pipes', pipes)
            print(name, 'pool', pool, 'pipes', pipes, '-> %.2f ms per call' % t(), flush=True)
