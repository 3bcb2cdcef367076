"""Pool size and ring depth in the steady state (batches pipelined across calls): S2 1080p 64 spp, 24 calls + one sync."""
import sys, time
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
def t(calls=24, n=2):
    best = 1e9
    for _ in range(n):
        r.reset(); r.sync()
        t0 = time.perf_counter()
        for _ in range(calls): r.frame(64)
        r.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3 / calls)
    return best
for name, bands in (('frame', None), ('1/4 bands', (8, 4, 1)), ('1/8 bands', (8, 8, 3))):
    if bands: r.set_row_bands(*bands)
    for ring in (2, 4):
        out = []
        for pool in (0, 1 << 21, 3 << 20, 1 << 22, 1 << 23):
            r.set_option('wf_pool', pool).set_option('wf_ring', ring)
            out.append('%.0fM: %.2f' % (pool / 2**20, t()))
        print(name, 'ring', ring, '|', ' | '.join(out), flush=True)
