"""Pipes x traversal waves per CU in the steady state: S2 1080p 64 spp, 16 pipelined calls + sync."""
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
for name, bands in (('frame', None), ('1/8 bands', (8, 8, 3))):
    if bands: r.set_row_bands(*bands)
    out = []
    for pipes in (2, 3, 4):
        for wpc in (10, 16):
            r.set_option('wf_pipes', pipes).set_option('wf_waves_per_cu', wpc)
            out.append('K%d w%d: %.2f' % (pipes, wpc, t()))
    print(name, ' | '.join(out), flush=True)
