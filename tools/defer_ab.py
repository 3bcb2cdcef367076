"""A/B of batches parked across crt_trace calls (wf_defer): S2 1080p, 64 spp per call, 8 calls back to
back and one sync; whole frame and the 1/8 and 1/2 row-band shares of a multi-GPU run."""
import sys, time
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
def t(calls=8, n=3):
    best = 1e9
    for _ in range(n):
        r.reset(); r.sync()
        t0 = time.perf_counter()
        for _ in range(calls): r.frame(64)
        r.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3 / calls)
    return best
for name, bands in (('frame', None), ('1/8 bands', (8, 8, 3)), ('1/2 bands', (8, 2, 1))):
    if bands: r.set_row_bands(*bands)
    for rnd in range(2):
        for defer in (0, 1):
            r.set_option('wf_defer', defer)
            print(name, 'wf_defer', defer, '-> %.2f ms per call' % t(), flush=True)
