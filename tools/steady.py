"""Steady-state cost per 64-spp call with batches pipelined across calls: S2 1080p, 24 calls + one sync, for the
whole frame and for the row-band share of a 2-, 4- and 8-GPU run (one GPU rendering one share); sweeps of the
chunk size (iterations per status readback), the park threshold and k_wf_finish's paths per wave."""
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
for name, bands in (('frame', None), ('1/2 bands', (8, 2, 1)), ('1/4 bands', (8, 4, 1)), ('1/8 bands', (8, 8, 3))):
    if bands: r.set_row_bands(*bands)
    out = []
    for chunk, feed, ppw in ((2, 100, 64), (4, 100, 64), (2, 80, 64), (2, 125, 64), (1, 100, 16)):
        r.set_option('wf_chunk', chunk).set_option('wf_feed_pct', feed).set_option('wf_side_ppw', ppw)
        out.append('c%d f%d w%d: %.2f' % (chunk, feed, ppw, t()))
    print(name, ' | '.join(out), flush=True)
