"""Steady-state cost per 64-spp call with batches pipelined across calls: S2 1080p, 24 calls + one sync, for the
whole frame and for the row-band share of a 2-, 4- and 8-GPU run (one GPU rendering one share)."""
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
    for opts in ({}, {'wf_pool_spp': 4}, {'wf_pool_spp': 16}, {'wf_cohort': 128}, {'wf_ahead': 5}):
        for k, v in opts.items(): r.set_option(k, v)
        out.append('%s: %.2f' % (','.join('%s=%d' % kv for kv in opts.items()) or 'default', t()))
        for k, v in {'wf_pool_spp': 8, 'wf_cohort': 16, 'wf_ahead': 3}.items(): r.set_option(k, v)
    print(name, ' | '.join(out), flush=True)
