"""Sweeps of the straggler knobs: S2 1080p 64 spp per call.  synced = one call + sync (latency of a lone
frame: wf_flush_at, wf_flush_ppw); pipelined = 8 calls + one sync (wf_finish_at, wf_side_ppw)."""
import sys, time
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
def t(calls, n=3):
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
    for fl in (0, 1024, 4096, 16384):
        for ppw in ((4, 8, 16, 64) if fl else (8,)):
            r.set_option('wf_flush_at', fl).set_option('wf_flush_ppw', ppw)
            print(name, 'flush_at', fl, 'flush_ppw', ppw, '-> synced %.2f ms/call' % t(1, 4), flush=True)
    r.set_option('wf_flush_at', 4096).set_option('wf_flush_ppw', 8)
    for fin in (4096, 32768):
        for ppw in (16, 64):
            r.set_option('wf_finish_at', fin).set_option('wf_side_ppw', ppw)
            print(name, 'finish_at', fin, 'side_ppw', ppw, '-> pipelined %.2f ms/call' % t(8), flush=True)
    r.set_option('wf_finish_at', 32768).set_option('wf_side_ppw', 64)
