"""Latency of ONE synced call (frame(n).sync(), nothing in flight before it) on S2 at 1080p, by pool size."""
import sys, time
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
for pool_spp in (1, 2, 4):
    r.set_option('wf_pool_spp', pool_spp)
    for n in (1, 4, 16):
        r.frame(n).sync()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); r.frame(n).sync(); best = min(best, (time.perf_counter() - t0) * 1e3)
        print('wf_pool_spp %d: frame(%d).sync() %.2f ms' % (pool_spp, n, best), flush=True)
