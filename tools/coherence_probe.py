"""Does the 1-spp loop lose to the 64-spp loop because its paths in flight cover the whole image (64-spp batches keep
them inside a band)?  Rays/s of 1-spp loops on the full frame and on bands of it, pool held at 8 M slots."""
import sys, time
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
r.set_option('wf_pool', 1 << 23)
for name, tile, calls in (('full', (0, 0, 1920, 1080), 300), ('half', (0, 270, 1920, 810), 600), ('quarter', (0, 405, 1920, 675), 1200), ('eighth', (0, 472, 1920, 608), 2400)):
    r.set_tile(*tile)
    r.enable_counters(True).reset_counters(); r.frame(1).sync(); rays = r.counters()['rays']; r.enable_counters(False)
    r.reset(); r.frame(1).sync()
    t0 = time.perf_counter()
    for _ in range(calls): r.frame(1)
    r.sync()
    dt = time.perf_counter() - t0
    print('%-8s %7.3f ms per call, %.0f Mrays/s' % (name, dt * 1e3 / calls, rays * calls / dt / 1e6), flush=True)
