"""Host binned-SAH build vs GPU LBVH build: build time (crt_build_accel, incl. collapse / quantise / upload) and
rendering cost (1080p, 16 spp per call, 8 pipelined calls + sync) with per-ray node and primitive counts."""
import sys, time
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
r = Renderer(0)
def t(spp=16, calls=8, n=2):
    best = 1e9
    for _ in range(n):
        r.reset(); r.sync()
        t0 = time.perf_counter()
        for _ in range(calls): r.frame(spp)
        r.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3 / calls)
    return best
scenes = [('mesh10k', lambda: scenes_synth.mesh10k(1920, 1080)), ('atrium250k', lambda: scenes_synth.atrium250k(1920, 1080)),
          ('soup1M', lambda: scenes_synth.soup(1_000_000, 1920, 1080))]
if len(sys.argv) > 1 and sys.argv[1] == 'big':
    scenes.append(('soup10M', lambda: scenes_synth.soup(10_000_000, 1920, 1080)))
for name, make in scenes:
    ps = make(); r.upload(ps)
    for mode in ('bvh2', 'lbvh'):
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter(); r.build_accel(mode); best = min(best, time.perf_counter() - t0)
        st = r.accel_stats()
        ms = t()
        r.reset(); r.enable_counters(True).reset_counters(); r.frame(4).sync(); c = r.counters(); r.enable_counters(False)
        print('%-10s %-5s build %8.1f ms  depth %2d  render %7.2f ms per 16 spp  boxes/ray %.1f  prims/ray %.2f' %
              (name, mode, best * 1e3, st['max_depth'], ms, c['nodes'] / c['rays'], c['prims'] / c['rays']), flush=True)
