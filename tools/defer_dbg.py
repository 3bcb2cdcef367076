import sys, time, os
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
r.set_row_bands(8, 8, 3)
r.reset(); r.frame(64); r.frame(64); r.sync()
os.environ['CRT_DEBUG'] = '1'
t0 = time.perf_counter()
for i in range(3):
    print('--- call', i, 't=%.2f ms' % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
    r.frame(64)
print('--- sync t=%.2f ms' % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
r.sync()
print('--- done t=%.2f ms' % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
