import sys
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
for N in (8, 4, 2):
    ts = []
    for k in range(N):
        r.set_row_bands(8, N, k)
        r.frame(64).sync(); r.reset(); r.frame(64).sync()
        ts.append(r.last_trace_ms()[0])
    print(N, [round(t, 1) for t in ts], 'max', round(max(ts), 2), 'mean', round(sum(ts) / N, 2))
r.set_tile(0, 0, 1920, 1080); r.frame(64).sync(); r.reset(); r.frame(64).sync(); print('full', r.last_trace_ms())
