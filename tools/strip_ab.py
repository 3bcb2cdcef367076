import sys
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps)
r.set_row_bands(8, 8, 3).build_accel('bvh2')
def t(n=5):
    best = 1e9
    for _ in range(n):
        r.reset(); r.frame(64).sync(); best = min(best, r.last_trace_ms()[0])
    return best
for rnd in range(2):
    for pool in (0, 2088960, 1 << 21):
        r.set_option('wf_pool', pool)
        print('strip pool', pool, '-> %.2f ms' % t(), flush=True)
