import sys
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
for pipes, wpc, pool in [(2, 12, 1 << 22), (2, 14, 1 << 22), (2, 16, 6 << 20), (2, 16, 1 << 23), (2, 12, 1 << 23), (3, 10, 6 << 20), (3, 12, 6 << 20)]:
    r.set_option('wf_pipes', pipes).set_option('wf_waves_per_cu', wpc).set_option('wf_pool', pool)
    best = 1e9
    for _ in range(3):
        r.reset(); r.frame(64).sync(); best = min(best, r.last_trace_ms()[0])
    print('pipes', pipes, 'waves/CU', wpc, 'pool', pool, 'ms per 64 spp: %.2f' % best, flush=True)
