"""Small batches (1-8 spp per call at 1080p): pool size, pipes and ring depth, steady state (60 pipelined calls + sync)."""
import sys, time
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
name = sys.argv[1] if len(sys.argv) > 1 else 'atrium250k'
ps = scenes_synth.SCENES[name](1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
def t(spp, calls=60, n=2):
    best = 1e9
    for _ in range(n):
        r.reset(); r.sync()
        t0 = time.perf_counter()
        for _ in range(calls): r.frame(spp)
        r.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3 / calls)
    return best
for spp in (1, 2):
    for ring in (4,):
        out = []
        for pool in (0, 1 << 21, 3 << 20, 1 << 22, 6 << 20):
            for pipes in (1, 2):
                r.set_option('wf_pool', pool).set_option('wf_pipes', pipes).set_option('wf_ring', ring)
                out.append('%.0fM/%d: %.2f' % (pool / 2**20, pipes, t(spp)))
        print(name, spp, 'spp ring', ring, '|', ' | '.join(out), flush=True)
