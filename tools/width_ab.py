"""A/B of the traversal node width (wf_width 4: 64-byte nodes, 8: 128-byte nodes): S2 1080p 64 spp,
6 pipelined calls + sync, with the counting kernels' per-ray node / primitive counts."""
import sys, time
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
name = sys.argv[1] if len(sys.argv) > 1 else 'atrium250k'
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ps = scenes_synth.SCENES[name](1920, 1080) if name != 'soup' else scenes_synth.soup(int(sys.argv[3]), 1920, 1080)
r = Renderer(0); r.upload(ps)
def t(calls=6, n=2):
    best = 1e9
    for _ in range(n):
        r.reset(); r.sync()
        t0 = time.perf_counter()
        for _ in range(calls): r.frame(spp)
        r.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3 / calls)
    return best
for rnd in range(2):
    for w in (4, 8):
        r.set_option('wf_width', w).build_accel('bvh2')
        ms = t()
        line = '%s width %d -> %.2f ms/call' % (name, w, ms)
        if rnd == 0:
            r.reset(); r.enable_counters(True).reset_counters(); r.frame(spp).sync(); c = r.counters(); r.enable_counters(False)
            line += '  rays %.1fM  boxes/ray %.2f  nodes/ray %.2f  prims/ray %.2f  Mrays/s %.0f' % (c['rays'] / 1e6, c['nodes'] / c['rays'], c['nodes'] / c['rays'] / w, c['prims'] / c['rays'], c['rays'] / ms / 1e3)
        print(line, flush=True)
