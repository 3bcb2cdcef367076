import sys
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
r.enable_counters(True).reset_counters(); r.frame(16).sync(); c = r.counters(); p = r.debug_probes()
print(c)
ii, ia, li, la, pi, sc, rf, rl = p
print('shard scans', sc, 'per trace launch', sc / max(r.last_kernel_ms()[1], 1) if False else '')
print('inner: wave-iters %d, avg active lanes %.1f' % (ii, ia / ii))
print('leaf : wave-passes %d, avg active lanes %.1f, avg max-cnt trips %.2f, prim tests %d -> util in prim loop %.2f' % (li, la / li, pi / li, c['prims'], c['prims'] / (pi * 64.0)))
print('refills %d, lanes per refill %.1f' % (rf, rl / max(rf, 1)))
print('node visits %d (%.1f per inner wave-iter)' % (c['nodes'] // 4, c['nodes'] / 4 / ii))
