"""Lane utilisation of the traversal kernel's passes (counting kernels): python tools/probes.py [name=value ...]"""
import sys
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
for o in sys.argv[1:]:
    k, v = o.split('='); r.set_option(k, int(v))
r.enable_counters(True).reset_counters(); r.frame(16).sync(); c = r.counters(); p = r.debug_probes()
print(sys.argv[1:], c)
ii, ia, li, la, pi, sc, rf, rl = p
print('inner: wave-passes %d, avg active lanes %.1f' % (ii, ia / ii))
print('leaf : wave-passes %d, avg active lanes / tasks %.1f, prim-loop trips or rounds %.2f per pass, prim tests %d -> lane use of the primitive tests %.2f' % (li, la / li, pi / li, c['prims'], c['prims'] / (pi * 64.0)))
print('refills %d, rays per refill %.1f, shard scans (form 1) / 64 x leaf rounds with a patch or sphere task (form 2) %d (= %.1f %% of the leaf rounds)' % (rf, rl / max(rf, 1), sc, sc / 64.0 / max(li, 1) * 100))
print('node visits %d (%.1f per inner wave-pass); per walked ray: %.2f node visits, %.2f prim tests, %.3f inner passes/64, %.3f leaf passes/64' % (
    c['nodes'] // 4, c['nodes'] / 4 / ii, c['nodes'] / 4 / c['walked'], c['prims'] / c['walked'], ii * 64 / c['walked'] / 64, li * 64 / c['walked'] / 64))
