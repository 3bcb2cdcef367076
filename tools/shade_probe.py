import sys, time; sys.path.insert(0,'.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920,1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
for pipes in (1, 2):
    r.set_option('wf_pipes', pipes)
    r.reset(); r.reset_counters(); r.sync()
    t0 = time.perf_counter()
    for _ in range(6): r.frame(64)
    r.sync(); dt = (time.perf_counter() - t0) * 1e3 / 6
    p = r.debug_probes()
    n = max(p[6], 1)
    names = ['slot streams arrive', 'hit record arrives', 'shading (incl. its loads)', 're-arm', 'write-back', 'list append']
    print('pipes %d: %.2f ms per 64-spp call; shade waves %d; cycles per wave: total %.0f' % (pipes, dt, n, sum(p[:6]) / n))
    for k in range(6): print('   %-28s %8.0f cycles (%.1f %%)' % (names[k], p[k] / n, 100.0 * p[k] / max(sum(p[:6]), 1)))
