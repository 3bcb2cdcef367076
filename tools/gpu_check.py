"""Scratch GPU check: device math + Cornell parity vs the oracle."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, cornell
from oracle import orc

r = Renderer(0)
rng = np.random.default_rng(1)
for name, code, a, b in [
    ('sin', 0, rng.uniform(0, 6.3, 1 << 20), None), ('cos', 1, rng.uniform(0, 6.3, 1 << 20), None),
    ('exp', 2, rng.uniform(-110, 90, 1 << 20), None), ('log2', 3, np.exp(rng.uniform(-90, 88, 1 << 20)), None),
    ('exp2', 4, rng.uniform(-155, 130, 1 << 20), None),
    ('pow', 5, np.exp(rng.uniform(-10, 10, 1 << 20)), rng.uniform(-3, 3, 1 << 20)),
    ('sqrt', 6, np.exp(rng.uniform(-90, 88, 1 << 20)), None),
    ('div', 7, rng.normal(size=1 << 20) * 1e3, np.exp(rng.uniform(-30, 30, 1 << 20))),
    ('tan', 8, rng.uniform(0, 1.5, 1 << 20), None)]:
    a = a.astype(np.float32); b = None if b is None else b.astype(np.float32)
    g = r.debug_math(code, a, b); c = orc.math_eval(name, a, b)
    print(name, 'mismatch', int((g.view(np.uint32) != c.view(np.uint32)).sum()))

for size, spp in [(256, 1), (256, 17)]:
    ps = cornell(size, size)
    sc = orc.Scene.from_packed(ps)
    t = time.time(); acc_o, rgba_o, cnt_o = sc.render(spp); t_o = time.time() - t
    for mode in ('none', 'bvh2'):
        r.upload(ps).build_accel(mode).enable_counters(True).reset_counters()
        r.frame(spp).sync()
        acc = r.read_accum(); rgba = r.read_rgba8()
        ms, nl = r.last_trace_ms()
        c = r.counters()
        bad = (acc.view(np.uint32)[..., :3] != acc_o.view(np.uint32)[..., :3]).any(-1)
        print(size, spp, mode, 'accum mismatched px', int(bad.sum()), 'rgba mismatch', int((rgba != rgba_o).sum()),
              'ms', round(ms, 3), 'rays', c['rays'], 'oracle rays', int(cnt_o[0]), 'bounces', c['bounces'], int(cnt_o[3]),
              'nodes', c['nodes'], 'prims', c['prims'], 'oracle s', round(t_o, 2))
        if bad.any():
            ys, xs = np.nonzero(bad); print(' first bad', list(zip(xs[:5], ys[:5])), acc[ys[0], xs[0]], acc_o[ys[0], xs[0]])
print(r.accel_stats())
