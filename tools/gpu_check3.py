import sys, time, numpy as np
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, cornell
from oracle import orc
r = Renderer(0)
for size, spp in [(256, 17), (1000, 4)]:
    ps = cornell(size, size); sc = orc.Scene.from_packed(ps)
    t = time.time(); acc_o, rgba_o, cnt_o = sc.render(spp); t_o = time.time() - t
    r.upload(ps).build_accel('bvh2').enable_counters(True).reset_counters()
    r.frame(spp).sync()
    acc = r.read_accum(); rgba = r.read_rgba8(); ms, nl = r.last_trace_ms(); c = r.counters()
    bad = (acc.view(np.uint32)[..., :3] != acc_o.view(np.uint32)[..., :3]).any(-1)
    print(size, spp, 'accum mismatched px', int(bad.sum()), 'rgba mismatch', int((rgba != rgba_o).sum()), 'ms', round(ms, 3),
          'rays', c['rays'], int(cnt_o[0]), 'oracle s', round(t_o, 2))
    r.enable_counters(False).reset(); r.frame(spp).sync(); ms2, _ = r.last_trace_ms(); print(' no-count ms', ms2, 'Mrays/s', c['rays'] / ms2 / 1e3)
