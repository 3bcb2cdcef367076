import sys, numpy as np
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, cornell
from oracle import orc
ps = cornell(); acc_o, rgba_o, cnt = orc.Scene.from_packed(ps).render(2)
r = Renderer(0)
for fin, count in [(0, 1), (4096, 1), (4096, 0), (0, 0)]:
    r.set_option('wf_finish_at', fin).set_option('wf_pipes', 2)
    nbad = []
    for rep in range(10):
        r.upload(ps).build_accel('bvh2').enable_counters(bool(count)).reset_counters()
        r.frame(2).sync()
        acc = r.read_accum()
        nbad.append(int((acc.view(np.uint32)[..., :3] != acc_o.view(np.uint32)[..., :3]).any(-1).sum()))
    print('finish_at', fin, 'count', count, 'bad px per run', nbad, flush=True)
