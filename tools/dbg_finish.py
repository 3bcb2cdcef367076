import sys, numpy as np
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, cornell
from oracle import orc
ps = cornell(); acc_o, rgba_o, cnt = orc.Scene.from_packed(ps).render(2)
r = Renderer(0)
for count, fin, pipes in [(1, 0, 2), (0, 0, 2), (1, 4096, 2), (0, 4096, 2), (1, 0, 1)]:
    nbad = []
    for rep in range(8):
        r.set_option('wf_finish_at', fin).set_option('wf_pipes', pipes)
        r.upload(ps).build_accel('bvh2').enable_counters(bool(count)).reset_counters()
        r.frame(2).sync()
        acc = r.read_accum()
        bad = (acc.view(np.uint32)[..., :3] != acc_o.view(np.uint32)[..., :3]).any(-1)
        nbad.append(int(bad.sum()))
        if bad.any() and len(nbad) < 3:
            ys, xs = np.nonzero(bad)
            for y, x in list(zip(ys, xs))[:4]:
                print('   px', x, y, 'gpu', acc[y, x, :3], 'oracle', acc_o[y, x, :3])
    print('count', count, 'finish_at', fin, 'pipes', pipes, 'bad px per run', nbad, flush=True)
