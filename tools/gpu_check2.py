import sys, numpy as np
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, cornell
from oracle import orc
ps = cornell(256, 256); sc = orc.Scene.from_packed(ps)
r = Renderer(0); r.upload(ps).build_accel('bvh2')
log = sc.ray_log(219, 109, 1)
out = r.debug_intersect(log[:, 0:3], log[:, 3:6], log[:, 6].view(np.uint32))
for i in range(len(log)):
    print(i, 'o', log[i, :3], 'd', log[i, 3:6], 'excl', log[i, 6:7].view(np.uint32)[0], 'oracle idx/t', log[i, 7:8].view(np.uint32)[0], log[i, 8],
          'gpu idx/t', out[i, 7:8].view(np.uint32)[0], out[i, 0])
# broad ray-level sweep
bad = 0; tot = 0
for y in range(0, 256, 3):
    for x in range(0, 256, 3):
        log = sc.ray_log(x, y, 1)
        out = r.debug_intersect(log[:, 0:3], log[:, 3:6], log[:, 6].view(np.uint32))
        m = (log[:, 7].view(np.uint32) != out[:, 7].view(np.uint32))
        hit = log[:, 7].view(np.uint32) != 0xFFFFFFFF
        m |= hit & (log[:, 8].view(np.uint32) != out[:, 0].view(np.uint32))
        tot += len(log); bad += int(m.sum())
        if m.any() and bad < 10: print('mismatch at', x, y, np.nonzero(m)[0])
print('rays', tot, 'mismatches', bad)
