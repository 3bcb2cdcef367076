import sys, os, numpy as np
sys.path.insert(0, '.')
os.environ['CRT_DEBUG'] = '1'
from computeraytracer_amd import Renderer, cornell
from oracle import orc
ps = cornell(); acc_o, rgba_o, cnt = orc.Scene.from_packed(ps).render(2)
r = Renderer(0)
r.set_option('wf_finish_at', 0).set_option('wf_pipes', 2)
for rep in range(3):
    r.upload(ps).build_accel('bvh2').enable_counters(True).reset_counters()
    r.frame(2).sync()
    acc = r.read_accum()
    bad = (acc.view(np.uint32)[..., :3] != acc_o.view(np.uint32)[..., :3]).any(-1)
    print('=== bad px', int(bad.sum()), r.counters()['rays'], int(cnt[0]), file=sys.stderr, flush=True)
