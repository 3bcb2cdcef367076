"""Fixed workload for rocprofv3: the reference's frame loop (1 spp per crt_trace call) on S2 at 1080p."""
import sys, time
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
name = sys.argv[1] if len(sys.argv) > 1 else 'atrium250k'
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 60
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ps = scenes_synth.SCENES[name](1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
for k, v in (a.split('=') for a in sys.argv[4:]): r.set_option(k, int(v))
r.frame(spp).sync()
t0 = time.perf_counter()
for _ in range(calls): r.frame(spp)
r.sync()
print('%.3f ms per call' % ((time.perf_counter() - t0) * 1e3 / calls))
