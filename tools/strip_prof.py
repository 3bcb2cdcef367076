import sys
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps)
r.set_row_bands(8, 8, 3).build_accel('bvh2')
r.frame(64).sync(); r.reset(); r.frame(64).sync(); print(r.last_trace_ms())
