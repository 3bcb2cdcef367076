import sys, os
sys.path.insert(0, '.')
os.environ['CRT_DEBUG'] = '1'
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 135
r.set_tile(0, 540, 1920, 540 + rows).build_accel('bvh2')
r.frame(64).sync(); print('warm', r.last_trace_ms(), file=sys.stderr)
r.frame(64).sync(); print('second', r.last_trace_ms(), file=sys.stderr)
