import sys, os
sys.path.insert(0, '.')
os.environ['CRT_DEBUG'] = '1'
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps)
r.set_row_bands(8, 8, 3).build_accel('bvh2')
for pool in (2097152, 2088960):
    r.set_option('wf_pool', pool)
    r.reset(); r.frame(64).sync()
    print('=== pool', pool, file=sys.stderr, flush=True)
    r.reset(); r.frame(64).sync()
    print('=== done', r.last_trace_ms(), file=sys.stderr, flush=True)
