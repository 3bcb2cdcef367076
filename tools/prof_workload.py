"""Small fixed workload for rocprofv3 counter passes: python3 tools/prof_workload.py <scene> <spp> <calls> <accel> [option=value ...]
(1920x1080; `calls` synced crt_trace calls of `spp` samples; scene soup = the 10 M-triangle soup)."""
import sys
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
scene = sys.argv[1] if len(sys.argv) > 1 else 'atrium250k'
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
accel = sys.argv[4] if len(sys.argv) > 4 and '=' not in sys.argv[4] else 'bvh2'
ps = scenes_synth.SCENES[scene](1920, 1080) if scene != 'soup' else scenes_synth.soup(10_000_000, 1920, 1080)
r = Renderer(0)
r.upload(ps)
for o in sys.argv[4:]:
    if '=' in o:
        k, v = o.split('='); r.set_option(k, int(v))
r.build_accel(accel)
print(r.accel_stats())
for _ in range(n):
    r.frame(spp).sync()
    print(r.last_trace_ms())
