"""Small fixed workload for rocprofv3 counter passes: S2 atrium250k 1080p, 3 launches of 8 spp."""
import sys
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
scene = sys.argv[1] if len(sys.argv) > 1 else 'atrium250k'
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ps = scenes_synth.SCENES[scene](1920, 1080) if scene != 'soup' else scenes_synth.soup(10_000_000, 1920, 1080)
r = Renderer(0)
r.upload(ps).build_accel('bvh2')
print(r.accel_stats())
for _ in range(n):
    r.frame(spp).sync()
    print(r.last_trace_ms())
