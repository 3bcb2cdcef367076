"""Scratch: timing of the wavefront pipeline under option sweeps (S2 1080p)."""
import sys, itertools
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ps = scenes_synth.atrium250k(1920, 1080)
r = Renderer(0)
r.upload(ps).build_accel('bvh2')
r.enable_counters(True).reset_counters(); r.frame(spp).sync(); c = r.counters(); r.enable_counters(False)
print('rays', c['rays'], 'nodes/ray', c['nodes'] / c['rays'], 'prims/ray', c['prims'] / c['rays'])
r.set_option('time_kernels', 1)
for wpc in (16, 20, 24):
    for pool in (1 << 20, 1 << 21, 1 << 22):
        r.set_option('wf_waves_per_cu', wpc).set_option('wf_pool', pool)
        best = 1e9
        for _ in range(3):
            r.reset(); r.frame(spp).sync(); ms, nl = r.last_trace_ms(); kms, kn = r.last_kernel_ms(); best = min(best, ms)
        print('waves/cu', wpc, 'pool', pool, 'ms', round(best, 2), 'Mrays/s', round(c['rays'] / best / 1e3, 1), 'trace-kernel ms', round(kms, 2), 'launches', kn)
