"""Scratch: trace-kernel throughput across scenes of growing footprint."""
import sys, time
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth, cornell
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r = Renderer(0)
r.set_option('time_kernels', 1).set_option('wf_waves_per_cu', 20)
for name, mk in [('cornell', lambda: cornell(1920, 1080)), ('mesh10k', lambda: scenes_synth.mesh10k()),
                 ('atrium250k', lambda: scenes_synth.atrium250k()), ('soup1M', lambda: scenes_synth.soup(1_000_000)),
                 ('soup10M', lambda: scenes_synth.soup(10_000_000))]:
    ps = mk()
    t0 = time.time(); r.upload(ps).build_accel('bvh2'); tb = time.time() - t0
    r.enable_counters(True).reset_counters(); r.frame(spp).sync(); c = r.counters(); r.enable_counters(False)
    best = (1e9, 0)
    for _ in range(3):
        r.reset(); r.frame(spp).sync(); ms, _ = r.last_trace_ms(); k, nl = r.last_kernel_ms(); best = min(best, (ms, k))
    st = r.accel_stats()
    print(f"{name:11s} build {tb:5.1f}s depth {st['max_depth']:2d} MB {st['bytes']/1e6:7.1f} rays {c['rays']/1e6:7.1f}M boxes/ray {c['nodes']/c['rays']:5.1f} prims/ray {c['prims']/c['rays']:4.1f} "
          f"total {best[0]:7.2f} ms trace {best[1]:7.2f} ms  => {c['rays']/best[0]/1e3:7.1f} Mrays/s overall, {c['rays']/best[1]/1e3:7.1f} in-kernel, "
          f"{best[1]*1e6/ (c['nodes']/2 + c['prims']):.3f} ns/step", flush=True)
