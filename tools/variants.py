"""Scratch: time each tuning build (tune_*.so) on S2 1080p."""
import glob, os, subprocess, sys
spp = sys.argv[1] if len(sys.argv) > 1 else '16'
code = r'''
import sys; sys.path.insert(0,'.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920,1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
spp = int(sys.argv[1]); wpc = int(sys.argv[2])
r.set_option('wf_waves_per_cu', wpc).set_option('time_kernels', 1)
best=(1e9,0)
for _ in range(4):
    r.reset(); r.frame(spp).sync(); ms,_ = r.last_trace_ms(); k,_ = r.last_kernel_ms(); best=min(best,(ms,k))
print('%.2f ms total, %.2f ms trace kernel' % best)
'''
libs = [None] + sorted(glob.glob('tune_*.so'))
for lib in libs:
    env = dict(os.environ)
    if lib: env['CRT_LIB'] = os.path.abspath(lib)
    for wpc in ((20, 24) if lib and 'w6' in lib else (20,)):
        out = subprocess.run([sys.executable, '-c', code, spp, str(wpc)], env=env, capture_output=True, text=True)
        print(lib or 'default', 'wpc', wpc, out.stdout.strip() or out.stderr[-300:], flush=True)
