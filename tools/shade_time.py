import sys, os, subprocess
code = r'''
import sys; sys.path.insert(0,'.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920,1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
r.set_option('time_kernels', 1)
wpc = int(sys.argv[1]) if len(sys.argv) > 1 else 20
r.set_option('wf_waves_per_cu', wpc)
best=(1e9,0)
for _ in range(3):
    r.reset(); r.frame(32).sync(); ms,_ = r.last_trace_ms(); k,_ = r.last_kernel_ms(); best=min(best,(ms,k))
print('wpc %d: %.2f ms total, %.2f ms trace kernel, %.2f other' % (wpc, best[0], best[1], best[0]-best[1]))
'''
for arg in sys.argv[1:] or ['-']:
    lib, _, wpc = arg.partition(':')
    env = dict(os.environ)
    if lib != '-': env['CRT_LIB'] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, '-c', code] + ([wpc] if wpc else []), env=env, capture_output=True, text=True)
    print(lib, out.stdout.strip() or out.stderr[-300:], flush=True)
