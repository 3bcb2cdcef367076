"""Time each tuning build (tune_*.so in the repo root, built with `make -C computeraytracer_amd/csrc
OUT=$PWD/tune_x.so EXTRA=-DCRT_WF_...`) in the steady state: S2 1080p, 12 pipelined 64-spp calls and 300 1-spp calls."""
import glob, os, subprocess, sys
code = r'''
import sys, time; sys.path.insert(0,'.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920,1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
out = []
for spp, calls in ((64, 12), (1, 300)):
    best = 1e9
    for _ in range(2):
        r.reset(); r.sync(); t0 = time.perf_counter()
        for _ in range(calls): r.frame(spp)
        r.sync(); best = min(best, (time.perf_counter() - t0) * 1e3 / calls)
    out.append('%d spp: %.3f ms' % (spp, best))
print(' | '.join(out))
'''
libs = [None] + sorted(glob.glob('tune_*.so'))
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ)
        if lib: env['CRT_LIB'] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
        print(lib or 'default', out.stdout.strip() or out.stderr[-300:], flush=True)
