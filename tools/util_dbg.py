"""Pool utilisation from the driver's status log: CRT_DEBUG=1 python tools/util_dbg.py [spp] [calls] 2> log; prints the
distribution of rays listed per iteration (a full 12 M-slot pipe lists ~1.4 rays per slot)."""
import os, re, sys, subprocess
sys.path.insert(0, '.')
if os.environ.get("CRT_DEBUG") != "1":
    env = dict(os.environ, CRT_DEBUG="1")
    p = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, stderr=subprocess.PIPE, text=True)
    rays = [int(m.group(1)) for m in re.finditer(r"rays (\d+) open", p.stderr)]
    per_it = [float(m.group(1)) for m in re.finditer(r"per_it (\d+)", p.stderr)]
    rays = [r for r in rays if r > 0]
    rays.sort()
    n = len(rays)
    print("statuses", n, "rays per iteration: p10 %.2fM p50 %.2fM p90 %.2fM mean %.2fM" % (rays[n // 10] / 1e6, rays[n // 2] / 1e6, rays[9 * n // 10] / 1e6, sum(rays) / n / 1e6))
    print("per_it (work items per iteration, last):", per_it[-5:])
    print(p.stdout)
    sys.exit(p.returncode)
from computeraytracer_amd import Renderer, scenes_synth
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ps = scenes_synth.SCENES['atrium250k'](1920, 1080)
r = Renderer(0)
r.upload(ps).build_accel('bvh2')
for o in sys.argv[3:]:
    k, v = o.split('=')
    r.set_option(k, int(v))
import time
r.frame(spp).sync()
t0 = time.perf_counter()
for _ in range(calls):
    r.frame(spp)
r.sync()
print("ms per call %.2f" % ((time.perf_counter() - t0) * 1e3 / calls))
