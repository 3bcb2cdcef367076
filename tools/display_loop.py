"""The display loop of host/display_loop.js from Python, with the host time split by call: python tools/display_loop.py [frames] [lag] [ring] [option=value ...]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import ctypes as C
from computeraytracer_amd import Renderer, scenes_synth
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 640
lag = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ring = int(sys.argv[3]) if len(sys.argv) > 3 else 64
ps = scenes_synth.atrium250k(1920, 1080)
r = Renderer(0)
r.upload(ps).build_accel('bvh2').set_option('frame_ring', ring)
for o in sys.argv[4:]:
    if '=' in o:
        k, v = o.split('='); r.set_option(k, int(v))
buf = np.empty((1080, 1920, 4), np.uint8)
pin = '--nopin' not in sys.argv
if pin:
    assert r._lib.crt_pin_host(buf.ctypes.data, buf.nbytes) == 0
for _ in range(40): r.frame(1)
r.sync(); r.reset()
t_trace = t_read = t_again = 0.0
again = '--again' in sys.argv
t0 = time.perf_counter()
for k in range(1, frames + 1):
    a = time.perf_counter(); r.frame(1); b = time.perf_counter(); t_trace += b - a
    if k > lag:
        r._chk(r._lib.crt_read_sample_rgba8(r._h, k - lag, buf.ctypes.data)); c = time.perf_counter(); t_read += c - b
        if again:
            r._chk(r._lib.crt_read_sample_rgba8(r._h, k - lag, buf.ctypes.data)); t_again += time.perf_counter() - c
for k in range(max(1, frames - lag + 1), frames + 1):
    r._chk(r._lib.crt_read_sample_rgba8(r._h, k, buf.ctypes.data))
total = time.perf_counter() - t0
print('frames %d lag %d ring %d pinned %s: %.3f ms per frame (host: trace %.3f ms, read %.3f ms per frame%s)' % (frames, lag, ring, pin, total * 1e3 / frames, t_trace * 1e3 / frames, t_read * 1e3 / frames,
      ', the same frame read again %.3f ms' % (t_again * 1e3 / frames) if again else ''))
