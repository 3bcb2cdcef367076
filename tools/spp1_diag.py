"""The reference's own frame loop (one sample per crt_trace call, main.js:597-611) on S2 at 1080p:
ms per call in the steady state over pipeline settings, and a per-chunk status log (CRT_DEBUG=1) of a few calls."""
import os, sys, time
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
name = sys.argv[1] if len(sys.argv) > 1 else 'atrium250k'
ps = scenes_synth.SCENES[name](1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')


def t(spp, calls=80, n=2):
    best, host = 1e9, 0
    for _ in range(n):
        r.reset(); r.sync()
        for _ in range(8): r.frame(spp)      # reach the steady state first
        t0 = time.perf_counter()
        for _ in range(calls): r.frame(spp)
        t1 = time.perf_counter()
        r.sync()
        t2 = time.perf_counter()
        if (t2 - t0) < best: best, host = t2 - t0, t1 - t0
    return best * 1e3 / calls, host * 1e3 / calls


for spp in (1, 4):
    out = []
    for opts in ({}, {'wf_pool_spp': 4}, {'wf_pool_spp': 3}, {'wf_pool_spp': 4, 'wf_ahead': 2}, {'wf_pool_spp': 4, 'wf_ahead': 4}, {'wf_pool_spp': 4, 'wf_side_ppw': 16}, {'wf_pool_spp': 4, 'wf_finish_at': 65536}, {'wf_pool_spp': 4, 'wf_finish_at': 8192}, {'wf_ring': 16}, {'wf_chunk': 2}):
        for k, v in opts.items(): r.set_option(k, v)
        a, h = t(spp)
        out.append('%s: %.2f (host %.2f)' % (','.join('%s=%d' % kv for kv in opts.items()) or 'default', a, h))
        for k, v in {'wf_pool': 0, 'wf_pipes': 2, 'wf_chunk': 1, 'wf_finish_at': 32768, 'wf_feed_pct': 100, 'wf_pool_spp': 4, 'wf_ring': 32, 'wf_ahead': 3, 'wf_side_ppw': 64}.items(): r.set_option(k, v)
    print(name, spp, 'spp |', ' | '.join(out), flush=True)

r.reset()
for _ in range(8): r.frame(1)
os.environ['CRT_DEBUG'] = '1'
t0 = time.perf_counter()
for i in range(6):
    print('--- call', i, 't=%.2f ms' % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
    r.frame(1)
print('--- sync t=%.2f ms' % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
r.sync()
print('--- done t=%.2f ms' % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
