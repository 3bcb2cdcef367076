"""Does running K independent contexts (row bands) concurrently on one GPU beat one context?"""
import sys, threading, time
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080)
spp = 64
def run(K, wpc):
    rs = []
    for k in range(K):
        r = Renderer(0); r.upload(ps)
        if K > 1: r.set_row_bands(8, K, k)
        r.set_option('wf_waves_per_cu', wpc)
        r.build_accel('bvh2'); rs.append(r)
    def work(r): r.frame(spp).sync()
    best = 1e9
    for rep in range(3):
        for r in rs: r.reset()
        for r in rs: r.sync()
        ts = [threading.Thread(target=work, args=(r,)) for r in rs]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        best = min(best, time.perf_counter() - t0)
    for r in rs: r.close()
    return best
for K, wpc in [(1, 20), (2, 20), (2, 12), (2, 8), (3, 8), (4, 6)]:
    print('contexts', K, 'waves/CU each', wpc, 'ms per 64 spp frame: %.2f' % (run(K, wpc) * 1e3), flush=True)
