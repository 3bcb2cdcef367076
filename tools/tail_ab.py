"""A/B: pipes joined onto one stream in the tail (wf_serial_tail=1) or left concurrent (0);
S2 1080p 64 spp, whole frame and the 1/8 row-band share of an 8-GPU run."""
import sys
sys.path.insert(0, '.')
from computeraytracer_amd import Renderer, scenes_synth
ps = scenes_synth.atrium250k(1920, 1080); r = Renderer(0); r.upload(ps).build_accel('bvh2')
def t(n=5):
    best = 1e9
    for _ in range(n):
        r.reset(); r.frame(64).sync(); best = min(best, r.last_trace_ms()[0])
    return best
for name, bands in (('frame', None), ('1/8 bands', (8, 8, 3)), ('1/2 bands', (8, 2, 1))):
    if bands: r.set_row_bands(*bands)
    for rnd in range(2):
        for serial in (1, 0):
            for fin in (4096, 16384, 65536):
                r.set_option('wf_serial_tail', serial).set_option('wf_finish_at', fin)
                print(name, 'serial_tail', serial, 'finish_at', fin, '-> %.2f ms' % t(), flush=True)
