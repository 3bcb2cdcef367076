"""Which part of bench.py's setup costs time?  S2 1080p 64 spp, 4 pipelined calls + sync."""
import sys, time
sys.path.insert(0, '.')
import torch
from computeraytracer_amd import Renderer, scenes_synth
dev = torch.device('cuda', 0)
ps = scenes_synth.atrium250k(1920, 1080)
def run(use_stream, bind, tk):
    r = Renderer(0)
    if use_stream:
        st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); r.set_stream(st.cuda_stream)
    r.upload(ps).build_accel('bvh2')
    if bind:
        acc = torch.zeros((1080, 1920, 4), dtype=torch.float32, device=dev); rgba = torch.zeros((1080, 1920, 4), dtype=torch.uint8, device=dev)
        r.bind_output(acc.data_ptr(), rgba.data_ptr())
    r.reset(); r.frame(64); r.sync()
    if tk: r.set_option('time_kernels', tk)
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); r.sync()
        t0 = time.perf_counter()
        for _ in range(4): r.frame(64)
        r.sync(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3 / 4)
        if tk: r.last_kernel_ms()
    print('stream', use_stream, 'bind', bind, 'time_kernels', tk, '-> %.2f ms/call' % best, flush=True)
    r.close()
cfg = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else None
if cfg:
    run(bool(cfg[0]), bool(cfg[1]), cfg[2])
else:
    run(False, False, 0); run(True, False, 0); run(False, True, 0); run(False, False, 1); run(False, False, 20000); run(True, True, 20000)
