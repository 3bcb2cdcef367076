#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the path-trace pass on the 250k-triangle scene at
1920x1080 (BASELINE.json metric), with the HBM roofline of the trace kernel and a
CPU baseline (the brute-force oracle) timed on the box's host cores.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  A "step" = `--spp` more samples of the whole frame
(progressive, like the reference's frame loop, main.js:584-621), the frame
partitioned into horizontal strips across ranks (scene replicated; RNG seeds use
global pixel coordinates so the image is bit-identical for every N), followed by
the RCCL gather of the strips.  Total work is fixed as N grows: scaling = strong.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


import contextlib


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints its version banner on STDOUT when a communicator is created; rank 0's stdout is ONE JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)            # (1.7 s of timed GPU work at 64 spp per step: long enough for an outside busy-sampler to see it)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="atrium250k", choices=["cornell", "mesh10k", "atrium250k", "soup"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64, help="samples per step (BASELINE config 3: 64)")
    ap.add_argument("--soup-tris", type=int, default=10_000_000)
    ap.add_argument("--band", type=int, default=8, help="rows per interleaved band for N>1 (0 = contiguous strips)")
    ap.add_argument("--accel", default="bvh2", choices=["bvh2", "lbvh"],
                    help="tree builder: bvh2 = binned SAH on the host (default), lbvh = crt_build_accel(LBVH), all on the GPU")
    ap.add_argument("--gather", default="torch", choices=["torch", "native"],
                    help="the strips' gather: torch = all_gather_into_tensor on torch-owned buffers (the harness); native = crt_comm_* / "
                         "crt_gather, the RCCL all-gather issued by libcrt itself (what the Node host uses; the id travels through torch.distributed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="crt_set_option before the run (e.g. wf_pipes=1 for the single-pipe profile); noted in config")
    ap.add_argument("--cpu-crop", default="256x144", help="oracle sample: centre crop WxH at 1 spp")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Plain `python bench.py --gpus N`: start the N rank processes (one per GPU) before anything in THIS process
        # touches the GPU, relay rank 0's JSON line and the exit code.  (Never exec from a process that has
        # initialised HIP; a child process is fine.)
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")
    # CRT_BENCH_BACKEND=gloo + CRT_BENCH_ONE_GPU=1: rehearsal of the N>1 path on a one-GPU box (all
    # ranks share cuda:0, strips gathered through host memory).  The measured configuration is RCCL.
    backend = os.environ.get("CRT_BENCH_BACKEND", "nccl")
    if os.environ.get("CRT_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
                t_ = torch.zeros(1, device=dev)
                dist.all_reduce(t_)                      # (communicators are created at the first collective)
                torch.cuda.synchronize()
            else:
                dist.init_process_group(backend)

    from computeraytracer_amd import Renderer, scenes_synth
    from computeraytracer_amd.distributed import StripFrame

    class HostStagedStripFrame(StripFrame):
        """Rehearsal only (CRT_BENCH_BACKEND=gloo): gloo has no device all_gather, so the strips go through host
        memory.  The measured configuration is StripFrame's own gather: one RCCL all_gather_into_tensor."""

        def gather(self, accum: bool = True):
            if self.world == 1:
                return
            if accum:
                out = torch.empty(self.full_accum.shape, dtype=self.full_accum.dtype)
                dist.all_gather_into_tensor(out, self.accum.cpu())
                self.full_accum.copy_(out)
            out = torch.empty(self.full_rgba.shape, dtype=self.full_rgba.dtype)
            dist.all_gather_into_tensor(out, self.rgba.cpu())
            self.full_rgba.copy_(out)

    W, H = args.width, args.height
    if args.scene == "soup":
        ps = scenes_synth.soup(args.soup_tris, W, H)
    else:
        ps = scenes_synth.SCENES[args.scene](W, H)
    ntri = int((ps.primitives["category"] == 2).sum())

    r = Renderer(local_rank)
    # one stream for the renderer and the collective (the null stream's handle, 0, would mean "the context's own")
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    r.set_stream(stream.cuda_stream)
    native = args.gather == "native"
    if native:
        # the communicator of the C ABI: rank 0 makes the RCCL id (ncclGetUniqueId inside libcrt), the others get it
        with stdout_to_stderr():
            ids = [Renderer.comm_unique_id(local=False) if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(ids, src=0)
            r.comm_init(ids[0], rank, world)
    r.upload(ps)
    for o in args.opt:
        k, v = o.split("=")
        r.set_option(k, int(v))
    sf = (HostStagedStripFrame if backend != "nccl" else StripFrame)(W, H, world, rank, dev, band=(args.band if world > 1 else 0))
    if native:
        r.comm_partition(args.band if world > 1 else 0)          # rows, strip / gather / frame buffers: all inside libcrt
    else:
        sf.apply(r)
    t0 = time.time()
    r.build_accel(args.accel)
    t_build = time.time() - t0
    if native:
        class NativeFrame:
            # the same surface as StripFrame, over crt_gather / crt_read_frame_*
            local_rows = r.tile[3]

            def gather(self, accum=True):
                r.gather(rgba8=True, accum=accum)

            def image(self):
                return torch.from_numpy(r.read_frame_accum()).to(dev), torch.from_numpy(r.read_frame_rgba8()).to(dev)
        sf = NativeFrame()
    else:
        # strips live in torch tensors (padded to equal size) so RCCL can gather them
        r.bind_output(sf.accum.data_ptr(), sf.rgba.data_ptr())

    # A step = one frame() of args.spp samples + the path's one exchange step, the all_gather of the rgba8
    # strips (RCCL over xGMI; the accumulator stays on its GPU like the reference's and is gathered once at
    # the end).  Steps are software-pipelined the way a display loop is: crt_trace returns with the frame's
    # last, longest paths still in flight (they finish under the next frames), so the gather issued after
    # frame k ships the latest COMPLETE frame in stream order (k-1 in practice) and frame K-1 is shipped
    # after the final sync.  K frames traced and completed, K gathers, all inside the timed region.
    trace = os.environ.get("CRT_BENCH_TRACE") == "1"

    def run_steps(k):
        ts = [time.perf_counter()]
        for i in range(k):
            r.frame(args.spp)
            if i > 0:
                sf.gather(accum=False)
            ts.append(time.perf_counter())
        r.sync()
        if k > 0:
            sf.gather(accum=False)
        ts.append(time.perf_counter())
        if trace:
            print("[trace] host ms per call:", [round((b - a) * 1e3, 2) for a, b in zip(ts, ts[1:])], file=sys.stderr, flush=True)

    def barrier():
        r.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    full_check = os.environ.get("CRT_BENCH_CHECK") == "1"    # rehearsal: compare the gathered frame with a 1-GPU render

    r.reset()
    run_steps(args.warmup)
    # from here on: HIP events around every launch of the traversal kernel, on the stream it is launched on
    # (the event pairs are created now, outside the timed region)
    if os.environ.get("CRT_BENCH_NOTK") != "1":
        r.set_option("time_kernels", args.steps * 40 * args.spp + 4096)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = r.last_kernel_ms() if os.environ.get("CRT_BENCH_NOTK") != "1" else (1.0, 1)
    r.set_option("time_kernels", 0)
    total_ms = elapsed * 1e3
    sf.gather(accum=True)      # final readout of the f32 XYZ accumulator (outside the timed steps)
    if full_check and (world > 1 or native) and rank == 0:
        acc_all, rgba_all = sf.image()
        ref = Renderer(local_rank)
        ref.upload(ps).build_accel(args.accel).frame((args.warmup + args.steps) * args.spp).sync()
        same = bool((torch.from_numpy(ref.read_accum()).to(dev) == acc_all).all()) and \
            bool((torch.from_numpy(ref.read_rgba8()).to(dev) == rgba_all).all())
        print(f"[check] gathered frame identical to a single-GPU render: {same}", file=sys.stderr, flush=True)
        ref.close()
        if not same:
            raise SystemExit("gathered frame differs from the single-GPU render")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Ray / node / primitive counts of exactly the timed samples: the path is a pure function of
    # (scene, pixel, sample), so re-running the same sample range with the counting kernel
    # variant (untimed) gives the exact counts.
    r.reset()
    r.write_accum(np.zeros((sf.local_rows, W, 4), np.float32), args.warmup * args.spp)
    r.enable_counters(True).reset_counters()
    for _ in range(args.steps):
        r.frame(args.spp)
    r.sync()
    c = r.counters()
    r.enable_counters(False)
    # SURVEY 8(d)'s canonical accounting prices a ray by the boxes a BVH2 walk tests (32 B each: 2 x vec3 + 2 x u32
    # per node).  Counted with the single-kernel form, which walks the BVH2 itself, on the first sample of the timed
    # range (per-ray counts are a property of the scene and the sample; the timed kernels walk the 4-wide tree).
    canon = None
    if rank == 0:
        r.set_option("pipeline", 0)
        r.reset()
        r.write_accum(np.zeros((sf.local_rows, W, 4), np.float32), args.warmup * args.spp)
        r.enable_counters(True).reset_counters()
        r.frame(1).sync()
        c2 = r.counters()
        r.enable_counters(False)
        r.set_option("pipeline", 1)
        canon = (32.0 * c2["nodes"] + 48.0 * c2["prims"] + 16.0 * c2["hits"] + 36.0 * c2["paths"]) / max(c2["rays"], 1)
        canon_counts = (c2["nodes"] / max(c2["rays"], 1), c2["prims"] / max(c2["rays"], 1))
    # The dominant kernel with the GPU to itself (one pipe, nothing beside it): two more untimed steps.  In the product two pipes
    # overlap a traversal launch with the other pipe's shade / generate launches, so the kernel's summed launch durations
    # (roofline.frac) measure how the pipes share the GPU as much as the kernel; this is the kernel alone.
    alone = None
    if rank == 0 and world == 1 and os.environ.get("CRT_BENCH_NOTK") != "1" and not any(o.startswith("wf_pipes=") for o in args.opt):
        r.set_option("wf_pipes", 1).set_option("wf_waves_per_cu", 20)
        r.reset()
        r.frame(args.spp).sync()
        r.set_option("time_kernels", 80 * args.spp + 4096)
        r.frame(args.spp).frame(args.spp).sync()
        a_ms, a_n = r.last_kernel_ms()
        r.set_option("time_kernels", 0).set_option("wf_pipes", 2).set_option("wf_waves_per_cu", 0)
        alone = (a_ms, a_n, 2)
    cnt = torch.tensor([c["rays"], c["nodes"], c["prims"], c["hits"], c["paths"], c["shadow"], c["walked"]], dtype=torch.float64, device=dev)
    mine = cnt.clone()
    if world > 1:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    rays, nodes, prims, hits, paths, shadow, walked = [float(v) for v in cnt.tolist()]

    if rank == 0:
        mrays = rays / elapsed / 1e6
        # Roofline of the dominant kernel (k_wf_trace, the BVH walk) on THIS rank: algorithmic bytes
        # (SURVEY.md 8d with this build's record sizes: node bytes per child box tested, 48 B per
        # primitive tested, 16 B per closest-hit attribute fetch, 36 B per pixel-sample of
        # framebuffer traffic) over the summed HIP-event durations of its launches in the timed region.
        st = r.accel_stats()
        m_rays, m_nodes, m_prims, m_hits, m_paths, _, _ = [float(v) for v in mine.tolist()]
        alg_bytes = st["bytes_per_box"] * m_nodes + 48.0 * m_prims + 16.0 * m_hits + 36.0 * m_paths
        per_launch = alg_bytes / max(launches, 1)
        avg_ms = kernel_ms / max(launches, 1)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        whole_pass = alg_bytes / (total_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC passes of this exact workload (a bench run cannot profile
        # itself: rocprofv3 --pmc serialises the kernels): reported only while the kernel sources are the ones it was
        # measured on, null otherwise.
        traffic, traffic_src, traffic_all = None, None, None
        trace_form = 2
        for o in args.opt:
            if o.startswith("wf_trace_form="):
                trace_form = int(o.split("=")[1])
        kernel_name = "k_wf_trace2" if (trace_form == 2 and st["bytes_per_box"] == 16 and st["width"] == 4) else "k_wf_trace"
        try:
            with open(os.path.join(ROOT, "profiles", "r03_traffic.json")) as f:
                entries = json.load(f)
            key = f"{args.scene} {W}x{H} {args.spp}spp n_gpus={world} accel={args.accel}"
            for tj in entries:
                if tj["workload"] == key and tj.get("options", []) == args.opt and tj.get("kernel_sources_sha256") == kernel_sources_sha256():
                    for kn, kv in tj["kernels"].items():
                        if kn.startswith(kernel_name + "<false"):
                            traffic = round(kv["traffic_bytes_per_launch"])
                            traffic_src = f"profiles/r03_traffic.json ({tj['method']}; {kv['launches']} launches, L2 hit rate {kv['l2_hit_rate']:.3f})"
                    # every kernel of the pass, per timed step is not known here (launch counts differ between the PMC
                    # workload and this run): totals of the PMC workload itself
                    traffic_all = {kn: {"bytes_per_launch": round(kv["traffic_bytes_per_launch"]), "l2_hit_rate": round(kv["l2_hit_rate"], 4)}
                                   for kn, kv in tj["kernels"].items() if "<true" not in kn}
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "Mrays/sec (+ HBM GB/s vs peak), 1080p 250k-tri scene, 1/2/4/8 MI355X",
            "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"S2 {args.scene}: {ntri} triangles + cornell walls/light, {W}x{H}, {args.spp} spp per step"
                                    + (" (BASELINE config 3)" if (W, H, args.spp) == (1920, 1080, 64) else
                                       " (BASELINE config 4)" if (W, H, args.spp) == (3840, 2160, 256) else ""))
                                   if args.scene == "atrium250k" else
                                   f"{args.scene}: {ntri} triangles, {W}x{H}, {args.spp} spp per step",
                       "partition": (f"rows dealt to {world} GPUs in bands of {args.band}" if world > 1 and args.band else f"{world} horizontal strip(s)")
                                    + ", scene replicated, all_gather of the rgba8 strips per step"
                                    + (" (crt_gather: RCCL issued by libcrt)" if native else " (torch.distributed all_gather_into_tensor)"),
                       "accel": ("binned-SAH BVH2 collapsed to 4-wide 64-byte quantised nodes (host build %.2f s; crt_build_accel(LBVH) builds on the GPU in milliseconds)" % t_build)
                                if args.accel == "bvh2" else
                                ("LBVH built, collapsed to 4-wide 64-byte quantised nodes and leaf-ordered on the GPU (crt_build_accel(LBVH): %.3f s)" % t_build),
                       "options": args.opt,
                       "pipelining": "steps are software-pipelined like a display loop: frame() returns with its batch in flight "
                                     "(up to 32 batches, retired in order under the following frames); the gather after frame k ships the "
                                     "latest complete frame, the last frame is shipped after the final sync; every frame is complete "
                                     "and gathered inside the timed region"},
            "mpaths_per_s": round(paths / elapsed / 1e6, 3),
            "rays_per_path": round(rays / max(paths, 1), 3),
            "rays": {"intersect_calls": int(rays), "walked_bvh": int(walked), "shadow": int(shadow),
                     "note": "value counts the reference's intersect() invocations; shadow rays whose NEE term is exactly zero "
                             "(cos_theta == 0) or whose light primitive is missed are decided without a BVH walk"},
            "mrays_walked_per_s": round(walked / elapsed / 1e6, 3),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_all_kernels": traffic_all,
                         "measured_hbm_GBs_of_kernel": (round(traffic / (avg_ms * 1e-3) / 1e9, 1) if traffic else None),
                         "kernel": kernel_name, "launches": launches, "avg_launch_ms": round(avg_ms, 4),
                         "algorithmic_bytes_per_launch": round(per_launch),
                         "bytes_per_ray": round(alg_bytes / max(m_rays, 1), 1),
                         "node_bytes_per_box": st["bytes_per_box"], "node_width": st["width"],
                         "boxes_per_ray": round(m_nodes / max(m_rays, 1), 2), "prims_per_ray": round(m_prims / max(m_rays, 1), 2),
                         "kernel_share_of_pass": round(kernel_ms / max(total_ms, 1e-9), 3),
                         "kernel_alone": (None if not alone else {
                             "what": "the same kernel with the GPU to itself: one pipe (no shade / generate launch beside it), 20 traversal waves per CU, "
                                     "two untimed steps after the timed region; algorithmic bytes per step as in the timed steps",
                             "launches": alone[1], "avg_launch_ms": round(alone[0] / max(alone[1], 1), 4),
                             "achieved": round(alg_bytes / args.steps * alone[2] / (alone[0] * 1e-3) / 1e9, 2),
                             "frac": round(alg_bytes / args.steps * alone[2] / (alone[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}),
                         "note": "two half-pool pipes run on two streams, so a traversal launch and the other pipe's k_wf_shade / k_wf_gen overlap: "
                                 "summed launch durations exceed the wall time of the pass; whole_pass_* = bytes / wall time",
                         "whole_pass_GBs": round(whole_pass, 1), "whole_pass_frac": round(whole_pass / HBM_PEAK_GBS, 5),
                         "accounting": "this build's records: 16 B per child box of a 64-byte quantised 4-wide node + 48 B per primitive "
                                       "+ 16 B per hit + 36 B per path (conservative: fewer bytes than the canonical form)",
                         "canonical": {"what": "SURVEY 8(d) / BASELINE.md: 32 B per BVH2 box tested + 48 B per primitive + 16 B per hit + 36 B per path, "
                                               "per-ray counts of the BVH2 walk (single-kernel form) on the first timed sample",
                                       "bytes_per_ray": round(canon, 1), "boxes_per_ray": round(canon_counts[0], 2), "prims_per_ray": round(canon_counts[1], 2),
                                       "achieved": round(canon * m_rays / (kernel_ms * 1e-3) / 1e9, 2),
                                       "frac": round(canon * m_rays / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                       "whole_pass_frac": round(canon * m_rays / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ps, args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


def kernel_sources_sha256():
    """Hash of the device sources the PMC traffic figure was measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "computeraytracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) or name in ("crt_api.cpp", "crt_bvh.cpp"):     # (crt_api.cpp: the pool size sets the launch size; crt_bvh.cpp: the tree)
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def cpu_baseline(ps, args):
    """The CPU restatement of the reference shader (oracle, brute force = the reference's
    algorithm) on a bounded sample of the same workload, all host cores."""
    from oracle import orc
    cw, ch = [int(v) for v in args.cpu_crop.split("x")]
    W, H = ps.width, ps.height
    x0, y0 = (W - cw) // 2, (H - ch) // 2 + H // 8
    sc = orc.Scene.from_packed(ps)
    cores = orc.max_threads()           # the OpenMP threads the oracle actually runs on
    t0 = time.perf_counter()
    _, _, cnt = sc.render(1, rect=(x0, y0, x0 + cw, y0 + ch))
    dt = time.perf_counter() - t0
    return {"value": round(float(cnt[0]) / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{cw}x{ch} crop at ({x0},{y0}) of the same frame, 1 spp, brute force over all "
                      f"{len(ps.primitives)} primitives (the reference's O(N) loop), {float(cnt[1]):.3g} "
                      f"primitive tests in {dt:.1f} s"}


if __name__ == "__main__":
    main()
