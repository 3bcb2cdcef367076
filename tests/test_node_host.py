"""GPU: the Node host (host/main.js -> N-API addon -> libcrt.so) renders the
reference scene and matches the oracle bit for bit."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT, bits

pytestmark = pytest.mark.gpu
NODE = shutil.which("node")


@pytest.mark.skipif(NODE is None, reason="node not installed")
@pytest.mark.parametrize("unfused", [False, True])
def test_node_host_matches_oracle(tmp_path, orc, unfused):
    from computeraytracer_amd import cornell
    cmd = [NODE, os.path.join(ROOT, "host", "index.js"), "--width", "96", "--height", "72", "--spp", "3",
           "--dump", str(tmp_path / "img"), "--out", str(tmp_path / "img.ppm")] + (["--unfused"] if unfused else [])
    out = subprocess.run(cmd, capture_output=True, text=True, check=True)
    info = json.loads(out.stdout.strip().splitlines()[-1])
    assert info["sample"] == 3 and info["width"] == 96 and info["height"] == 72
    acc = np.frombuffer((tmp_path / "img.accum.bin").read_bytes(), np.float32).reshape(72, 96, 4)
    rgba = np.frombuffer((tmp_path / "img.rgba8.bin").read_bytes(), np.uint8).reshape(72, 96, 4)
    acc_o, rgba_o, cnt = orc.Scene.from_packed(cornell(96, 72)).render(3)
    assert np.array_equal(bits(acc)[..., :3], bits(acc_o)[..., :3]) and np.array_equal(rgba, rgba_o)
    assert info["rays"] == int(cnt[0])
    ppm = (tmp_path / "img.ppm").read_bytes()
    assert ppm.startswith(b"P6\n96 72\n255\n") and len(ppm) == 13 + 96 * 72 * 3


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_webgpu_shaped_host_matches_oracle(tmp_path, orc):
    """host/webgpu_main.js speaks WebGPU only (buffers, bind groups, command encoders,
    requestAnimationFrame); host/webgpu.js turns its submits into libcrt calls."""
    from computeraytracer_amd import cornell
    cmd = [NODE, os.path.join(ROOT, "host", "webgpu_main.js"), "--size", "80", "--frames", "3",
           "--dump", str(tmp_path / "fb.bin"), "--out", str(tmp_path / "fb.ppm")]
    out = subprocess.run(cmd, capture_output=True, text=True, check=True)
    info = json.loads(out.stdout.strip().splitlines()[-1])
    assert info == {"width": 80, "height": 80, "frames": 3}
    rgba = np.frombuffer((tmp_path / "fb.bin").read_bytes(), np.uint8).reshape(80, 80, 4)
    _, rgba_o, _ = orc.Scene.from_packed(cornell(80, 80)).render(3)
    assert np.array_equal(rgba, rgba_o)


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_async_napi_keeps_the_event_loop_free(tmp_path):
    """traceAsync / syncAsync / readRgba8Async (napi_async_work, Promises): the event loop turns while a 64-spp trace
    runs on a worker thread, blocking calls are refused meanwhile (ERR_CRT_BUSY), the image equals the synchronous
    one, a destroyed handle rejects."""
    out = subprocess.run([NODE, os.path.join(ROOT, "host", "async_demo.js"), "--size", "512", "--spp", "64"],
                         capture_output=True, text=True, check=True)
    info = json.loads(out.stdout.strip().splitlines()[-1])
    assert info["equal"] and info["busy"] and info["rejected"] and info["sample"] == 64
    assert info["ticks"] > 10, info


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_async_napi_random_walk(tmp_path):
    """host/async_walk.js: promises of traceAsync / readSampleRgba8Async / readRgba8Async / syncAsync queued without waiting
    for the ones before, invalid requests in between (they reject, the queue goes on), blocking calls refused meanwhile;
    every delivered frame equals the one a synchronous loop shows at that index."""
    out = subprocess.run([NODE, os.path.join(ROOT, "host", "async_walk.js"), "--size", "96", "--epochs", "6"],
                         capture_output=True, text=True, check=True)
    info = json.loads(out.stdout.strip().splitlines()[-1])
    assert info["ok"] and info["checked"] > 10 and info["rejected"] > 0, info


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_node_display_loop_shows_every_frame_once(tmp_path):
    """host/display_loop.js: trace(1) per frame, the display lagging by a cohort and a half; every frame index shown once,
    in order, each equal to the frame a synchronous trace(1); sync(); readRgba8() loop shows."""
    out = subprocess.run([NODE, os.path.join(ROOT, "host", "display_loop.js"), "--width", "160", "--height", "120", "--frames", "70",
                          "--lag", "24", "--ring", "48", "--check", "1"], capture_output=True, text=True, check=True)
    info = json.loads(out.stdout.strip().splitlines()[-1])
    assert info["shown"] == 70 and info["every_index_once_in_order"] and info["equal_to_synced_loop"] and info["latest"] == 70, info


@pytest.mark.skipif(NODE is None, reason="node not installed")
@pytest.mark.parametrize("world,band", [(2, 8), (3, 0)])
def test_node_multi_gpu_host_local_transport(tmp_path, orc, world, band):
    """host/multi.js --local: `world` ranks in one Node process on device 0 through the addon's multi-GPU calls
    (commInit / commPartition / gather / readFrameRgba8); the assembled frame is the oracle's."""
    from computeraytracer_amd import cornell
    cmd = [NODE, os.path.join(ROOT, "host", "multi.js"), "--gpus", str(world), "--local", "--width", "120", "--height", "77",
           "--spp", "2", "--frames", "2", "--band", str(band), "--dump", str(tmp_path / "frame.bin")]
    out = subprocess.run(cmd, capture_output=True, text=True, check=True)
    info = json.loads(out.stdout.strip().splitlines()[-1])
    assert info["gpus"] == world and sum(info["rows"]) == 77 and info["samples"] == 4
    rgba = np.frombuffer((tmp_path / "frame.bin").read_bytes(), np.uint8).reshape(77, 120, 4)
    _, rgba_o, _ = orc.Scene.from_packed(cornell(120, 77)).render(4)
    assert np.array_equal(rgba, rgba_o)


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_node_multi_gpu_host_forked_worker_over_rccl(tmp_path):
    """host/multi.js with one forked worker per GPU (here: the box's one GPU, world = 1): the parent never touches the
    GPU, rank 0 makes the RCCL id, the worker renders, gathers through RCCL and reports."""
    cmd = [NODE, os.path.join(ROOT, "host", "multi.js"), "--gpus", "1", "--width", "96", "--height", "64", "--spp", "2", "--frames", "2",
           "--out", str(tmp_path / "f.ppm")]
    out = subprocess.run(cmd, capture_output=True, text=True, check=True, timeout=120)
    info = json.loads(out.stdout.strip().splitlines()[-1])
    assert info["transport"] == "rccl" and info["rows"] == [64] and info["every_rank_holds_the_same_frame"]
    assert (tmp_path / "f.ppm").read_bytes().startswith(b"P6\n96 64\n255\n")
