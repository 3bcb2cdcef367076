"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the
C ABI against the CPU oracle on the same inputs -- bit-exact on the f32 XYZ
accumulator and on every rgba8 byte -- plus size-independent properties at the
benchmark sizes (tile-partition invariance, fused-vs-incremental samples,
BVH == the reference's own loop)."""
import json
import os
import time

import numpy as np
import pytest

from conftest import GOLDEN, bits

pytestmark = pytest.mark.gpu

MAXU = 0xFFFFFFFF


def assert_same_image(acc, rgba, acc_o, rgba_o, rect=None):
    if rect is not None:
        x0, y0, x1, y1 = rect
        acc_o, rgba_o = acc_o[y0:y1, x0:x1], rgba_o[y0:y1, x0:x1]
    bad = (bits(acc)[..., :3] != bits(acc_o)[..., :3]).any(-1)
    assert not bad.any(), f"{int(bad.sum())} accumulator pixels differ, first at {np.argwhere(bad)[0][::-1]}"
    assert np.array_equal(rgba, rgba_o), f"{int((rgba != rgba_o).sum())} rgba8 bytes differ"


def render(r, ps, spp, mode="bvh2", tile=None):
    r.upload(ps)
    if tile is not None:
        r.set_tile(*tile)
    r.build_accel(mode)
    r.frame(spp).sync()
    return r.read_accum(), r.read_rgba8()


# ------------------------------------------------------------------ numerics layer
def test_device_math_is_bit_exact(renderer, orc):
    rng = np.random.default_rng(1)
    n = 1 << 18
    cases = [("sin", 0, rng.uniform(0, 6.3, n), None), ("cos", 1, rng.uniform(0, 6.3, n), None),
             ("exp", 2, rng.uniform(-110, 90, n), None), ("log2", 3, np.exp(rng.uniform(-95, 88, n)), None),
             ("exp2", 4, rng.uniform(-155, 130, n), None),
             ("pow", 5, np.exp(rng.uniform(-10, 10, n)), rng.uniform(-3, 3, n)),
             ("sqrt", 6, np.exp(rng.uniform(-95, 88, n)), None),
             ("div", 7, rng.normal(size=n) * 1e3, np.exp(rng.uniform(-30, 30, n))),
             ("tan", 8, rng.uniform(0, 1.5, n), None)]
    for name, code, a, b in cases:
        a = a.astype(np.float32)
        b = None if b is None else b.astype(np.float32)
        assert np.array_equal(bits(renderer.debug_math(code, a, b)), bits(orc.math_eval(name, a, b))), name
    # denormal results and special values go the same way on both sides
    a = np.float32([1e-30, 1e-38, 3e-39, 0.0, 1e30, np.inf, np.nan, -1.0])
    b = np.float32([1e10, 1e5, 7.0, 1.0, 1e-10, 2.0, 1.0, 0.0])
    assert np.array_equal(bits(renderer.debug_math(7, a, b)), bits(orc.math_eval("div", a, b)))
    assert np.array_equal(bits(renderer.debug_math(2, np.float32([-100, -103.9, -90]))),
                          bits(orc.math_eval("exp", np.float32([-100, -103.9, -90]))))


# ------------------------------------------------------------------ reference scene
@pytest.mark.parametrize("mode", ["none", "bvh2"])
def test_cornell_matches_committed_golden(renderer, golden_buffers, golden_256, mode):
    """Against tests/golden (no oracle at run time): spp 1, 2, 16, 17 (17 wraps sample % 16)."""
    from computeraytracer_amd.scene import PackedScene, PRIM_DTYPE
    b = golden_buffers
    ps = PackedScene(b["primitives"].view(PRIM_DTYPE), b["lights"].view(PRIM_DTYPE), b["camera"], b["spectra"], b["cie"])
    renderer.upload(ps).build_accel(mode)
    done = 0
    for spp in (1, 2, 16, 17):
        renderer.frame(spp - done).sync()          # progressive, like the reference's frame loop
        done = spp
        acc, rgba = renderer.read_accum(), renderer.read_rgba8()
        assert np.array_equal(rgba, golden_256[f"rgba_{spp}"])
        assert np.array_equal(bits(acc[112:144, 112:144]), bits(golden_256[f"accum_crop_{spp}"]))
        np.testing.assert_allclose(acc.astype(np.float64).sum((0, 1)), golden_256[f"accum_sum_{spp}"], rtol=1e-12)
    with open(os.path.join(GOLDEN, "probes.json")) as f:
        probes = json.load(f)["probe_pixels"]
    want = golden_256["accum_probe_17"]
    got = np.asarray([acc[y, x] for x, y in probes])
    assert np.array_equal(bits(got), bits(want))


def test_cornell_native_size_vs_oracle(renderer, orc):
    """The reference's own configuration: 1000 x 1000 (cornell.json camera)."""
    from computeraytracer_amd import cornell
    ps = cornell()
    assert (ps.width, ps.height) == (1000, 1000)
    acc_o, rgba_o, cnt_o = orc.Scene.from_packed(ps).render(2)
    renderer.upload(ps).build_accel("bvh2").enable_counters(True).reset_counters()
    renderer.frame(2).sync()
    c = renderer.counters()
    renderer.enable_counters(False)
    assert_same_image(renderer.read_accum(), renderer.read_rgba8(), acc_o, rgba_o)
    assert (c["rays"], c["paths"], c["bounces"], c["shadow"]) == (int(cnt_o[0]), int(cnt_o[2]), int(cnt_o[3]), int(cnt_o[4]))


def test_ragged_size_and_brute_mode(renderer, orc):
    """Width/height not multiples of the 8x8 tile; the reference loop on the GPU."""
    from computeraytracer_amd import cornell
    ps = cornell(250, 131)
    acc_o, rgba_o, _ = orc.Scene.from_packed(ps).render(3)
    for mode in ("none", "bvh2"):
        acc, rgba = render(renderer, ps, 3, mode)
        assert acc.shape == (131, 250, 4)
        assert_same_image(acc, rgba, acc_o, rgba_o)


# ------------------------------------------------------------------ triangle scenes
def _crops_vs_oracle(renderer, orc, ps, spp, crops):
    acc, rgba = render(renderer, ps, spp)
    sc = orc.Scene.from_packed(ps)
    for rect in crops:
        acc_o, rgba_o, _ = sc.render(spp, rect=rect)
        x0, y0, x1, y1 = rect
        assert_same_image(acc[y0:y1, x0:x1], rgba[y0:y1, x0:x1], acc_o, rgba_o, rect)


def test_mesh10k_1080p_crops_vs_oracle(renderer, orc):
    """BASELINE config 2 (10k-tri mesh, 1920x1080): GPU renders the full frame, the
    brute-force oracle re-renders crops of it."""
    from computeraytracer_amd.scenes_synth import mesh10k
    ps = mesh10k(1920, 1080)
    assert len(ps.primitives) == 6 + 10368
    _crops_vs_oracle(renderer, orc, ps, 2, [(900, 500, 948, 532), (700, 620, 748, 652), (1200, 200, 1232, 232),
                                            (0, 0, 16, 16), (1904, 1064, 1920, 1080)])


def test_atrium250k_1080p_crops_vs_oracle(renderer, orc):
    """BASELINE config 3 scene (250k tris, deep BVH) at 1080p."""
    from computeraytracer_amd.scenes_synth import atrium250k
    ps = atrium250k(1920, 1080)
    assert len(ps.primitives) == 6 + 253952
    # (four fused samples: the pool then holds paths of several samples and bounces at once; crops on the floor, the columns,
    # the ceiling light, a wall and a frame corner)
    _crops_vs_oracle(renderer, orc, ps, 4, [(940, 600, 988, 624), (700, 400, 748, 424), (1100, 820, 1148, 844), (930, 20, 978, 44),
                                            (0, 500, 32, 524), (1888, 1056, 1920, 1080)])


def test_soup_vs_oracle(renderer, orc):
    """BASELINE config 5 generator at reduced count (random-triangle soup)."""
    from computeraytracer_amd.scenes_synth import soup
    ps = soup(200_000, 96, 54)
    acc_o, rgba_o, _ = orc.Scene.from_packed(ps).render(2)
    acc, rgba = render(renderer, ps, 2)
    assert_same_image(acc, rgba, acc_o, rgba_o)


def _mixed_scene(w, h):
    """Patches + a glass and a diffuse sphere + triangles in one BVH."""
    from computeraytracer_amd import scene as S
    from computeraytracer_amd.scenes_synth import mesh10k
    base = mesh10k(w, h)
    idx = base.spectrum_index
    sph = S.make_primitives([1, 1], [[120, 90, 150], [430, 110, 180]], [[60] * 3, [75] * 3], [[0] * 3] * 2,
                            [idx["dark"]] * 2, [idx["red"], idx["white"]], [0, 2], first_index=len(base.primitives))
    prims = np.zeros(len(base.primitives) + 2, S.PRIM_DTYPE)
    prims[:-2] = base.primitives
    prims[-2:] = sph
    return S.PackedScene(prims, S.lights_of(prims), base.camera, base.spectra, base.cie)


def test_mixed_categories_with_glass(renderer, orc):
    ps = _mixed_scene(160, 90)
    acc_o, rgba_o, _ = orc.Scene.from_packed(ps).render(4)
    acc, rgba = render(renderer, ps, 4)
    assert_same_image(acc, rgba, acc_o, rgba_o)


def _three_lights_scene(w, h):
    """Cornell with two more light patches (another emission spectrum, other sizes): the light is chosen per lane, its
    record and its primitive's record are gathered per lane (the one-light scalar path does not apply), and
    `lights[emission_index]` (sic) clamps to the LAST light instead of the only one."""
    from computeraytracer_amd import scene as S
    c = S.cornell(w, h)
    idx = c.spectrum_index
    extra = S.make_primitives([0, 0], [[2.0, 180.0, 150.0], [300.0, 2.0, 60.0]], [[0.0, 120.0, 0.0], [90.0, 0.0, 0.0]],
                              [[0.0, 0.0, 160.0], [0.0, 0.0, 70.0]], [idx["lightAlt"], idx["light"]], [idx["white"]] * 2,
                              [S.TYPE_INDEX["light"]] * 2, first_index=len(c.primitives))
    prims = np.zeros(len(c.primitives) + 2, S.PRIM_DTYPE)
    prims[:-2] = c.primitives
    prims[-2:] = extra
    lights = S.lights_of(prims)
    assert len(lights) == 3
    return S.PackedScene(prims, lights, c.camera, c.spectra, c.cie)


def test_several_lights(renderer, orc):
    ps = _three_lights_scene(128, 96)
    acc_o, rgba_o, cnt_o = orc.Scene.from_packed(ps).render(6)
    for pipeline in (1, 0):
        renderer.set_option("pipeline", pipeline)
        acc, rgba = render(renderer, ps, 6)
        assert_same_image(acc, rgba, acc_o, rgba_o)
    renderer.set_option("pipeline", 1)


def test_no_primitives(renderer, orc):
    """Empty primitive array (only the light record): every ray misses."""
    from computeraytracer_amd import cornell, scene as S
    c = cornell(40, 24)
    ps = S.PackedScene(np.zeros(0, S.PRIM_DTYPE), c.lights, c.camera, c.spectra, c.cie)
    for mode in ("none", "bvh2"):
        acc, rgba = render(renderer, ps, 2, mode)
        assert not acc.any() and not rgba[..., :3].any() and (rgba[..., 3] == 255).all()


# ------------------------------------------------------------------ ray level
def test_bvh_equals_reference_loop_ray_level(renderer, orc):
    """Closest hit through the BVH == the reference's loop over every primitive
    (GPU brute force), incl. the tie rule, exclude and NaN rays; and the GPU
    loop == the oracle's."""
    ps = _mixed_scene(64, 36)
    rng = np.random.default_rng(5)
    n = 400_000
    o = rng.uniform(-50, 600, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    # axis-parallel and grazing rays, rays on the coplanar light/ceiling pair, NaN rays
    d[:2000] = np.float32([0, 1, 0]); o[:2000, 1] = rng.uniform(0, 500, 2000)
    d[2000:3000] = np.float32([1, 0, 0]); d[3000:4000] = np.float32([0, 0, -1])
    d[4000:4100] = np.nan
    excl = np.full(n, MAXU, np.uint32)
    excl[::7] = rng.integers(0, len(ps.primitives), len(excl[::7]))
    renderer.upload(ps).build_accel("bvh2")
    bvh = renderer.debug_intersect(o, d, excl)
    renderer.build_accel("none")
    brute = renderer.debug_intersect(o, d, excl)
    hit = brute[:, 7].view(np.uint32) != MAXU
    assert np.array_equal(bvh[:, 7].view(np.uint32), brute[:, 7].view(np.uint32))
    fin = hit & ~np.isnan(d).any(1)
    assert np.array_equal(bits(bvh[fin]), bits(brute[fin]))
    assert 0.2 < hit.mean() < 1.0
    sc = orc.Scene.from_packed(ps)
    for i in list(range(0, 6000, 40)) + list(range(6000, n, 4001)):
        of, ou = sc.intersect(o[i], d[i], int(excl[i]))
        gi = int(brute[i, 7:8].view(np.uint32)[0])
        assert gi == (int(ou[1]) if ou[0] else MAXU)
        if ou[0] and not np.isnan(d[i]).any():      # NaN payloads are not part of the contract
            assert np.array_equal(bits(brute[i, :7]), bits(of))


# ------------------------------------------------------------------ properties at benchmark sizes
def test_tile_partition_is_bit_identical_1080p(renderer):
    """A tile is the same pixels of the full frame (seeds use global coordinates)."""
    from computeraytracer_amd.scenes_synth import atrium250k
    ps = atrium250k(1920, 1080)
    full_acc, full_rgba = render(renderer, ps, 2)
    for rect in [(0, 0, 1920, 135), (0, 945, 1920, 1080), (333, 217, 1001, 403)]:
        renderer.set_tile(*rect)
        renderer.frame(2).sync()
        x0, y0, x1, y1 = rect
        assert np.array_equal(bits(renderer.read_accum()), bits(full_acc[y0:y1, x0:x1]))
        assert np.array_equal(renderer.read_rgba8(), full_rgba[y0:y1, x0:x1])


def test_row_band_partition_is_bit_identical(renderer):
    """Row-interleaved partition (bands dealt round-robin): each part equals those rows of the frame."""
    from computeraytracer_amd.partition import band_rows
    from computeraytracer_amd.scenes_synth import mesh10k
    ps = mesh10k(640, 363)                       # height not a multiple of band * parts
    full_acc, full_rgba = render(renderer, ps, 3)
    for parts, band in [(4, 8), (3, 5), (8, 8)]:
        seen = np.zeros(363, bool)
        for part in range(parts):
            rows = band_rows(363, parts, part, band)
            renderer.set_row_bands(band, parts, part)
            renderer.frame(3).sync()
            acc, rgba = renderer.read_accum(), renderer.read_rgba8()
            assert acc.shape == (len(rows), 640, 4)
            assert np.array_equal(bits(acc), bits(full_acc[rows])) and np.array_equal(rgba, full_rgba[rows])
            seen[rows] = True
        assert seen.all()
    renderer.set_tile(0, 0, 640, 363)


def test_fused_samples_equal_incremental_frames(renderer):
    from computeraytracer_amd.scenes_synth import mesh10k
    ps = mesh10k(480, 270)
    a17, r17 = render(renderer, ps, 17)
    renderer.reset()
    for _ in range(17):
        renderer.frame(1)
    renderer.sync()
    assert np.array_equal(bits(renderer.read_accum()), bits(a17)) and np.array_equal(renderer.read_rgba8(), r17)
    renderer.reset().set_option("spp_per_launch", 5)
    renderer.frame(17).sync()
    renderer.set_option("spp_per_launch", 0)
    assert np.array_equal(bits(renderer.read_accum()), bits(a17)) and np.array_equal(renderer.read_rgba8(), r17)
    assert renderer.sample == 17


def test_bvh_equals_reference_loop_image_level(renderer):
    """Whole-image form of the BVH contract on a window of the 1080p frame."""
    from computeraytracer_amd.scenes_synth import atrium250k
    ps = atrium250k(1920, 1080)
    tile = (832, 476, 1088, 604)
    a_bvh, r_bvh = render(renderer, ps, 1, "bvh2", tile)
    a_ref, r_ref = render(renderer, ps, 1, "none", tile)
    assert np.array_equal(bits(a_bvh), bits(a_ref)) and np.array_equal(r_bvh, r_ref)


def test_checkpoint_resume(renderer):
    """accum + sample index is a resumable checkpoint."""
    from computeraytracer_amd import cornell
    ps = cornell(128, 128)
    a8, r8 = render(renderer, ps, 8)
    renderer.reset().frame(5).sync()
    ckpt = renderer.read_accum()
    renderer.reset()
    renderer.write_accum(ckpt, 5).frame(3).sync()
    assert np.array_equal(bits(renderer.read_accum()), bits(a8)) and np.array_equal(renderer.read_rgba8(), r8)


def test_every_pipeline_form_gives_the_same_image(renderer, orc):
    """Wavefront pipeline with quantised 4-wide nodes (default: the regrouped traversal kernel, ray ring + primitive
    tasks), the same with the first form of the traversal kernel, with quantised 8-wide nodes, with plain
    4-wide nodes, and the single-kernel form on the BVH2: one image, bit for bit (and equal to the oracle
    on a crop)."""
    ps = _mixed_scene(640, 360)
    imgs = []
    try:
        for pipeline, quant, width, form in [(1, 1, 4, 2), (1, 1, 4, 1), (1, 1, 8, 2), (1, 0, 4, 2), (0, 1, 4, 2)]:
            renderer.set_option("pipeline", pipeline).set_option("quantize", quant).set_option("wf_width", width).set_option("wf_trace_form", form)
            acc, rgba = render(renderer, ps, 5)
            imgs.append((acc, rgba))
            st = renderer.accel_stats()
            assert st["width"] == (width if pipeline else 2) and st["bytes_per_box"] == (16 if (pipeline and quant) else 32)
    finally:
        renderer.set_option("pipeline", 1).set_option("quantize", 1).set_option("wf_width", 4).set_option("wf_trace_form", 2)
    for acc, rgba in imgs[1:]:
        assert np.array_equal(bits(acc), bits(imgs[0][0])) and np.array_equal(rgba, imgs[0][1])
    rect = (300, 170, 340, 200)
    acc_o, rgba_o, _ = orc.Scene.from_packed(ps).render(5, rect=rect)
    x0, y0, x1, y1 = rect
    assert_same_image(imgs[0][0][y0:y1, x0:x1], imgs[0][1][y0:y1, x0:x1], acc_o, rgba_o, rect)


def test_counters_match_oracle_and_walked_is_a_subset(renderer, orc):
    from computeraytracer_amd.scenes_synth import mesh10k
    ps = mesh10k(192, 108)
    _, _, cnt = orc.Scene.from_packed(ps).render(3)
    renderer.upload(ps).build_accel("bvh2").enable_counters(True).reset_counters()
    renderer.frame(3).sync()
    c = renderer.counters()
    renderer.enable_counters(False)
    assert (c["rays"], c["paths"], c["bounces"], c["shadow"]) == (int(cnt[0]), int(cnt[2]), int(cnt[3]), int(cnt[4]))
    assert 0 < c["walked"] <= c["rays"] and c["rays"] - c["walked"] <= c["shadow"]


def test_two_pipes_with_long_tails_repeatable(renderer, orc):
    """Regression: cornell 1000x1000 has ~1000 paths trapped in the glass sphere until MAXDEPTH, i.e. a
    100-iteration tail.  In the tail the shade kernel walks the ray lists, where a slot with an extension
    ray AND a shadow ray appears twice; the thread holding the shadow entry once looked at the slot's flags
    to find out -- flags that the thread holding the extension entry rewrites in the same launch.  With two
    pipes on two streams (block start times spread out) that gave tens of wrong pixels per frame.  Must be
    bit-exact every time: list-walking shade or not, stragglers moved to the side pool (k_wf_evict +
    k_wf_finish) early, late or never, counting kernels or not."""
    from computeraytracer_amd import cornell
    ps = cornell()
    acc_o, rgba_o, _ = orc.Scene.from_packed(ps).render(2)
    try:
        for tail_walk, finish_at, count in [(1, 0, True), (1, 4096, True), (1, 32768, False), (0, 0, False), (0, 65536, True)]:
            renderer.set_option("wf_finish_at", finish_at).set_option("wf_pipes", 2).set_option("wf_tail_walk", tail_walk)
            for _ in range(3):
                renderer.upload(ps).build_accel("bvh2").enable_counters(count).reset_counters()
                renderer.frame(2).sync()
                assert_same_image(renderer.read_accum(), renderer.read_rgba8(), acc_o, rgba_o)
    finally:
        renderer.enable_counters(False).set_option("wf_finish_at", 32768).set_option("wf_tail_walk", 1)


def test_batches_pipelined_across_calls(renderer, orc):
    """crt_trace returns with the batch's last paths still running on the side stream (under the next call's
    pool work) and its resolve pass still to come.  Whatever the interleaving -- deferred or not, equal or
    changing batch sizes, long glass-sphere tails -- the frame after sync is the oracle's; and a buffer
    bound with bind_output holds, in stream order and without any sync, a COMPLETE earlier frame."""
    import torch
    from computeraytracer_amd import cornell
    ps = cornell(640, 640)                      # 410k pixels: pool = work, K = 2 at 2 spp
    sc = orc.Scene.from_packed(ps)
    frames = {n: sc.render(n)[:2] for n in (2, 4, 6, 8, 9)}
    try:
        for defer in (1, 0):
            renderer.upload(ps).build_accel("bvh2").set_option("wf_defer", defer)
            for _ in range(4):
                renderer.frame(2)               # no sync in between
            renderer.sync()
            assert_same_image(renderer.read_accum(), renderer.read_rgba8(), *frames[8])
            renderer.frame(1).sync()            # a different batch size continues correctly
            assert_same_image(renderer.read_accum(), renderer.read_rgba8(), *frames[9])
        # stream-ordered view of a bound output between calls: a complete frame, at most wf_ring calls old
        renderer.set_option("wf_defer", 1)
        dev = torch.device("cuda", 0)
        acc_t = torch.zeros((640, 640, 4), dtype=torch.float32, device=dev)
        rgba_t = torch.zeros((640, 640, 4), dtype=torch.uint8, device=dev)
        stream = torch.cuda.Stream(device=dev)  # (the null stream's handle is 0, which means "the context's own")
        torch.cuda.synchronize()
        renderer.set_stream(stream.cuda_stream)
        renderer.set_option("wf_cohort", 1)     # (every call its own batch: this is about the ring of batches)
        for ring in (2, 4):
            renderer.set_option("wf_ring", ring)
            acc_t.zero_(); rgba_t.zero_(); torch.cuda.synchronize()
            renderer.upload(ps).build_accel("bvh2").bind_output(acc_t.data_ptr(), rgba_t.data_ptr())
            seen = []
            with torch.cuda.stream(stream):
                for k in range(1, 5):
                    renderer.frame(2)
                    seen.append((k, rgba_t.clone()))  # enqueued on the same stream, no sync with the renderer
            renderer.sync()
            torch.cuda.synchronize()
            for k, snap in seen:
                snap = snap.cpu().numpy()
                ok = [j for j in range(max(1, k - ring), k + 1) if np.array_equal(snap, frames[2 * j][1])]
                assert ok or (k <= ring and not snap.any()), f"ring {ring}: after call {k} the bound framebuffer is no complete frame of the last {ring} calls"
        assert np.array_equal(rgba_t.cpu().numpy(), frames[8][1])
        assert np.array_equal(bits(acc_t.cpu().numpy())[..., :3], bits(frames[8][0])[..., :3])
    finally:
        renderer.set_stream(None)
        renderer.set_option("wf_defer", 1).set_option("wf_ring", 32).set_option("wf_cohort", 16)


def test_pipelined_calls_at_full_scale(renderer):
    """S2 at 1080p with the default settings (two pipes, 8 M-slot pool, four batches in flight, eviction under
    the next batches): calls of changing size back to back equal one fused call, bit for bit."""
    from computeraytracer_amd.scenes_synth import atrium250k
    ps = atrium250k(1920, 1080)
    a, r8 = render(renderer, ps, 30)
    renderer.reset()
    for n in (8, 8, 4, 1, 1, 8):
        renderer.frame(n)
    renderer.sync()
    assert renderer.sample == 30
    assert np.array_equal(bits(renderer.read_accum()), bits(a)) and np.array_equal(renderer.read_rgba8(), r8)
    renderer.set_row_bands(8, 8, 5)             # the 1/8 share of an 8-GPU run: small batches, pool = a quarter of one
    renderer.frame(12).sync()
    a12, r12 = renderer.read_accum(), renderer.read_rgba8()
    renderer.reset()
    for n in (4, 4, 2, 2):
        renderer.frame(n)
    renderer.sync()
    assert np.array_equal(bits(renderer.read_accum()), bits(a12)) and np.array_equal(renderer.read_rgba8(), r12)


def test_pipeline_state_machine_random_walk(renderer, orc):
    """Seeded random sequences of crt_trace calls of changing size, with syncs / reads in between or not, over
    random pipeline settings (pool size, pipes, chunk size, park threshold, eviction thresholds, tiles): after
    each sequence the frame is the oracle's, bit for bit.  (Several batches in flight, a work queue each, side
    pools, eviction under the next batches and the flush at sync all have to agree for that.)"""
    from computeraytracer_amd import cornell
    W = H = 448
    ps = cornell(W, H)
    sc = orc.Scene.from_packed(ps)
    full = {}
    rng = np.random.default_rng(20260104)
    renderer.upload(ps).build_accel("bvh2")
    try:
        for epoch in range(int(os.environ.get("CRT_TEST_EPOCHS", "8"))):
            opts = {"wf_pool": int(rng.choice([0, 1 << 18, 1 << 19])), "wf_pipes": int(rng.choice([1, 2, 2, 3, 4])),
                    "wf_chunk": int(rng.choice([1, 2, 4])), "wf_feed_pct": int(rng.choice([50, 100, 200])),
                    "wf_finish_at": int(rng.choice([0, 512, 32768])), "wf_flush_at": int(rng.choice([0, 64, 4096])),
                    "wf_tail_walk": int(rng.choice([0, 1])), "wf_defer": int(rng.choice([1, 1, 1, 0])),
                    "wf_ring": int(rng.choice([2, 3, 4, 32])), "wf_pool_spp": int(rng.choice([1, 2, 4])), "wf_ahead": int(rng.choice([2, 3, 6])),
                    "wf_cohort": int(rng.choice([1, 1, 4, 16])), "wf_trace_form": int(rng.choice([1, 2, 2])),
                    "wf_gen_blocks": int(rng.choice([1, 16, 128])), "wf_waves_per_cu": int(rng.choice([0, 2, 13]))}
            for k, v in opts.items():
                renderer.set_option(k, v)
            rect = None
            if rng.random() < 0.4:
                x0, y0 = int(rng.integers(0, W // 2)), int(rng.integers(0, H // 2))
                rect = (x0, y0, int(rng.integers(x0 + 40, W + 1)), int(rng.integers(y0 + 40, H + 1)))
                renderer.set_tile(*rect)
            else:
                renderer.set_tile(0, 0, W, H)
            total = 0
            for _ in range(int(rng.integers(2, 12))):
                n = int(rng.choice([1, 1, 2, 2, 3, 5]))
                renderer.frame(n)
                total += n
                what = rng.random()
                if what < 0.2:
                    renderer.sync()
                elif what < 0.3:
                    renderer.read_rgba8()
                elif what < 0.5:
                    time.sleep(float(rng.random()) * 0.003)     # (the host falls behind the device, or the other way round)
                elif what < 0.6:
                    # an option changed in the middle of the run: a flush WITHOUT a host synchronisation, then a new pool
                    renderer.set_option("wf_chunk", int(rng.choice([1, 2, 4]))).set_option("wf_ahead", int(rng.choice([2, 3, 6])))
            renderer.sync()
            if total not in full:
                full[total] = sc.render(total)[:2]
            print(epoch, opts, rect, total)          # (shown by pytest if the comparison fails)
            assert_same_image(renderer.read_accum(), renderer.read_rgba8(), *full[total], rect=rect)
            assert renderer.sample == total
    finally:
        for k, v in {"wf_pool": 0, "wf_pipes": 2, "wf_chunk": 1, "wf_feed_pct": 100, "wf_finish_at": 32768, "wf_ahead": 3,
                     "wf_flush_at": 4096, "wf_tail_walk": 1, "wf_defer": 1, "wf_ring": 32, "wf_pool_spp": 8, "wf_cohort": 16,
                     "wf_trace_form": 2, "wf_gen_blocks": 128, "wf_waves_per_cu": 0}.items():
            renderer.set_option(k, v)


def test_staging_is_never_reallocated_under_a_live_pool(renderer, orc):
    """Regression (round-2 advisor, high): a large batch after small ones kept the pool live when staging buffer 0
    was large enough, and grew the OTHER ring buffers under the batches in flight.  Here: a large batch with a ring
    of 2 (buffers 0-1 large), small batches with a ring of 32 (buffers 2-31 small), then a large batch and more small
    ones without a sync in between.  The frame is the oracle's."""
    from computeraytracer_amd import cornell
    ps = cornell(256, 256)
    sc = orc.Scene.from_packed(ps)
    try:
        renderer.upload(ps).build_accel("bvh2")
        renderer.set_option("wf_pool", 1 << 18).set_option("wf_cohort", 1).set_option("wf_ring", 2)
        renderer.frame(24).sync()
        renderer.set_option("wf_ring", 32)
        total = 24
        for n in (1, 1, 1, 1, 1, 1, 24, 1, 1, 24, 2):     # (no sync: the pool stays live across the size changes)
            renderer.frame(n)
            total += n
        renderer.sync()
        assert renderer.sample == total
        assert_same_image(renderer.read_accum(), renderer.read_rgba8(), *sc.render(total)[:2])
    finally:
        renderer.set_option("wf_pool", 0).set_option("wf_cohort", 16).set_option("wf_ring", 32)


def test_driver_rings_wrap_many_times(renderer, orc):
    """One long run of the driver (round-2 verdict, item 6): 1 spp per call, every call its own batch, four pipes,
    no sync -- 420 calls wrap the ring of batch ids (32), the status ring of every pipe (64 records) and the event
    rings behind them several times, with up to 20 batches in flight in a pool that holds 20 calls' worth of paths
    (publish, pump, evict, finish and resolve all run under back-pressure).  Bit-identical to the oracle at the end,
    and a bound output is a complete frame whenever it is looked at."""
    from computeraytracer_amd import cornell
    W = H = 160
    ps = cornell(W, H)
    sc = orc.Scene.from_packed(ps)
    try:
        renderer.upload(ps).build_accel("bvh2")
        for k, v in {"wf_pool": 1 << 19, "wf_pipes": 4, "wf_cohort": 1, "wf_ring": 32, "wf_finish_at": 2048}.items():
            renderer.set_option(k, v)
        calls = int(os.environ.get("CRT_TEST_LONG_CALLS", "420"))
        for _ in range(calls):
            renderer.frame(1)
        renderer.sync()
        assert renderer.sample == calls
        assert_same_image(renderer.read_accum(), renderer.read_rgba8(), *sc.render(calls)[:2])
    finally:
        for k, v in {"wf_pool": 0, "wf_pipes": 2, "wf_cohort": 16, "wf_ring": 32, "wf_finish_at": 32768}.items():
            renderer.set_option(k, v)


def test_gpu_lbvh_build_gives_the_same_image(renderer, orc):
    """crt_build_accel(CRT_ACCEL_LBVH): the BVH2 is built on the GPU (Morton order, Karras hierarchy).  The
    closest hit does not depend on the tree, so the image is the SAH build's bit for bit: on the mixed scene
    (patches, spheres, glass, triangles), on the 250k-triangle frame (crops), in every
    pipeline form and ray by ray; and the tree is a proper BVH over all primitives."""
    from computeraytracer_amd.scenes_synth import atrium250k
    ps = _mixed_scene(640, 360)
    a_sah, r_sah = render(renderer, ps, 4, "bvh2")
    assert renderer.accel_stats()["builder"] == "sah-host"
    a_lb, r_lb = render(renderer, ps, 4, "lbvh")
    st = renderer.accel_stats()
    assert st["builder"] == "lbvh-gpu" and st["leaves"] == len(ps.primitives) and st["nodes"] == st["leaves"] - 1
    assert 1 < st["max_depth"] <= 62
    assert np.array_equal(bits(a_lb), bits(a_sah)) and np.array_equal(r_lb, r_sah)
    try:
        for pipeline, width in [(0, 4), (1, 8)]:
            renderer.set_option("pipeline", pipeline).set_option("wf_width", width)
            a2, r2 = render(renderer, ps, 4, "lbvh")
            assert np.array_equal(bits(a2), bits(a_sah)) and np.array_equal(r2, r_sah)
    finally:
        renderer.set_option("pipeline", 1).set_option("wf_width", 4)
    big = atrium250k(1920, 1080)
    tile = (800, 400, 1120, 640)
    a1, r1 = render(renderer, big, 2, "bvh2", tile)
    a2, r2 = render(renderer, big, 2, "lbvh", tile)
    assert renderer.accel_stats()["leaves"] == len(big.primitives)
    assert np.array_equal(bits(a1), bits(a2)) and np.array_equal(r1, r2)
    # ... and the oracle's (the reference loop on the CPU) on a crop of that window
    rect = (940, 600, 972, 616)
    acc_o, rgba_o, _ = orc.Scene.from_packed(big).render(2, rect=rect)
    cx0, cy0 = rect[0] - tile[0], rect[1] - tile[1]
    assert_same_image(a2[cy0:cy0 + 16, cx0:cx0 + 32], r2[cy0:cy0 + 16, cx0:cx0 + 32], acc_o, rgba_o, rect)
    # ray level, against the reference loop on the GPU
    rng = np.random.default_rng(5)
    o = rng.uniform(-1.0, 1.0, (20000, 3)).astype(np.float32) * np.float32(2.0)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    renderer.upload(big).build_accel("lbvh")
    h_lb = renderer.debug_intersect(o, d)
    renderer.build_accel("none")
    h_ref = renderer.debug_intersect(o, d)
    assert np.array_equal(bits(h_lb), bits(h_ref))


def test_one_sample_per_pixel_and_tiny_tiles(renderer, orc):
    """Pool larger than the work (1 spp on a small tile) and a 1x1 tile."""
    from computeraytracer_amd import cornell
    ps = cornell(100, 60)
    acc_o, rgba_o, _ = orc.Scene.from_packed(ps).render(1)
    acc, rgba = render(renderer, ps, 1)
    assert_same_image(acc, rgba, acc_o, rgba_o)
    acc, rgba = render(renderer, ps, 1, tile=(37, 21, 38, 22))
    assert acc.shape == (1, 1, 4) and np.array_equal(bits(acc[0, 0, :3]), bits(acc_o[21, 37, :3]))
    acc, rgba = render(renderer, ps, 2, tile=(5, 5, 5, 9))          # empty tile
    assert acc.size == 0


def test_mesh_scene_and_python_cli(renderer, orc, tmp_path):
    """OBJ mesh + glass sphere scene (scenes/cornell_mesh.json) vs the oracle; CLI writes PNG + checkpoint."""
    import subprocess
    import sys
    from conftest import ROOT
    from computeraytracer_amd import image, scene as S
    path = os.path.join(ROOT, "scenes", "cornell_mesh.json")
    sc = S.load_scene(path)
    sc["camera"]["width"], sc["camera"]["height"] = 96, 96
    ps = S.pack_scene(sc, base_dir=os.path.join(ROOT, "scenes"))
    acc_o, rgba_o, _ = orc.Scene.from_packed(ps).render(6)
    acc, rgba = render(renderer, ps, 6)
    assert_same_image(acc, rgba, acc_o, rgba_o)
    ck = str(tmp_path / "ck.npz")
    for spp in (2, 4):                               # 2 samples, then resume for 4 more = the 6 above
        subprocess.run([sys.executable, "-m", "computeraytracer_amd", "--scene", path, "--width", "96", "--height", "96",
                        "--spp", str(spp), "--out", str(tmp_path / "o.png"), "--checkpoint", ck], cwd=ROOT, check=True,
                       capture_output=True)
    d = np.load(ck)
    assert int(d["sample"]) == 6 and np.array_equal(bits(d["accum"]), bits(acc))
    assert (tmp_path / "o.png").read_bytes()[:4] == b"\x89PNG"


# ------------------------------------------------------------------ BASELINE configs 4 and 5 at their real sizes
def test_c4_4k_row_band_share_vs_oracle(renderer, orc):
    """BASELINE config 4: S2 at 3840 x 2160, the frame dealt to 8 GPUs in bands of 8 rows.  The full 4K frame
    equals the oracle on three crops, and a rank's share (crt_set_row_bands(8, 8, k)) equals those rows of it."""
    from computeraytracer_amd.partition import band_rows
    from computeraytracer_amd.scenes_synth import atrium250k
    ps = atrium250k(3840, 2160)
    full_acc, full_rgba = render(renderer, ps, 2)
    assert full_acc.shape == (2160, 3840, 4)
    sc = orc.Scene.from_packed(ps)
    for rect in [(1880, 1200, 1912, 1216), (1400, 800, 1432, 816), (2200, 1640, 2232, 1656)]:
        acc_o, rgba_o, _ = sc.render(2, rect=rect)
        x0, y0, x1, y1 = rect
        assert_same_image(full_acc[y0:y1, x0:x1], full_rgba[y0:y1, x0:x1], acc_o, rgba_o, rect)
    for part in (0, 3, 7):
        rows = band_rows(2160, 8, part, 8)
        renderer.set_row_bands(8, 8, part)
        renderer.frame(2).sync()
        acc, rgba = renderer.read_accum(), renderer.read_rgba8()
        assert acc.shape == (len(rows), 3840, 4) and len(rows) in (264, 272)
        assert np.array_equal(bits(acc), bits(full_acc[rows])) and np.array_equal(rgba, full_rgba[rows])
    renderer.set_tile(0, 0, 3840, 2160)


def test_c5_soup_10M_triangles_full_size(renderer, orc):
    """BASELINE config 5 at its real size: 10 000 000 random triangles (a 1.4 GB tree, depth-28+ stacks, the leaf
    encoding at 24-bit slot numbers), both builders.  Ray level: 20 000 random rays through the BVH == the reference
    loop on the GPU, 200 of them == the oracle's loop.  Image level: a 16 x 16 crop at 1 spp == the oracle; a window
    through the wavefront pipeline == the single-kernel form; a tile == the same pixels of the 1080p frame."""
    from computeraytracer_amd.scenes_synth import soup
    ps = soup(10_000_000, 1920, 1080)
    assert len(ps.primitives) == 6 + 10_000_000
    sc = orc.Scene.from_packed(ps)
    rng = np.random.default_rng(11)
    n = 20_000
    o = rng.uniform(-20, 575, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    excl = np.full(n, MAXU, np.uint32)
    excl[::5] = rng.integers(0, len(ps.primitives), len(excl[::5]))
    crop = (952, 532, 968, 548)
    acc_o, rgba_o, _ = sc.render(1, rect=crop)
    window = (832, 476, 1088, 604)
    renderer.upload(ps)
    got = {}
    for mode in ("bvh2", "lbvh"):
        renderer.build_accel(mode)
        st = renderer.accel_stats()
        assert st["builder"] == ("sah-host" if mode == "bvh2" else "lbvh-gpu") and st["max_depth"] <= 62
        got[mode] = renderer.debug_intersect(o, d, excl)
        renderer.set_tile(*crop)
        renderer.frame(1).sync()
        assert_same_image(renderer.read_accum(), renderer.read_rgba8(), acc_o, rgba_o, crop)
        renderer.set_tile(*window)
        renderer.frame(1).sync()
        got[mode + "_win"] = (renderer.read_accum(), renderer.read_rgba8())
        renderer.set_tile(0, 0, 1920, 1080)
    # (the GPU LBVH build is the last one built) whole frame: tile invariance, and the single-kernel form on the window
    renderer.frame(1).sync()
    full_acc, full_rgba = renderer.read_accum(), renderer.read_rgba8()
    x0, y0, x1, y1 = window
    for key in ("bvh2_win", "lbvh_win"):
        assert np.array_equal(bits(got[key][0]), bits(full_acc[y0:y1, x0:x1])) and np.array_equal(got[key][1], full_rgba[y0:y1, x0:x1])
    try:
        renderer.set_option("pipeline", 0)
        renderer.set_tile(*window)
        renderer.frame(1).sync()
        assert np.array_equal(bits(renderer.read_accum()), bits(full_acc[y0:y1, x0:x1]))
    finally:
        renderer.set_option("pipeline", 1)
        renderer.set_tile(0, 0, 1920, 1080)
    renderer.build_accel("none")
    brute = renderer.debug_intersect(o, d, excl)
    hit = brute[:, 7].view(np.uint32) != MAXU
    assert 0.5 < hit.mean() <= 1.0
    for mode in ("bvh2", "lbvh"):
        assert np.array_equal(got[mode][:, 7].view(np.uint32), brute[:, 7].view(np.uint32))
        assert np.array_equal(bits(got[mode][hit]), bits(brute[hit]))
    for i in range(0, n, 100):
        of, ou = sc.intersect(o[i], d[i], int(excl[i]))
        gi = int(brute[i, 7:8].view(np.uint32)[0])
        assert gi == (int(ou[1]) if ou[0] else MAXU)
        if ou[0]:
            assert np.array_equal(bits(brute[i, :7]), bits(of))


# ------------------------------------------------------------------ non-finite shadow rays
def _nan_light_scene(w, h, light_last):
    """Cornell's patches (no spheres) with the light record's origin at +inf: every light sample gives a NaN light
    direction.  Under the reference's reject-form tests (ComputeShader.wgsl:546,557,566) a NaN ray passes every test
    of every patch, so "the closest hit" of shadow_intersect is the LAST patch of the array that is not excluded:
    with light_last the light itself (NEE term = NaN, added), otherwise an ordinary wall (blocked)."""
    from computeraytracer_amd import cornell, scene as S
    c = cornell(w, h)
    keep = np.flatnonzero(c.primitives["category"] == 0)
    if light_last:
        is_light = c.primitives["data4"][keep, 2] == 1
        keep = np.concatenate([keep[~is_light], keep[is_light]])
    prims = np.zeros(len(keep), S.PRIM_DTYPE)            # (indexing into a fresh array keeps the 80-byte stride)
    prims[:] = c.primitives[keep]
    prims["data4"][:, 3] = np.arange(len(prims))
    src = S.lights_of(prims)
    lights = np.zeros(len(src), S.PRIM_DTYPE)
    lights[:] = src
    lights["data1"][:, 0] = np.inf
    return S.PackedScene(prims, lights, c.camera, c.spectra, c.cie)


@pytest.mark.parametrize("light_last", [True, False])
def test_non_finite_shadow_rays_follow_the_reference_loop(renderer, orc, light_last):
    """One decision for every code path: the wavefront pipeline, the single-kernel form and the reference loop on
    the GPU all give the oracle's image (NaN pixels where the oracle has NaN, bit-identical elsewhere)."""
    ps = _nan_light_scene(96, 96, light_last)
    acc_o, rgba_o, _ = orc.Scene.from_packed(ps).render(3)
    nan_o = np.isnan(acc_o[..., :3]).any(-1)
    assert nan_o.mean() > 0.5 if light_last else not nan_o.any()
    try:
        for pipeline, mode in [(1, "bvh2"), (0, "bvh2"), (1, "none"), (1, "lbvh")]:
            renderer.set_option("pipeline", pipeline)
            acc, rgba = render(renderer, ps, 3, mode)
            nan_g = np.isnan(acc[..., :3]).any(-1)
            assert np.array_equal(nan_g, nan_o), (pipeline, mode)
            assert np.array_equal(bits(acc[~nan_o])[..., :3], bits(acc_o[~nan_o])[..., :3]), (pipeline, mode)
            assert np.array_equal(rgba, rgba_o), (pipeline, mode)
    finally:
        renderer.set_option("pipeline", 1)


def test_non_finite_camera_follows_the_reference_loop(renderer, orc):
    """A camera whose eye is at infinity: every camera ray is non-finite, and the reference's reject-form tests let a
    NaN pass every patch and sphere (:546,557,566,605,609), so `intersect` returns the last primitive of the array --
    decided by the reference loop itself in every pipeline form (k_wf_gen flags such rays, the next shade step's
    write-back resolves them)."""
    from computeraytracer_amd import cornell, scene as S
    c = cornell(48, 40)
    cam = c.camera.copy()
    cam[0] = np.inf
    ps = S.PackedScene(c.primitives, c.lights, cam, c.spectra, c.cie)
    acc_o, rgba_o, _ = orc.Scene.from_packed(ps).render(3)
    nan_o = np.isnan(acc_o[..., :3]).any(-1)
    try:
        for pipeline, mode in [(1, "bvh2"), (0, "bvh2"), (1, "none")]:
            renderer.set_option("pipeline", pipeline)
            acc, rgba = render(renderer, ps, 3, mode)
            assert np.array_equal(np.isnan(acc[..., :3]).any(-1), nan_o), (pipeline, mode)
            assert np.array_equal(bits(acc[~nan_o])[..., :3], bits(acc_o[~nan_o])[..., :3]), (pipeline, mode)
            assert np.array_equal(rgba, rgba_o), (pipeline, mode)
    finally:
        renderer.set_option("pipeline", 1)


def test_out_of_memory_is_reported_and_the_context_recovers(orc):
    """A failed device allocation (injected: option debug_fail_alloc = k fails the k-th one) gives CRT_ENOMEM, never a
    launch on a null pointer; the same context renders correctly afterwards."""
    from computeraytracer_amd import Renderer, cornell
    from computeraytracer_amd._lib import CrtError
    ps = cornell(96, 64)
    acc_o, rgba_o, _ = orc.Scene.from_packed(ps).render(2)
    for k in (1, 3, 9, 12):                           # ray lists, pool arrays, staging buffers, ... of a fresh context
        r = Renderer(0)
        try:
            r.upload(ps).build_accel("bvh2")
            r.set_option("debug_fail_alloc", k)
            with pytest.raises(CrtError) as e:
                r.frame(2).sync()                    # (a small call may wait to be merged with the next ones: sync publishes it)
            assert e.value.code == -4, str(e.value)
            assert r.sample == 0
            r.frame(2).sync()
            assert_same_image(r.read_accum(), r.read_rgba8(), acc_o, rgba_o)
        finally:
            r.set_option("debug_fail_alloc", 0)
            r.close()
    r = Renderer(0)
    try:
        r.set_option("debug_fail_alloc", 1)           # ... and while a scene is being uploaded
        with pytest.raises(CrtError):
            r.upload(ps)
        with pytest.raises(CrtError, match="upload a scene first"):
            r.frame(1)
        r.set_option("debug_fail_alloc", 0)
        r.upload(ps).build_accel("bvh2").frame(2).sync()
        assert_same_image(r.read_accum(), r.read_rgba8(), acc_o, rgba_o)
    finally:
        r.set_option("debug_fail_alloc", 0)
        r.close()
    # ... and while the acceleration structure is being built (either builder): the previous structure's arrays are
    # gone by then, so the context must refuse to trace until a build has succeeded
    for mode in ("bvh2", "lbvh"):
        for k in (1, 2, 4):
            r = Renderer(0)
            try:
                r.upload(ps).build_accel("bvh2").frame(1).sync()
                r.reset()
                r.set_option("debug_fail_alloc", k)
                with pytest.raises(CrtError) as e:
                    r.build_accel(mode)
                assert e.value.code == -4, str(e.value)
                r.set_option("debug_fail_alloc", 0)
                with pytest.raises(CrtError, match="crt_build_accel first"):
                    r.frame(1)
                r.build_accel(mode).frame(2).sync()
                assert_same_image(r.read_accum(), r.read_rgba8(), acc_o, rgba_o)
            finally:
                r.set_option("debug_fail_alloc", 0)
                r.close()


def test_every_frame_of_a_pipelined_run_can_be_shown_once(renderer, orc):
    """The display step of the reference's loop (src/main.js:597-620 shows every sample's frame) without stopping the
    pipeline: with option frame_ring the resolve pass keeps the rgba8 frame of every sample; crt_read_sample_rgba8(k)
    returns frame k -- bit-identical to the oracle's frame after k samples -- while calls run ahead of the reads and are
    merged into cohorts; crt_read_latest_rgba8 returns a complete frame and its index without flushing."""
    from computeraytracer_amd import cornell
    from computeraytracer_amd._lib import CrtError
    W = H = 200
    ps = cornell(W, H)
    sc = orc.Scene.from_packed(ps)
    want = {}
    try:
        renderer.upload(ps).build_accel("bvh2").set_option("frame_ring", 24).set_option("wf_cohort", 8)
        lag, frames, shown = 10, 40, []
        for k in range(1, frames + 1):
            renderer.frame(1)
            if k > lag:
                shown.append((k - lag, renderer.read_sample_rgba8(k - lag)))
            if k == 25:
                img, s = renderer.read_latest_rgba8()                 # no flush: some complete frame at most 25 samples old
                assert 0 < s <= 25 and renderer.sample == 25
                want[s] = want.get(s) or sc.render(s)[1]
                assert np.array_equal(img, want[s])
        for k in range(frames - lag + 1, frames + 1):
            shown.append((k, renderer.read_sample_rgba8(k)))
        assert [k for k, _ in shown] == list(range(1, frames + 1))
        for k, img in shown:
            if k in (1, 2, 7, 8, 9, 16, 17, 31, 39, 40):             # (the oracle renders k samples from scratch each time)
                assert np.array_equal(img, sc.render(k)[1]), k
        with pytest.raises(CrtError, match="left the ring"):
            renderer.read_sample_rgba8(5)
        renderer.sync()
        assert renderer.latest_sample == frames
        assert_same_image(renderer.read_accum(), renderer.read_rgba8(), *sc.render(frames)[:2])
    finally:
        renderer.set_option("frame_ring", 0).set_option("wf_cohort", 16)


def test_display_state_machine_random_walk(renderer, orc):
    """Seeded random sequences of small crt_trace calls with the non-flushing reads in between -- crt_read_sample_rgba8 of a
    random frame still in the ring, crt_read_latest_rgba8, crt_latest_sample -- over random driver settings (pipes, batch
    ring, cohort, frame ring, chunk size, eviction thresholds): every frame handed out is the oracle's frame of that sample
    index, bit for bit, the indices crt_latest_sample reports never go back, and the run ends on the oracle's frame."""
    from computeraytracer_amd import cornell
    W = H = 160                                     # (large enough for two and four pipes when the ring allows: 25 600 x 31 work items)
    ps = cornell(W, H)
    sc = orc.Scene.from_packed(ps)
    frames = {}

    def want(k):
        if k not in frames:
            frames[k] = sc.render(k)[1]
        return frames[k]

    rng = np.random.default_rng(20261005)
    renderer.upload(ps).build_accel("bvh2")
    try:
        for epoch in range(int(os.environ.get("CRT_TEST_EPOCHS", "8"))):
            opts = {"wf_pipes": int(rng.choice([1, 2, 2, 4])), "wf_ring": int(rng.choice([2, 3, 32])), "wf_cohort": int(rng.choice([1, 4, 8, 16])),
                    "frame_ring": int(rng.choice([8, 17, 64])), "wf_chunk": int(rng.choice([1, 2, 4])), "wf_pool": int(rng.choice([0, 1 << 16, 1 << 19, 1 << 20])),
                    "wf_finish_at": int(rng.choice([0, 512, 32768])), "wf_flush_at": int(rng.choice([0, 64, 4096]))}
            for k, v in opts.items():
                renderer.set_option(k, v)
            renderer.reset()
            total, latest_seen, log = 0, 0, []
            for _ in range(int(rng.integers(6, 40))):
                n = int(rng.choice([1, 1, 1, 2, 3]))
                if total + n > 40:
                    break
                renderer.frame(n)
                total += n
                what = rng.random()
                if what < 0.35:
                    lo = max(1, total - opts["frame_ring"] + 1)
                    k = int(rng.integers(lo, total + 1))
                    log.append(("sample", k))
                    assert np.array_equal(renderer.read_sample_rgba8(k), want(k)), (epoch, opts, log)
                elif what < 0.5:
                    img, s = renderer.read_latest_rgba8()
                    log.append(("latest", s))
                    assert latest_seen <= s <= total, (epoch, opts, log)
                    latest_seen = s
                    if s:
                        assert np.array_equal(img, want(s)), (epoch, opts, log)
                elif what < 0.6:
                    s = renderer.latest_sample
                    log.append(("query", s))
                    assert latest_seen <= s <= total, (epoch, opts, log)
                    latest_seen = s
                elif what < 0.65:
                    renderer.sync()
                    log.append(("sync", total))
                    assert renderer.latest_sample == total
                    latest_seen = total
                elif what < 0.8:
                    time.sleep(float(rng.random()) * 0.003)
                elif what < 0.85:
                    renderer.set_option("wf_chunk", int(rng.choice([1, 2, 4])))     # (a flush without a host synchronisation, then a new pool)
                    log.append(("option", total))
            renderer.sync()
            print(epoch, opts, total, log)               # (shown by pytest if a comparison fails)
            assert renderer.sample == total and renderer.latest_sample == total
            if total:
                assert np.array_equal(renderer.read_rgba8(), want(total))
                lo = max(1, total - opts["frame_ring"] + 1)
                assert np.array_equal(renderer.read_sample_rgba8(lo), want(lo))
    finally:
        for k, v in {"wf_pipes": 2, "wf_ring": 32, "wf_cohort": 16, "frame_ring": 0, "wf_chunk": 1, "wf_pool": 0, "wf_finish_at": 32768,
                     "wf_flush_at": 4096}.items():
            renderer.set_option(k, v)


def test_flush_without_host_sync_then_quick_batches(renderer, orc):
    """Regression (found by the walk above): a flush that the driver makes on its own (here: a one-sample batch after
    three-sample ones wants a pool less than half the size) returns with the last batches' finish / resolve passes still
    queued on the context's stream; the queue / side-counter reset of the NEW run's second batch, on the publishing stream,
    overtook them, k_wf_finish found side_count 0 and the paths evicted at the flush were lost (two to fifty pixels short of
    a sample, two runs in three).  The exact sequence, six times: every frame is the oracle's."""
    from computeraytracer_amd import cornell
    W = H = 96
    ps = cornell(W, H)
    sc = orc.Scene.from_packed(ps)
    want = {k: sc.render(k)[1] for k in (1, 3, 17, 24)}
    acc24 = sc.render(24)[0]
    renderer.upload(ps).build_accel("bvh2")
    opts = {"wf_pipes": 2, "wf_ring": 3, "wf_cohort": 1, "frame_ring": 17, "wf_chunk": 4, "wf_finish_at": 0, "wf_flush_at": 256}
    try:
        for k, v in opts.items():
            renderer.set_option(k, v)
        for rep in range(6):
            renderer.reset()
            renderer.frame(2).frame(1).frame(1).frame(1).sync()
            renderer.frame(3)
            renderer.frame(1).frame(1).frame(1).sync()                        # 11
            renderer.frame(3)
            assert np.array_equal(renderer.read_sample_rgba8(3), want[3])     # (a read that does not flush: the host falls behind)
            renderer.frame(3)
            assert np.array_equal(renderer.read_sample_rgba8(1), want[1])
            renderer.frame(1).frame(1).frame(1)                               # 18 (the pool shrinks: flush), 19, 20
            assert np.array_equal(renderer.read_sample_rgba8(17), want[17]), rep
            renderer.frame(2).frame(2).sync()
            assert renderer.sample == 24
            assert_same_image(renderer.read_accum(), renderer.read_rgba8(), acc24, want[24])
    finally:
        for k, v in {"wf_pipes": 2, "wf_ring": 32, "wf_cohort": 16, "frame_ring": 0, "wf_chunk": 1, "wf_finish_at": 32768, "wf_flush_at": 4096}.items():
            renderer.set_option(k, v)


# ------------------------------------------------------------------ multi-GPU through the C ABI
@pytest.mark.parametrize("world,band", [(2, 8), (3, 0), (4, 5)])
def test_native_gather_assembles_the_single_gpu_frame(orc, world, band):
    """The multi-GPU path of the C ABI (crt_comm_*, crt_gather, crt_read_frame_*) on ONE GPU: `world` contexts on device 0
    joined by the in-process transport (device-to-device copies stand where RCCL's all-gather does; partition, strip
    buffers, stream ordering and assembly are the same code).  Each rank renders its rows; the assembled frame equals
    the oracle's, bit for bit, on every rank -- mid-run gathers of a pipelined loop included."""
    from computeraytracer_amd import Renderer, cornell
    W, H, spp = 136, 93, 3                           # (H not a multiple of the band, nor of the world size)
    ps = cornell(W, H)
    acc_o, rgba_o, _ = orc.Scene.from_packed(ps).render(2 * spp)
    cid = Renderer.comm_unique_id(local=True)
    rs = [Renderer(0) for _ in range(world)]
    try:
        for k, r in enumerate(rs):
            r.comm_init(cid, k, world).upload(ps).comm_partition(band).build_accel("bvh2")
            assert r.comm_info() == dict(rank=k, world=world, transport="local", rows=r.tile[3])
        assert sum(r.tile[3] for r in rs) == H
        for r in rs:
            r.frame(spp).gather(rgba8=True)                          # a gather in the middle of the run (no sync)
        for r in rs:
            r.frame(spp).sync().gather(rgba8=True, accum=True)
        for r in rs:                                                 # all_gather: every rank holds the frame
            assert_same_image(r.read_frame_accum(), r.read_frame_rgba8(), acc_o, rgba_o)
        with pytest.raises(Exception, match="has not posted"):
            rs[0].gather(rgba8=True).read_frame_rgba8()              # a rank alone cannot read a frame the others did not post
    finally:
        for r in rs:
            r.close()


def test_native_gather_random_walk_shows_only_complete_frames(orc):
    """The gather path under a display loop's pacing: seeded random sequences in which the ranks (contexts on device 0,
    in-process transport) trace at different paces, gather in the middle of the run in random rank order, and a random
    rank reads the assembled frame.  What crt_trace promises for bound outputs must hold row by row: each rank's rows
    of an assembled frame are the oracle's frame after SOME number of samples that rank had been asked for by then --
    complete, never a half-resolved one, and never older than the rank's previous gather showed; after a sync of every
    rank the frame is the oracle's at the full count."""
    from computeraytracer_amd import Renderer, cornell
    from computeraytracer_amd.partition import strip_rows, band_rows
    W, H, K = 64, 50, 20
    ps = cornell(W, H)
    sc = orc.Scene.from_packed(ps)
    frames = {0: np.zeros((H, W, 4), np.uint8)}
    for k in range(1, K + 1):
        frames[k] = sc.render(k)[1]
    rng = np.random.default_rng(20261006)
    for epoch in range(int(os.environ.get("CRT_TEST_EPOCHS", "8")) // 2):
        world, band = int(rng.choice([2, 3])), int(rng.choice([0, 4, 8]))
        cid = Renderer.comm_unique_id(local=True)
        rs = [Renderer(0) for _ in range(world)]
        try:
            for k, r in enumerate(rs):
                r.comm_init(cid, k, world).upload(ps).comm_partition(band).build_accel("bvh2")
                r.set_option("wf_cohort", int(rng.choice([1, 2, 16]))).set_option("wf_pool", int(rng.choice([0, 1 << 14])))
            rows = [np.arange(*strip_rows(H, world, k)) if band == 0 else band_rows(H, world, k, band) for k in range(world)]
            asked, shown = [0] * world, [0] * world
            for _ in range(int(rng.integers(4, 14))):
                for k in rng.permutation(world):
                    n = int(rng.choice([0, 1, 1, 2, 3]))
                    if asked[k] + n <= K and n:
                        rs[k].frame(n)
                        asked[k] += n
                    if rng.random() < 0.3:
                        time.sleep(float(rng.random()) * 0.002)
                if rng.random() < 0.2:
                    for r in rs:
                        r.sync()
                for k in rng.permutation(world):
                    rs[k].gather(rgba8=True)
                img = rs[int(rng.integers(0, world))].read_frame_rgba8()
                for k in range(world):
                    if len(rows[k]) == 0:
                        continue
                    match = [j for j in range(shown[k], asked[k] + 1) if np.array_equal(img[rows[k]], frames[j][rows[k]])]
                    assert match, (epoch, world, band, k, shown, asked)
                    shown[k] = match[0]
            for r in rs:
                r.sync()
            for k in rng.permutation(world):
                rs[k].gather(rgba8=True, accum=True)
            total = asked[0]
            if all(a == total for a in asked) and total:
                assert_same_image(rs[0].read_frame_accum(), rs[0].read_frame_rgba8(), *sc.render(total)[:2])
            else:
                img = rs[-1].read_frame_rgba8()
                for k in range(world):
                    if len(rows[k]):
                        assert np.array_equal(img[rows[k]], frames[asked[k]][rows[k]]), (epoch, k, asked)
        finally:
            for r in rs:
                r.close()


def test_comm_init_failures_leave_nothing_behind():
    """A failed crt_comm_init (rank taken, world sizes that disagree, a second communicator on one context) leaves neither a
    member in the id's group nor a communicator on the context: the rank can be taken afterwards, and the group still works."""
    from computeraytracer_amd import Renderer, cornell
    from computeraytracer_amd._lib import CrtError
    ps = cornell(48, 40)
    cid = Renderer.comm_unique_id(local=True)
    a, b, c = Renderer(0), Renderer(0), Renderer(0)
    try:
        a.comm_init(cid, 0, 2)
        with pytest.raises(CrtError, match="rank of the id is taken"):
            b.comm_init(cid, 0, 2)
        with pytest.raises(CrtError, match="disagree about the world size"):
            b.comm_init(cid, 1, 3)
        with pytest.raises(CrtError, match="already has a communicator"):
            a.comm_init(cid, 1, 2)
        assert b.comm_info()["world"] == 1                         # (no communicator on b)
        b.comm_init(cid, 1, 2)                                     # the rank is still free
        for r in (a, b):
            r.upload(ps).comm_partition(8).build_accel("bvh2")
        for r in (a, b):
            r.frame(2).sync().gather(rgba8=True)
        c.upload(ps).build_accel("bvh2")
        c.frame(2)
        assert np.array_equal(a.read_frame_rgba8(), c.read_rgba8()) and np.array_equal(b.read_frame_rgba8(), c.read_rgba8())
    finally:
        for r in (a, b, c):
            r.close()


def test_native_gather_over_rccl_single_rank():
    """The RCCL transport itself on the one GPU the test box has: librccl is loaded on demand, ncclCommInitRank and
    ncclAllGather run with world = 1, and the frame read through the communicator equals the context's own."""
    from computeraytracer_amd import Renderer, cornell
    ps = cornell(64, 48)
    r = Renderer(0)
    try:
        r.comm_init(Renderer.comm_unique_id(local=False), 0, 1).upload(ps).comm_partition(8).build_accel("bvh2")
        assert r.comm_info()["transport"] == "rccl"
        r.frame(2).sync().gather(rgba8=True, accum=True)
        a, g = r.read_frame_accum(), r.read_frame_rgba8()
        assert np.array_equal(bits(a), bits(r.read_accum())) and np.array_equal(g, r.read_rgba8())
    finally:
        r.close()


# ------------------------------------------------------------------ error behaviour
def test_errors_are_reported_not_fatal():
    from computeraytracer_amd import Renderer, cornell, scene as S
    from computeraytracer_amd._lib import CrtError
    r = Renderer(0)
    with pytest.raises(CrtError, match="upload a scene first"):
        r.frame(1)
    ps = cornell(32, 32)
    r.upload(ps)
    with pytest.raises(CrtError, match="crt_build_accel first"):
        r.frame(1)
    bad = ps.primitives.copy()
    bad["data4"][3, 3] = 9
    with pytest.raises(CrtError, match="must equal the array position"):
        r.upload(S.PackedScene(bad, ps.lights, ps.camera, ps.spectra, ps.cie))
    with pytest.raises(CrtError, match="outside"):
        r.upload(ps).set_tile(0, 0, 64, 8)
    r.upload(ps).build_accel("bvh2").frame(1).sync()       # still usable afterwards
    assert r.read_rgba8().shape == (32, 32, 4)
    r.close()
