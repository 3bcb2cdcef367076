"""Pins for the CPU oracle (CPU-only).

The reference has no tests, golden images or known-answer vectors
(package.json:9), so the float path is "parity unpinned" against a real WGSL
runtime.  What can be pinned independently of any float implementation is
pinned here, with the values derived from the WGSL/JS text in SURVEY.md 8c:
the integer RNG, the packed buffer sizes and the table checksums.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits

# SURVEY.md 8c
TEA_KAT = {(0, 0): 0x741C187D, (1, 100): 0x86664F3A, (3, 500): 0xD350E6F6, (1919, 107900): 0xC0F8545B}
RAND_KAT = {
    (0, 0, 1): [3982936, 1279938, 10369472, 15862708],
    (3, 5, 1): [11592416, 12499111, 7622025, 7224101],
    (999, 999, 1): [1147759, 7560048, 8088559, 11396148],
    (3, 5, 2): [2020326, 11129530, 11769162, 16104197],
    (1919, 1079, 64): [1288756, 5462617, 9844093, 10010033],
}
SEED_AFTER_4 = [0x80F20BB4, 0x23662435, 0x85F38847, 0x07A86206]
SPECTRA_SHA = "6ef5ac501722cf5cf44402d61647035838521c6d761a543b69b4887c679ffa8b"
CIE_SHA = "965e386c9f38c3f54e70cff8e2416f851ba5a44566e512033bd0166490eed13b"
ROW_SUMS = [218.0720003, 59.8915000, 72.8329998, 5046.6999969, 0.0, 8370.3499303, 11.5050000]


def test_tea_kat(orc):
    for (a, b), want in TEA_KAT.items():
        assert orc.tea(a, b) == want


def test_rand_kat(orc):
    for (x, y, s), want in RAND_KAT.items():
        got, seed = orc.rand_kat(x, y, s, 4)
        assert list(map(int, got)) == want
        if (x, y, s) == (0, 0, 1):
            assert list(map(int, seed)) == SEED_AFTER_4


def test_constants():
    assert float(np.float32(0x7F800000)) == 2139095040.0          # Q1: INFINITY is an integer cast
    assert float(np.float32(3.14159265359)) == 3.1415927410125732


def test_buffer_sizes_and_table_hashes(golden_buffers):
    b = golden_buffers
    assert b["primitives"].nbytes == 1440 and b["patches"].nbytes == 1024 and b["lights"].nbytes == 80
    assert b["camera"].nbytes == 64 and b["spectra"].nbytes == 8428 and b["cie"].nbytes == 5652
    assert hashlib.sha256(b["spectra"].tobytes()).hexdigest() == SPECTRA_SHA
    assert hashlib.sha256(b["cie"].tobytes()).hexdigest() == CIE_SHA
    np.testing.assert_allclose(b["spectra"].astype(np.float64).sum(1), ROW_SUMS, rtol=0, atol=5e-7)
    assert list(b["spectra"][3, :3]) == [np.float32(15.0), np.float32(15.03), np.float32(15.06)]
    # integ constant of spectral_to_xyz is the sum of ALL 471 CIE_Y samples (Q10)
    assert abs(float(b["cie"][1].astype(np.float64).sum()) - 106.856895) < 1e-4


def test_math_spec_accuracy(orc):
    """The fixed polynomial kernels stay within a few ulp of the true functions,
    i.e. far inside WGSL's accuracy envelope (sin/cos abs err 2^-11, exp 3+2|x| ulp, ...)."""
    rng = np.random.default_rng(7)

    def ulp_err(got, ref64):
        ref32 = ref64.astype(np.float32)
        u = np.abs(np.spacing(ref32)).astype(np.float64)
        return np.max(np.abs(got.astype(np.float64) - ref64) / u)

    x = rng.uniform(0, 2 * np.pi, 200000).astype(np.float32)
    assert np.max(np.abs(orc.math_eval("sin", x).astype(np.float64) - np.sin(x.astype(np.float64)))) < 2e-7
    assert np.max(np.abs(orc.math_eval("cos", x).astype(np.float64) - np.cos(x.astype(np.float64)))) < 2e-7
    x = rng.uniform(-80, 80, 200000).astype(np.float32)
    assert ulp_err(orc.math_eval("exp", x), np.exp(x.astype(np.float64))) < 2.0
    x = np.exp(rng.uniform(-80, 80, 200000)).astype(np.float32)
    ref = np.log2(x.astype(np.float64))
    assert np.max(np.abs(orc.math_eval("log2", x).astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0)) < 1.5e-7
    x = rng.uniform(-120, 120, 200000).astype(np.float32)
    assert ulp_err(orc.math_eval("exp2", x), np.exp2(x.astype(np.float64))) < 2.0
    x = rng.uniform(0.0031308, 1.0, 200000).astype(np.float32)          # the gamma call site
    y = np.full_like(x, np.float32(1.0 / 2.4))
    ref = np.power(x.astype(np.float64), y.astype(np.float64))
    assert np.max(np.abs(orc.math_eval("pow", x, y).astype(np.float64) - ref) / ref) < 1e-6
    x = np.exp(rng.uniform(-3, 8, 200000)).astype(np.float32)           # pow(distance, 2)
    ref = x.astype(np.float64) ** 2
    assert np.max(np.abs(orc.math_eval("pow", x, np.full_like(x, 2.0)).astype(np.float64) - ref) / ref) < 4e-6
    # exact ops
    a = rng.normal(size=100000).astype(np.float32) * 100
    b = np.exp(rng.uniform(-5, 5, 100000)).astype(np.float32)
    assert np.array_equal(bits(orc.math_eval("div", a, b)), bits(a / b))
    assert np.array_equal(bits(orc.math_eval("sqrt", b)), bits(np.sqrt(b)))
    # edge cases
    e = orc.math_eval("exp", np.float32([-200, -104, 0, 89, np.nan]))
    assert e[0] == 0 and e[1] == 0 and e[2] == 1 and np.isinf(e[3]) and np.isnan(e[4])
    assert orc.math_eval("pow", np.float32([0.0]), np.float32([2.0]))[0] == 0.0


def test_oracle_matches_golden_images(cornell_oracle_scene, golden_256):
    """The oracle reproduces the committed fixtures bit-for-bit (guards against
    compiler / host drift: same answer with and without hardware FMA)."""
    sc = cornell_oracle_scene
    for spp in (1, 2, 16, 17):
        acc, rgba, cnt = sc.render(spp)
        assert np.array_equal(rgba, golden_256[f"rgba_{spp}"])
        assert np.array_equal(bits(acc[112:144, 112:144]), bits(golden_256[f"accum_crop_{spp}"]))
        assert np.array_equal(cnt, golden_256[f"counters_{spp}"])


def test_oracle_incremental_equals_fused(cornell_oracle_scene):
    """n dispatches == one fused call (the accumulator is summed in sample order)."""
    sc = cornell_oracle_scene
    rect = (96, 96, 160, 160)
    a17, r17, _ = sc.render(17, rect=rect)
    acc = None
    for s in range(1, 18):
        acc, rgba, _ = sc.render(1, first_sample=s, rect=rect, accum=acc)
    assert np.array_equal(bits(acc), bits(a17)) and np.array_equal(rgba, r17)


def test_probe_transcripts(cornell_oracle_scene):
    sc = cornell_oracle_scene
    with open(os.path.join(GOLDEN, "probes.json")) as f:
        g = json.load(f)
    assert int(np.float32(sc.hit_pad()).view(np.uint32)) == g["hit_pad_bits"]
    assert [int(v) for v in sc.camera_frame().view(np.uint32)] == g["camera_frame_bits"]
    seen_light = seen_glass = seen_miss = False
    for p in g["probes"]:
        t = sc.trace_pixel(p["x"], p["y"], p["sample"])
        hits = [int(h) for h in t.hits[:t.n_hits]]
        assert hits == p["hits"] and int(t.n_rand) == p["n_rand"]
        assert [int(v) for v in np.asarray(t.xyz[:], np.float32).view(np.uint32)] == p["xyz_bits"]
        seen_light |= hits[:1] == [2]
        seen_glass |= 17 in hits
        seen_miss |= hits[-1] == 0xFFFFFFFF
    assert seen_light and seen_glass           # light visible through the tie rule (Q4); glass paths covered


def test_tie_rule_light_beats_ceiling(cornell_oracle_scene):
    """Q4: ceiling (1) and light (2) are coplanar at y=555; equal t -> the later primitive wins."""
    of, ou = cornell_oracle_scene.intersect([278, 273, 279.5], [0, 1, 0])
    assert ou[0] == 1 and ou[1] == 2 and of[0] == np.float32(282.0)
    of, ou = cornell_oracle_scene.intersect([50, 273, 279.5], [0, 1, 0])
    assert ou[1] == 1


def test_exclude_and_tmin(cornell_oracle_scene):
    """Q5: self-hit avoidance by `exclude == index` and t_min = 0.001."""
    sc = cornell_oracle_scene
    of, ou = sc.intersect([278, 0, 279.5], [0, 1, 0], exclude=2)
    assert ou[1] == 1                          # light excluded -> ceiling
    of, ou = sc.intersect([278, 554.9995, 279.5], [0, 1, 0])
    assert ou[0] == 0                          # t = 0.0005 < t_min on both -> miss
