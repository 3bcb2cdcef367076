"""Deterministic synthetic scenes for the benchmark configs.

The reference ships one scene (cornell) and no triangle meshes, so the larger
configs of BASELINE.json are procedural.  Every scene keeps cornell's camera,
spectra, CIE table, light patch and the five wall patches (patches 0..5), so
shading / NEE / MIS run exactly as in the reference; the boxes and spheres are
replaced by category-2 triangles (v0, e1 = v1-v0, e2 = v2-v0).

    S1 mesh10k     torus 72 x 72 quads                         10 368 tris
    S2 atrium250k  256 x 256 heightfield + 64 columns         253 952 tris
    S3 soup        N random small triangles (default 10 M), seed 0xC0FFEE10

Generators use only +,-,*,/ and sqrt on float64 plus literal constants (no
libm), then one cast to float32, so a twin in another language can be made
bit-identical; tests pin the SHA-256 of the packed buffers.
"""
from __future__ import annotations

import numpy as np

from . import scene as _scene
from .scene import PackedScene, TYPE_INDEX, make_primitives

SEED_ATRIUM = 0x5EED0250
SEED_SOUP = 0xC0FFEE10

# cos/sin of 2*pi/72 and 2*pi/32 as literals (see module docstring)
_C72, _S72 = 0.9961946980917455, 0.08715574274765817
_C32, _S32 = 0.9807852804032304, 0.19509032201612825


# --------------------------------------------------------------------------- hashing
def pcg_hash(v: np.ndarray) -> np.ndarray:
    """PCG-RXS-M-XS 32-bit output function used as a counter hash (uint32 in/out)."""
    v = np.asarray(v, dtype=np.uint64) & 0xFFFFFFFF
    state = (v * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return (((word >> 22) ^ word) & 0xFFFFFFFF).astype(np.uint32)


def counter_u01(seed: int, ctr: np.ndarray) -> np.ndarray:
    """Counter-based uniform in [0,1): 24 high bits of pcg_hash(ctr ^ pcg_hash(seed))."""
    key = int(pcg_hash(np.asarray([seed]))[0])
    h = pcg_hash((np.asarray(ctr, dtype=np.uint64) & 0xFFFFFFFF) ^ key)
    return (h >> 8).astype(np.float64) / 16777216.0


def _circle(n: int, c: float, s: float) -> np.ndarray:
    """n points on the unit circle by complex-rotation recurrence (float64)."""
    out = np.empty((n, 2))
    x, y = 1.0, 0.0
    for k in range(n):
        out[k] = (x, y)
        x, y = x * c - y * s, x * s + y * c
    return out


# --------------------------------------------------------------------------- assembly
def _base(width: int, height: int):
    """cornell's spectra / camera / walls+light (patches 0..5)."""
    sc = _scene.load_scene()
    base = {"camera": dict(sc["camera"], width=width, height=height),
            "objects": {"patches": sc["objects"]["patches"][:6], "spheres": []},
            "spectra": sc["spectra"]}
    return _scene.pack_scene(base)


def _with_triangles(base: PackedScene, v0, v1, v2, reflectance_idx) -> PackedScene:
    v0 = np.asarray(v0, np.float64).astype(np.float32)
    v1 = np.asarray(v1, np.float64).astype(np.float32)
    v2 = np.asarray(v2, np.float64).astype(np.float32)
    n = len(v0)
    first = len(base.primitives)
    dark = base.spectrum_index["dark"]
    tris = make_primitives(np.full(n, 2, np.uint32), v0, v1 - v0, v2 - v0,
                           np.full(n, dark, np.uint32), np.asarray(reflectance_idx, np.uint32),
                           np.full(n, TYPE_INDEX["diffuse"], np.uint32), first_index=first)
    prims = np.zeros(first + n, _scene.PRIM_DTYPE)
    prims[:first] = base.primitives
    prims[first:] = tris
    return PackedScene(primitives=prims, lights=_scene.lights_of(prims), camera=base.camera,
                       spectra=base.spectra, cie=base.cie, patches=base.patches,
                       spectrum_index=base.spectrum_index)


def _grid_tris(P: np.ndarray, wrap_u: bool, wrap_v: bool):
    """P[nu, nv, 3] vertex grid -> two triangles per quad (v0,v1,v2 arrays)."""
    nu, nv = P.shape[:2]
    iu = np.arange(nu if wrap_u else nu - 1)
    iv = np.arange(nv if wrap_v else nv - 1)
    a, b = np.meshgrid(iu, iv, indexing="ij")
    a1, b1 = (a + 1) % nu, (b + 1) % nv
    p00, p10, p01, p11 = P[a, b], P[a1, b], P[a, b1], P[a1, b1]
    v0 = np.concatenate([p00.reshape(-1, 3), p10.reshape(-1, 3)])
    v1 = np.concatenate([p10.reshape(-1, 3), p11.reshape(-1, 3)])
    v2 = np.concatenate([p01.reshape(-1, 3), p01.reshape(-1, 3)])
    return v0, v1, v2


# --------------------------------------------------------------------------- S1
def mesh10k(width: int = 1920, height: int = 1080) -> PackedScene:
    """S1: torus, 72 x 72 quads = 10 368 triangles, centre (278,200,278), R=120, r=50."""
    base = _base(width, height)
    ring = _circle(72, _C72, _S72)
    R, r = 120.0, 50.0
    cphi, sphi = ring[:, 0][:, None], ring[:, 1][:, None]
    cth, sth = ring[:, 0][None, :], ring[:, 1][None, :]
    rad = R + r * cth
    P = np.stack([278.0 + rad * cphi, 200.0 + r * sth + 0.0 * cphi, 278.0 + rad * sphi], axis=-1)
    v0, v1, v2 = _grid_tris(P, True, True)
    return _with_triangles(base, v0, v1, v2, np.full(len(v0), base.spectrum_index["white"]))


# --------------------------------------------------------------------------- S2
def _value_noise(x: np.ndarray, z: np.ndarray, cells: int, seed: int) -> np.ndarray:
    """Bilinear value noise with smoothstep on a cells x cells lattice over [0,1]^2."""
    fx, fz = x * cells, z * cells
    ix = np.minimum(fx.astype(np.int64), cells - 1)
    iz = np.minimum(fz.astype(np.int64), cells - 1)
    tx, tz = fx - ix, fz - iz
    sx, sz = tx * tx * (3.0 - 2.0 * tx), tz * tz * (3.0 - 2.0 * tz)

    def lat(i, j):
        return counter_u01(seed, (i + (cells + 1) * j).astype(np.uint64))

    v00, v10, v01, v11 = lat(ix, iz), lat(ix + 1, iz), lat(ix, iz + 1), lat(ix + 1, iz + 1)
    return (v00 * (1 - sx) + v10 * sx) * (1 - sz) + (v01 * (1 - sx) + v11 * sx) * sz


def atrium250k(width: int = 1920, height: int = 1080) -> PackedScene:
    """S2 (Sponza-scale stand-in): 131 072-triangle heightfield floor + 64 columns of 1 920
    triangles each = 253 952 triangles; reflectance cycles white/red/green by object."""
    base = _base(width, height)
    idx = base.spectrum_index
    g = np.arange(257) / 256.0
    gx, gz = np.meshgrid(g, g, indexing="ij")
    h = 0.65 * _value_noise(gx, gz, 16, SEED_ATRIUM) + 0.35 * _value_noise(gx, gz, 64, SEED_ATRIUM + 1)
    P = np.stack([1.0 + 553.0 * gx, 0.5 + 8.0 * h, 1.0 + 553.0 * gz], axis=-1)
    v0, v1, v2 = _grid_tris(P, False, False)
    refl = [np.full(len(v0), idx["white"])]
    V0, V1, V2 = [v0], [v1], [v2]
    ring = _circle(32, _C32, _S32)
    cycle = [idx["white"], idx["red"], idx["green"]]
    yy = np.arange(31) / 30.0
    for k in range(64):
        ci, cj = k % 8, k // 8
        cx, cz = 555.0 * (ci + 0.5) / 8.0, 555.0 * (cj + 0.5) / 8.0
        u = counter_u01(SEED_ATRIUM + 2, np.asarray([3 * k, 3 * k + 1, 3 * k + 2]))
        r0 = 9.0 + 5.0 * u[0]
        top = 380.0 + 160.0 * u[1]
        bulge = 0.15 + 0.25 * u[2]
        rad = r0 * (1.0 + bulge * (4.0 * yy * (1.0 - yy)) - 0.3 * yy)          # entasis profile
        Pc = np.stack([cx + rad[None, :] * ring[:, 0][:, None],
                       0.0 * ring[:, 0][:, None] + (9.0 + (top - 9.0) * yy)[None, :],
                       cz + rad[None, :] * ring[:, 1][:, None]], axis=-1)     # [32 seg, 31 rings]
        a, b, c = _grid_tris(Pc, True, False)
        V0.append(a); V1.append(b); V2.append(c)
        refl.append(np.full(len(a), cycle[(k + 1) % 3]))
    return _with_triangles(base, np.concatenate(V0), np.concatenate(V1), np.concatenate(V2),
                           np.concatenate(refl))


# --------------------------------------------------------------------------- S3
def soup(n: int = 10_000_000, width: int = 1920, height: int = 1080, seed: int = SEED_SOUP) -> PackedScene:
    """S3: n random triangles, v0 ~ U[0,555]^3, e1,e2 ~ U[-2,2]^3, reflectance uniform over
    {white, green, red}; triangle k draws counters 10k .. 10k+9."""
    base = _base(width, height)
    idx = base.spectrum_index
    k = np.arange(n, dtype=np.uint64) * 10
    u = [counter_u01(seed, k + j) for j in range(10)]
    v0 = np.stack([555.0 * u[0], 555.0 * u[1], 555.0 * u[2]], -1)
    e1 = np.stack([4.0 * u[3] - 2.0, 4.0 * u[4] - 2.0, 4.0 * u[5] - 2.0], -1)
    e2 = np.stack([4.0 * u[6] - 2.0, 4.0 * u[7] - 2.0, 4.0 * u[8] - 2.0], -1)
    pick = np.minimum((3.0 * u[9]).astype(np.int64), 2)
    refl = np.asarray([idx["white"], idx["green"], idx["red"]], np.uint32)[pick]
    return _with_triangles(base, v0, v0 + e1, v0 + e2, refl)


SCENES = {"cornell": lambda w=256, h=256: _scene.cornell(w, h), "mesh10k": mesh10k,
          "atrium250k": atrium250k, "soup": soup}
