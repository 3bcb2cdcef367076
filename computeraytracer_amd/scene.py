"""Host-side scene flatten + pack + upload buffers.

Python mirror of the reference's inline packer, src/main.js:114-393 (the code
the north-star calls "sceneLoader scene upload"; src/sceneLoader.js itself is a
5-line stub).  Produces byte-identical host buffers for bind-group entries
b3..b8 of ComputeShader.wgsl:1-9:

    primitives  n x 80 B   main.js:211-246   (category@0 data1@16 data2@32 data3@48 data4@64)
    patches     n x 64 B   main.js:138-209   (bound by the reference, dead in the shader)
    lights      n x 80 B   main.js:255-296
    camera      16 f32     main.js:313-324
    spectra     n x 301    main.js:334-378   (f64 lerp, then f32)
    cie         3 x 471    main.js:380-393

Extensions (not in the reference): ``objects.triangles`` [{v0,v1,v2,...}] and
``objects.meshes`` [{obj: "file.obj" | vertices+indices, scale, translate,
emission, reflectance, type}] -> category 2 records, v0 = data1, e1 = v1-v0 =
data2, e2 = v2-v0 = data3 (f32 subtraction of f32 vertices), appended after the
spheres with index = array position (same rule as main.js:124,133).
The JS twin of this file is host/sceneLoader.js; tests check both against the
SHA-256 pins of SURVEY.md 8c and against each other.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass, field

import numpy as np

LAMBDA_MIN = 400  # main.js:334
LAMBDA_MAX = 700  # main.js:335
NLAMBDA = LAMBDA_MAX - LAMBDA_MIN + 1
NCIE = 471

TYPE_INDEX = {"diffuse": 0, "light": 1, "glass": 2}  # main.js:166-170
CATEGORY = {"patch": 0, "sphere": 1, "triangle": 2}

PRIM_DTYPE = np.dtype({
    "names": ["category", "data1", "data2", "data3", "data4"],
    "formats": ["<u4", ("<f4", 3), ("<f4", 3), ("<f4", 3), ("<u4", 4)],
    "offsets": [0, 16, 32, 48, 64],
    "itemsize": 80,
})
PATCH_DTYPE = np.dtype({
    "names": ["origin", "edge1", "edge2", "emission", "reflectance", "type", "index"],
    "formats": [("<f4", 3), ("<f4", 3), ("<f4", 3), "<u4", "<u4", "<u4", "<u4"],
    "offsets": [0, 16, 32, 44, 48, 52, 56],
    "itemsize": 64,
})

_HERE = os.path.dirname(os.path.abspath(__file__))
SCENES_DIR = os.path.normpath(os.path.join(_HERE, "..", "scenes"))


@dataclass
class PackedScene:
    """The host buffers the compute pass binds (b3..b8) plus image size."""
    primitives: np.ndarray            # PRIM_DTYPE [nprim]
    lights: np.ndarray                # PRIM_DTYPE [nlight]
    camera: np.ndarray                # float32 [16]
    spectra: np.ndarray               # float32 [nspectra, 301]
    cie: np.ndarray                   # float32 [3, 471]
    patches: np.ndarray = field(default_factory=lambda: np.zeros(0, PATCH_DTYPE))
    spectrum_index: dict = field(default_factory=dict)

    @property
    def width(self) -> int:
        return int(self.camera[11])

    @property
    def height(self) -> int:
        return int(self.camera[12])

    def with_size(self, width: int, height: int) -> "PackedScene":
        cam = self.camera.copy()
        cam[11] = width
        cam[12] = height
        return PackedScene(self.primitives, self.lights, cam, self.spectra, self.cie,
                           self.patches, self.spectrum_index)


# --------------------------------------------------------------------------- spectra
def _lerp(a, b, t):
    return a * (1 - t) + b * t  # main.js:626


def sample_spectrum(spectrum: dict, lam: float) -> float:
    """main.js:340-356 (JS doubles)."""
    wl = spectrum["wavelength"]
    val = spectrum["value"]
    index = -1
    for i, e in enumerate(wl):          # findIndex(e >= lambda)
        if e >= lam:
            index = i
            break
    start_index = max(index - 1, 0)
    end_index = min(index, len(wl) - 1)
    if end_index < 0:                   # JS: value[-1] is undefined -> NaN result
        return math.nan
    start, end = val[start_index], val[end_index]
    start_lambda, end_lambda = wl[start_index], wl[end_index]
    if start_lambda == end_lambda:
        return start
    return _lerp(start, end, (lam - start_lambda) / (end_lambda - start_lambda))


def resample_spectra(spectra: dict) -> tuple[np.ndarray, dict]:
    """main.js:157-164 (insertion-order indices) + main.js:358-367."""
    key_index = {}
    rows = []
    for index, (key, sp) in enumerate(spectra.items()):
        key_index[key] = index
        rows.append([sample_spectrum(sp, LAMBDA_MIN + i) for i in range(NLAMBDA)])
    table = np.asarray(rows, dtype=np.float64).astype(np.float32).reshape(len(rows), NLAMBDA)
    return table, key_index


def load_cie(path: str | None = None) -> np.ndarray:
    """main.js:380-382: Float32Array([...CIE_X, ...CIE_Y, ...CIE_Z])."""
    path = path or os.path.join(SCENES_DIR, "cie1931_xyz_1nm.json")
    with open(path) as f:
        d = json.load(f)
    x, y, z = (d.get("X", d.get("CIE_X")), d.get("Y", d.get("CIE_Y")), d.get("Z", d.get("CIE_Z")))
    cie = np.asarray([x, y, z], dtype=np.float64).astype(np.float32)
    if cie.shape != (3, NCIE):
        raise ValueError(f"CIE table must be 3x{NCIE}, got {cie.shape}")
    return cie


# --------------------------------------------------------------------------- packing
def pack_camera(cam: dict) -> np.ndarray:
    """main.js:313-324."""
    return np.asarray([*cam["eye"], 0, *cam["lookat"], 0, *cam["up"],
                       cam["width"], cam["height"], cam["focalLength"], 0, 0], dtype=np.float32)


def make_primitives(category, data1, data2, data3, emission, reflectance, material,
                    first_index: int = 0) -> np.ndarray:
    """Vectorised 80-byte record builder (index = array position)."""
    n = len(category)
    rec = np.zeros(n, PRIM_DTYPE)
    rec["category"] = category
    rec["data1"] = np.asarray(data1, np.float32).reshape(n, 3)
    rec["data2"] = np.asarray(data2, np.float32).reshape(n, 3)
    rec["data3"] = np.asarray(data3, np.float32).reshape(n, 3)
    rec["data4"][:, 0] = emission
    rec["data4"][:, 1] = reflectance
    rec["data4"][:, 2] = material
    rec["data4"][:, 3] = np.arange(first_index, first_index + n, dtype=np.uint32)
    return rec


def lights_of(primitives: np.ndarray) -> np.ndarray:
    """main.js:255-296: every primitive whose type is 'light', packed as a patch
    (category word written as 0)."""
    mask = primitives["data4"][:, 2] == TYPE_INDEX["light"]
    sel = np.zeros(int(mask.sum()), PRIM_DTYPE)
    sel[:] = primitives[mask]
    sel["category"] = 0
    return sel


def parse_obj(text: str):
    """Minimal Wavefront OBJ reader: `v x y z` and `f a b c ...` (1-based, negative = relative,
    `a/b/c` forms accepted, polygons fan-triangulated).  Returns (vertices [n][3], triangles [m][3])."""
    verts, tris = [], []
    for line in text.splitlines():
        t = line.split()
        if not t or t[0].startswith("#"):
            continue
        if t[0] == "v" and len(t) >= 4:
            verts.append([float(t[1]), float(t[2]), float(t[3])])
        elif t[0] == "f" and len(t) >= 4:
            idx = []
            for w in t[1:]:
                k = int(w.split("/")[0])
                idx.append(k - 1 if k > 0 else len(verts) + k)
            for j in range(1, len(idx) - 1):
                tris.append([idx[0], idx[j], idx[j + 1]])
    return verts, tris


def expand_meshes(scene: dict, base_dir: str | None = None) -> list:
    """objects.meshes -> list of triangle dicts (vertex = v * scale + translate, in doubles)."""
    out = []
    for m in scene.get("objects", {}).get("meshes", []):
        if "obj" in m:
            path = m["obj"] if os.path.isabs(m["obj"]) else os.path.join(base_dir or SCENES_DIR, m["obj"])
            with open(path) as f:
                verts, tris = parse_obj(f.read())
        else:
            verts = m["vertices"]
            flat = m["indices"]
            tris = [flat[i:i + 3] for i in range(0, len(flat), 3)] if flat and not isinstance(flat[0], (list, tuple)) else flat
        sc = m.get("scale", 1.0)
        sc = [sc, sc, sc] if not isinstance(sc, (list, tuple)) else sc
        tr = m.get("translate", [0.0, 0.0, 0.0])
        P = [[v[0] * sc[0] + tr[0], v[1] * sc[1] + tr[1], v[2] * sc[2] + tr[2]] for v in verts]
        for a, b, c in tris:
            out.append({"v0": P[a], "v1": P[b], "v2": P[c], "emission": m["emission"],
                        "reflectance": m["reflectance"], "type": m["type"]})
    return out


def pack_scene(scene: dict, cie: np.ndarray | None = None, base_dir: str | None = None) -> PackedScene:
    """Flatten (main.js:114-137) and pack (main.js:138-393) a scene dict in the
    reference's JSON schema."""
    objects = scene.get("objects", {})
    patches = objects.get("patches", [])
    spheres = objects.get("spheres", [])
    tris = list(objects.get("triangles", [])) + expand_meshes(scene, base_dir)
    spectra, key_index = resample_spectra(scene["spectra"])

    def names(items, key):
        return [key_index[o[key]] for o in items]

    def types(items):
        return [TYPE_INDEX[o["type"]] for o in items]

    n0, n1, n2 = len(patches), len(spheres), len(tris)
    cat = np.concatenate([np.zeros(n0, np.uint32), np.ones(n1, np.uint32), np.full(n2, 2, np.uint32)])
    d1 = ([p["origin"] for p in patches] + [s["center"] for s in spheres] + [t["v0"] for t in tris])
    d2 = ([p["edge1"] for p in patches] + [[s["radius"]] * 3 for s in spheres]
          + [np.subtract(np.float32(t["v1"]), np.float32(t["v0"])) for t in tris])
    d3 = ([p["edge2"] for p in patches] + [[0, 0, 0] for _ in spheres]
          + [np.subtract(np.float32(t["v2"]), np.float32(t["v0"])) for t in tris])
    allp = list(patches) + list(spheres) + list(tris)
    n = n0 + n1 + n2
    prims = make_primitives(cat,
                            np.asarray(d1, np.float64).reshape(n, 3),
                            np.asarray(d2, np.float64).reshape(n, 3),
                            np.asarray(d3, np.float64).reshape(n, 3),
                            names(allp, "emission"), names(allp, "reflectance"), types(allp))

    pp = np.zeros(n0, PATCH_DTYPE)          # main.js:172-209
    if n0:
        pp["origin"] = np.asarray([p["origin"] for p in patches], np.float32)
        pp["edge1"] = np.asarray([p["edge1"] for p in patches], np.float32)
        pp["edge2"] = np.asarray([p["edge2"] for p in patches], np.float32)
        pp["emission"] = names(patches, "emission")
        pp["reflectance"] = names(patches, "reflectance")
        pp["type"] = types(patches)
        pp["index"] = np.arange(n0, dtype=np.uint32)

    return PackedScene(primitives=prims, lights=lights_of(prims), camera=pack_camera(scene["camera"]),
                       spectra=spectra, cie=cie if cie is not None else load_cie(),
                       patches=pp, spectrum_index=key_index)


def load_scene(path: str | None = None) -> dict:
    path = path or os.path.join(SCENES_DIR, "cornell_box.json")
    with open(path) as f:
        return json.load(f)


def cornell(width: int | None = None, height: int | None = None) -> PackedScene:
    """The reference's default scene (S0 of SURVEY.md 8d)."""
    ps = pack_scene(load_scene())
    if width is not None:
        ps = ps.with_size(width, height if height is not None else width)
    return ps
