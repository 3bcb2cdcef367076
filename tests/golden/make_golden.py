#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the build container).

Sources: scenes/cornell_box.json + scenes/cie1931_xyz_1nm.json (the reference's
two data files, imported by tools/import_reference_data.py), packed by the
host packer, rendered by the CPU oracle (oracle/crt_oracle.c).  The reference
has no tests or golden images of its own (package.json:9), so these vectors
are minted here; the integer/hash pins inside them come from SURVEY.md 8c.

  cornell_buffers.npz   packed b4..b8 host buffers of cornell at 256x256
  cornell_256.npz       oracle output at 256x256: spp 1, 2, 16, 17
                        (accum crop rows/cols 112..144 + sparse probes, full rgba8)
  probes.json           per-path transcripts (hit index sequence, rand count)
"""
import json, os, sys
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from computeraytracer_amd import scene as S      # noqa: E402
from oracle import orc                           # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SPPS = (1, 2, 16, 17)
CROP = (112, 144)
def choose_probes(sc):
    """Deterministic probe pixels: for each kind of first hit (light, ceiling beside the
    light, glass sphere, red sphere, every box face, walls, miss) the first few pixels in
    scan order on a stride-3 grid, plus the four corners."""
    want = {}
    for y in range(0, 256, 3):
        for x in range(0, 256, 3):
            t = sc.trace_pixel(x, y, 1)
            first = int(t.hits[0])
            kind = first
            if first == 1 and 90 < x < 170:
                kind = "ceiling_beside_light"
            lst = want.setdefault(kind, [])
            if len(lst) < (6 if first == 17 else 3):
                lst.append((x, y))
    probes = [p for k in sorted(want, key=str) for p in want[k]]
    probes += [(0, 0), (255, 255), (0, 255), (255, 0)]
    return probes


def main():
    ps = S.cornell(256, 256)
    np.savez_compressed(os.path.join(OUT, "cornell_buffers.npz"),
                        primitives=ps.primitives.view(np.uint8), lights=ps.lights.view(np.uint8),
                        patches=ps.patches.view(np.uint8), camera=ps.camera, spectra=ps.spectra, cie=ps.cie)
    sc = orc.Scene.from_packed(ps)
    PROBES = choose_probes(sc)
    data = {}
    for spp in SPPS:
        acc, rgba, cnt = sc.render(spp)
        data[f"accum_crop_{spp}"] = acc[CROP[0]:CROP[1], CROP[0]:CROP[1]].copy()
        data[f"accum_probe_{spp}"] = np.asarray([acc[y, x] for x, y in PROBES])
        data[f"rgba_{spp}"] = rgba
        data[f"counters_{spp}"] = cnt
        data[f"accum_sum_{spp}"] = acc.astype(np.float64).sum((0, 1))
    np.savez_compressed(os.path.join(OUT, "cornell_256.npz"), **data)
    probes = []
    for (x, y) in PROBES:
        for s in (1, 2, 17):
            t = sc.trace_pixel(x, y, s)
            probes.append(dict(x=x, y=y, sample=s, hits=[int(h) for h in t.hits[:t.n_hits]], n_rand=int(t.n_rand),
                               wavelengths=[int(w) for w in t.wavelengths],
                               radiance_bits=[int(v) for v in np.asarray(t.radiance[:], np.float32).view(np.uint32)],
                               xyz_bits=[int(v) for v in np.asarray(t.xyz[:], np.float32).view(np.uint32)]))
    with open(os.path.join(OUT, "probes.json"), "w") as f:
        json.dump(dict(probe_pixels=PROBES, probes=probes, hit_pad_bits=int(np.float32(sc.hit_pad()).view(np.uint32)),
                       camera_frame_bits=[int(v) for v in sc.camera_frame().view(np.uint32)]), f, indent=0)
    print("wrote fixtures to", OUT)


if __name__ == "__main__":
    main()
