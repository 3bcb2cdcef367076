#!/usr/bin/env python3
"""Import the reference's two DATA files into this repo's scenes/ directory.

Reads  /root/reference/src/scenes/cornell.json  (scene + camera + spectra)
       /root/reference/src/scenes/CIE.json      (CIE 1931 2-degree CMFs, 360..830 nm @ 1 nm)
Writes scenes/cornell_box.json   -- same schema the reference's host code consumes
                                    (src/main.js:114-137,157-170,313-324,340-356),
                                    compact formatting, unused top-level `lights`
                                    dropped (never read by any code: SURVEY Q14)
       scenes/cie1931_xyz_1nm.json -- {"first_nm":360,"X":[..471],"Y":[..],"Z":[..]}

These are input data (numbers), not code.  Run only in the build container; the
GPU box has no /root/reference and uses the committed outputs.
"""
import json, os, sys

REF = "/root/reference/src/scenes"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scenes")


def compact(o, depth=0):
    """JSON with one primitive / one array per line."""
    if isinstance(o, dict):
        if depth >= 2 and all(not isinstance(v, dict) for v in o.values()) and \
           all(not (isinstance(v, list) and len(v) > 16) for v in o.values()):
            return json.dumps(o, separators=(", ", ": "))
        pad = "  " * (depth + 1)
        items = [f'{pad}{json.dumps(k)}: {compact(v, depth + 1)}' for k, v in o.items()]
        return "{\n" + ",\n".join(items) + "\n" + "  " * depth + "}"
    if isinstance(o, list):
        if all(not isinstance(v, (dict, list)) for v in o):
            return json.dumps(o, separators=(",", ":"))
        pad = "  " * (depth + 1)
        return "[\n" + ",\n".join(pad + compact(v, depth + 1) for v in o) + "\n" + "  " * depth + "]"
    return json.dumps(o)


def main():
    c = json.load(open(os.path.join(REF, "cornell.json")))
    scene = {
        "_provenance": "data imported from Meryx/ComputeRayTracer src/scenes/cornell.json "
                       "by tools/import_reference_data.py (numbers only)",
        "camera": c["camera"],
        "objects": {"patches": c["objects"]["patches"], "spheres": c["objects"]["spheres"]},
        "spectra": c["spectra"],
    }
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "cornell_box.json"), "w") as f:
        f.write(compact(scene) + "\n")
    cie = json.load(open(os.path.join(REF, "CIE.json")))
    assert len(cie["CIE_X"]) == len(cie["CIE_Y"]) == len(cie["CIE_Z"]) == 471
    out = {"_provenance": "CIE 1931 2-degree colour matching functions, 360..830 nm @ 1 nm "
                          "(imported from Meryx/ComputeRayTracer src/scenes/CIE.json)",
           "first_nm": 360, "X": cie["CIE_X"], "Y": cie["CIE_Y"], "Z": cie["CIE_Z"]}
    with open(os.path.join(OUT, "cie1931_xyz_1nm.json"), "w") as f:
        f.write(compact(out) + "\n")
    print("wrote", os.path.normpath(OUT))


if __name__ == "__main__":
    sys.exit(main())
