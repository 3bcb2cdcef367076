"""TEST INFRASTRUCTURE (not product): how far may a real WGSL runtime sit from the pinned numeric contract?

The reference holds no fixtures (package.json:9) and its WGSL cannot run here, so the float path is "parity
unpinned" against a real WebGPU stack.  What CAN be measured is the spread between CONFORMANT evaluations of the
same shader text: this script renders the reference's own scene (cornell, 256 x 256) with the oracle built in three
other float configurations (oracle/Makefile `variants`) --

    unfused        dot / cross / ray_at as separate multiplies and adds, left to right (no FMA contraction)
    libm           sin cos tan exp pow from the C library instead of the contract's fixed polynomial kernels
    unfused_libm   both

-- and reports, against the pinned contract: how many pixels keep a bit-identical accumulator at 1 and 16 spp (how
often a path takes the same discrete decisions), and max-rel / RMS-rel difference of the converged XYZ image at N
spp, next to the Monte-Carlo noise of an N-spp image of the contract itself (samples 1..N against N+1..2N).

    python -m oracle.sensitivity [--spp 4096] [--out tests/golden/sensitivity_r02.json]
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
VARIANTS = {"contract": "liborc.so", "unfused": "liborc_v_unfused.so", "libm": "liborc_v_libm.so",
            "unfused_libm": "liborc_v_unfused_libm.so"}

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from computeraytracer_amd import cornell
from oracle import orc
sc = orc.Scene.from_packed(cornell(256, 256))
spp = %(spp)d
out = {}
acc = np.zeros((256, 256, 4), np.float32)
done = 0
for upto in (1, 16, spp, 2 * spp if %(second)d else spp):
    if upto > done:
        acc, _, _ = sc.render(upto - done, first_sample=done + 1, accum=acc)
        done = upto
    out["acc_%%d" %% upto] = acc.copy()
np.savez(%(path)r, **out)
"""


def run_variant(name: str, spp: int, tmp: str) -> dict:
    lib = os.path.join(HERE, VARIANTS[name])
    path = os.path.join(tmp, f"sens_{name}.npz")
    env = dict(os.environ, ORC_LIB=lib)
    subprocess.check_call([sys.executable, "-c", CHILD % dict(root=ROOT, spp=spp, path=path, second=int(name == "contract"))], env=env)
    return dict(np.load(path))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=4096)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "sensitivity_r02.json"))
    ap.add_argument("--tmp", default="/tmp")
    args = ap.parse_args()
    subprocess.check_call(["make", "-s", "-C", HERE, "all", "variants"])
    N = args.spp
    res = {k: run_variant(k, N, args.tmp) for k in VARIANTS}
    ref = res["contract"]

    def xyz(a, n):
        return a[..., :3].astype(np.float64) / n

    def rel(a, b):
        """max and RMS of |a - b| / (|b| + 1 % of the image mean) over pixels and channels."""
        floor = 0.01 * np.abs(b).mean()
        d = np.abs(a - b) / (np.abs(b) + floor)
        return float(d.max()), float(np.sqrt((d ** 2).mean()))

    conv = xyz(ref[f"acc_{N}"], N)
    second = xyz(ref[f"acc_{2 * N}"] - ref[f"acc_{N}"], N)          # samples N+1 .. 2N of the contract
    noise = rel(second, conv)
    table = {"scene": "cornell 256x256 (the reference's own scene)", "spp": N,
             "noise_of_an_N_spp_image": {"max_rel": noise[0], "rms_rel": noise[1],
                                         "what": "the contract's samples N+1..2N against its samples 1..N"},
             "variants": {}}
    for name in ("unfused", "libm", "unfused_libm"):
        v = res[name]
        same1 = float((v["acc_1"].view(np.uint32)[..., :3] == ref["acc_1"].view(np.uint32)[..., :3]).all(-1).mean())
        same16 = float((v["acc_16"].view(np.uint32)[..., :3] == ref["acc_16"].view(np.uint32)[..., :3]).all(-1).mean())
        close1 = float((np.abs(v["acc_1"][..., :3] - ref["acc_1"][..., :3]) <= 1e-4 * np.abs(ref["acc_1"][..., :3]) + 1e-12).all(-1).mean())
        mx, rms = rel(xyz(v[f"acc_{N}"], N), conv)
        table["variants"][name] = {"pixels_bit_identical_at_1spp": same1, "pixels_within_1e-4_rel_at_1spp": close1,
                                   "pixels_bit_identical_at_16spp": same16,
                                   "converged_max_rel": mx, "converged_rms_rel": rms}
    with open(args.out, "w") as f:
        json.dump(table, f, indent=1)
    print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
