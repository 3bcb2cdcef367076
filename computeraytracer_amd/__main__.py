"""python -m computeraytracer_amd [--scene file.json] [--width W --height H] [--spp N] [--out image.png]"""
import argparse
import json
import os
import time

from . import Renderer, image, scene


def main():
    ap = argparse.ArgumentParser(prog="computeraytracer_amd")
    ap.add_argument("--scene", default=None, help="scene JSON in the reference's schema (default: the cornell box)")
    ap.add_argument("--width", type=int)
    ap.add_argument("--height", type=int)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--accel", default="bvh2", choices=["bvh2", "lbvh", "none"])
    ap.add_argument("--out", default="render.png")
    ap.add_argument("--checkpoint", default=None, help="resume from / save to this .npz")
    args = ap.parse_args()
    sc = scene.load_scene(args.scene)
    if args.width:
        sc["camera"]["width"], sc["camera"]["height"] = args.width, args.height or args.width
    ps = scene.pack_scene(sc, base_dir=os.path.dirname(os.path.abspath(args.scene)) if args.scene else None)
    with Renderer(0) as r:
        r.upload(ps).build_accel(args.accel)
        if args.checkpoint and os.path.exists(args.checkpoint):
            image.load_checkpoint(args.checkpoint, r)
        t0 = time.time()
        r.frame(args.spp).sync()
        dt = time.time() - t0
        rgba = r.read_rgba8()
        (image.write_ppm if args.out.endswith(".ppm") else image.write_png)(args.out, rgba)
        if args.checkpoint:
            image.save_checkpoint(args.checkpoint, r)
        print(json.dumps({"width": ps.width, "height": ps.height, "sample": r.sample, "seconds": round(dt, 4), "out": args.out}))


if __name__ == "__main__":
    main()
