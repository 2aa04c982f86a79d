#!/usr/bin/env python3
"""A/B timing of the kernel variants (vrt_set_variant) on one GPU: median launch time by hipEvents."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vrt_import  # noqa: E402

POSES = {"dragon": (63.5, 60.5, 140.5, -90.0, -10.0), "monu9": (48.5, 60.5, 170.5, -90.0, -12.0),
         "nature": (60.5, 80.5, 200.5, -90.0, -20.0), "terrain": (512.5, 420.5, 1000.5, -90.0, -20.0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map", default="dragon")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--variants", default="0,1,2,3,4,5,6,7,8,9")
    ap.add_argument("--modes", default="0,1")
    args = ap.parse_args()
    V = vrt_import.vrt()
    import bench
    w = bench.load_world(V, args.map)
    tex, dim = w.flatten()
    W, H = args.width, args.height
    p = POSES[args.map]
    ip, iv, cp, _ = V.camera_block(p[:3], p[3], p[4], W, H)
    ctx = V.Context(0)
    ctx.upload_octree(tex, dim)
    ctx.set_camera(ip, iv, cp)
    d_rgba = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    d_id = torch.zeros((H, W, 2), dtype=torch.int32, device="cuda")
    ref = {}
    for mode in [int(m) for m in args.modes.split(",")]:
        for v in [int(x) for x in args.variants.split(",")]:
            ctx.set_variant(v)
            ctx.dispatch_timed(W, H, 0, H, mode, d_rgba.data_ptr(), d_id.data_ptr(), 3)
            ms = ctx.dispatch_timed(W, H, 0, H, mode, d_rgba.data_ptr(), d_id.data_ptr(), args.iters)
            h = V.fnv1a64(d_rgba.cpu().numpy()) ^ V.fnv1a64(d_id.cpu().numpy())
            ref.setdefault(mode, h)
            print(json.dumps({"map": args.map, "mode": mode, "variant": v, "median_ms": round(float(np.median(ms)), 4),
                              "min_ms": round(float(ms.min()), 4), "Mrays/s": round(W * H / float(np.median(ms)) / 1e3, 1),
                              "same_pixels": h == ref[mode], "lds_records": ctx.scene_info()["lds_records"]}), flush=True)


if __name__ == "__main__":
    main()
