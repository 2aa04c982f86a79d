#!/usr/bin/env python3
"""(needs a `make AB=1` build) A/B of the full path tracer: one kernel (round 1) against two (trace_kernel<3> + bounce_kernel); per-launch time by the
events attached to the dispatch (vrt_set_profiling: first kernel's start to last kernel's end) and the frame rate of
back-to-back launches. usage: tools/full_split_ab.py [map ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import vrt_import  # noqa: E402

V = vrt_import.vrt()
SIZES = {"dragon": (1920, 1080), "monu9": (1280, 720), "nature": (3840, 2160), "terrain": (1920, 1080)}


def main():
    for name in sys.argv[1:] or ["dragon"]:
        W, H = SIZES[name]
        tex, dim = bench.load_world(V, name).flatten()
        p = bench.POSES[name]
        ip, iv, cp, _ = V.camera_block(p[:3], p[3], p[4], W, H)
        ctx = V.Context(0)
        ctx.upload_octree(tex, dim)
        ctx.set_camera(ip, iv, cp)
        d_rgba = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        d_id = torch.zeros((H, W, 2), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ref = None
        for split in (False, True):
            ctx.set_full_split(split)
            for sched in (0, 16):
                ctx.set_tile_scheduling(sched)
                for _ in range(20):
                    ctx.dispatch_rows(W, H, 0, H, 2, d_rgba.data_ptr(), d_id.data_ptr())
                ctx.synchronize()
                n = 100
                ctx.set_profiling(n, every=5)
                t0 = time.perf_counter()
                for _ in range(n):
                    ctx.dispatch_rows(W, H, 0, H, 2, d_rgba.data_ptr(), d_id.data_ptr())
                ctx.synchronize()
                dt = (time.perf_counter() - t0) / n
                ms = ctx.profile_read(n)
                ctx.set_profiling(0)
                h = (V.fnv1a64(d_rgba.cpu().numpy()), V.fnv1a64(d_id.cpu().numpy()))
                ref = ref or h
                print(f"{name} {W}x{H} full  {'two kernels' if split else 'one kernel '}  sched {sched:2d}: {dt * 1e6:8.2f} us/frame, "
                      f"launch(es) by events {float(np.mean(ms)) * 1e3:8.2f} us  same={h == ref}", flush=True)
        if os.environ.get("VRT_BOUNCE_SWEEP"):
            ctx.set_full_split(True)
            ctx.set_tile_scheduling(16)
            for wps in (1, 2, 4, 6):
                for below in (1, 16, 32, 40, 48, 64):
                    ctx.set_bounce(below, wps)
                    for _ in range(10):
                        ctx.dispatch_rows(W, H, 0, H, 2, d_rgba.data_ptr(), d_id.data_ptr())
                    ctx.synchronize()
                    n = 40
                    t0 = time.perf_counter()
                    for _ in range(n):
                        ctx.dispatch_rows(W, H, 0, H, 2, d_rgba.data_ptr(), d_id.data_ptr())
                    ctx.synchronize()
                    dt = (time.perf_counter() - t0) / n
                    h = (V.fnv1a64(d_rgba.cpu().numpy()), V.fnv1a64(d_id.cpu().numpy()))
                    print(f"{name} two kernels: {wps} waves/SIMD, refill below {below:2d}: {dt * 1e6:8.2f} us/frame same={h == ref}", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
