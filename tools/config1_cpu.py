#!/usr/bin/env python3
"""BASELINE config 1: custom.vox stand-in, 256x256 frame, single-thread CPU traversal with the host
library's octree_ray_cast (reference src/octree.cpp:405-485), one call per pixel with the same world direction the
kernel would use. Prints one JSON line. (Every pixel against the oracle: tests/test_host.py::test_cpu_ray_cast_matches_oracle_config1.)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import vrt_import  # noqa: E402


def world_dirs(ip, iv, W, H):
    """raytracing.comp:631-638 in numpy float32 with the conventions of the oracle (C1, C3)."""
    f = np.float32
    xs = (np.arange(W, dtype=f) / f(W)) * f(2.0) - f(1.0)
    ys = (np.arange(H, dtype=f) / f(H)) * f(2.0) - f(1.0)
    u, v = np.meshgrid(xs, ys)
    m = ip.reshape(4, 4)  # m[c][r]
    view = [(m[0][r] * u + m[1][r] * v) + (m[2][r] * f(-1.0) + m[3][r] * f(1.0)) for r in range(4)]
    w = view[3]
    div = np.abs(w) > f(1e-6)
    view = [np.where(div, c / w, c).astype(f) for c in view]
    d = ((view[0] * view[0] + view[1] * view[1]) + view[2] * view[2]).astype(f)
    inv = (f(1.0) / np.sqrt(d)).astype(f)
    vd = [(c * inv).astype(f) for c in view[:3]]
    n = iv.reshape(4, 4)
    wd = [((n[0][r] * vd[0] + n[1][r] * vd[1]) + (n[2][r] * vd[2] + n[3][r] * f(0.0))).astype(f) for r in range(3)]
    d2 = ((wd[0] * wd[0] + wd[1] * wd[1]) + wd[2] * wd[2]).astype(f)
    inv2 = (f(1.0) / np.sqrt(d2)).astype(f)
    return np.stack([(c * inv2).astype(f) for c in wd], axis=-1)


def main():
    ap = argparse.ArgumentParser()
    args = ap.parse_args()
    V = vrt_import.vrt()
    data = V.make_custom_vox()
    w = V.World()
    ok, n = w.load_vox_bytes(data)
    assert ok
    W = H = 256
    pos = (32.5, 40.5, 150.5)
    ip, iv, cp, _ = V.camera_block(pos, -90.0, -8.0, W, H)
    dirs = world_dirs(ip, iv, W, H)
    hit = np.zeros((H, W), np.uint8)
    coord = np.zeros((H, W, 3), np.int32)
    t0 = time.perf_counter()
    for y in range(H):
        for x in range(W):
            r = w.ray_cast(pos, dirs[y, x])
            if r is not None:
                hit[y, x] = 1
                coord[y, x] = r[0]
    dt = time.perf_counter() - t0
    out = {"config": "custom.vox stand-in 64^3, 256x256, single-thread CPU octree_ray_cast", "voxels": n,
           "rays": W * H, "hit_fraction": round(float(hit.mean()), 4), "seconds": round(dt, 3),
           "Mrays/s": round(W * H / dt / 1e6, 4), "note": "time includes the Python/ctypes call per ray"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
