#!/usr/bin/env python3
"""Traversal study on the CPU oracle's find trace: how many tree levels a find costs under different
restart schemes, per ray and per 8x8 wave tile (max over lanes per DDA step = what a converged wave pays)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402

REC = np.dtype([("pixel", "u4"), ("x", "i2"), ("y", "i2"), ("z", "i2"), ("fd", "u1"), ("sd", "u1"), ("leaf", "u1"),
                ("pad", "u1"), ("pad2", "u2")])


def trace(map_name, pose, W, H, mode):
    L = O.lib()
    t, ok, n = O.load_vox(os.path.join(ROOT, "tests/golden/maps", map_name + ".vox"))
    tex, dim = O.flatten(t)
    (ip, iv, cp), _ = O.camera_ubo(pose[:3], pose[3], pose[4], W, H)
    s = O.make_scene(tex, dim, ip, iv, cp)
    buf = np.zeros(8_000_000, REC)
    L.o_set_find_trace.argtypes = [C.c_void_p, C.c_size_t]
    L.o_find_trace_count.restype = C.c_size_t
    L.o_set_find_trace(buf.ctypes.data, buf.size)
    O.render(s, W, H, mode)
    n = L.o_find_trace_count()
    L.o_set_find_trace(None, 0)
    return buf[:n].copy()


def main():
    W, H = 480, 272
    r = trace("dragon", (63.5, 60.5, 140.5, -90.0, -10.0), W, H, 0)
    fd, sd = r["fd"].astype(int), r["sd"].astype(int)
    print("finds", len(r), "found depth mean", fd.mean(), "reference scheme levels/find", (fd - sd).mean() + 0,
          "root restarts", (sd == 0).mean())
    # consecutive finds of the same pixel: common-ancestor depth from the xor of query points (inside [0,1024)^3)
    same = np.zeros(len(r), bool)
    same[1:] = r["pixel"][1:] == r["pixel"][:-1]
    d = np.zeros(len(r), np.int64)
    for ax in ("x", "y", "z"):
        v = r[ax].astype(np.int64)
        d[1:] |= (v[1:] ^ v[:-1])
    inside = (r["x"] >= 0) & (r["y"] >= 0) & (r["z"] >= 0)
    msb = np.where(d > 0, np.floor(np.log2(np.maximum(d, 1))).astype(int) + 1, 0)  # side 2^msb cube contains both
    ca_depth = 1 + (10 - np.minimum(msb, 10))  # depth of the smallest aligned cube holding both points (root=0, [0,1024)=1)
    prev_fd = np.zeros(len(r), int)
    prev_fd[1:] = fd[:-1]
    par_depth_prev = prev_fd - 1
    for name, start in [
        ("parent or root (reference)", np.where(same & (ca_depth >= par_depth_prev), par_depth_prev, 0)),
        ("parent / anchor(32) / root", np.where(same & (ca_depth >= par_depth_prev), par_depth_prev,
                                                np.where(same & (ca_depth >= 6), np.minimum(6, par_depth_prev), 0))),
        ("parent / anchor(8) / anchor(64) / root", np.where(same & (ca_depth >= par_depth_prev), par_depth_prev,
                                                            np.where(same & (ca_depth >= 8), np.minimum(8, par_depth_prev),
                                                                     np.where(same & (ca_depth >= 5), np.minimum(5, par_depth_prev), 0)))),
        ("full ancestor stack", np.where(same, np.minimum(ca_depth, par_depth_prev), 0)),
    ]:
        start = np.where(inside, start, 0)
        lv = np.maximum(fd - start, 1)
        # per wave tile (8x8 pixels) and per step index: max over lanes
        px, py = r["pixel"] % W, r["pixel"] // W
        tile = (py // 8) * (W // 8) + (px // 8)
        step = np.zeros(len(r), int)
        first = ~same
        idx = np.arange(len(r))
        start_idx = np.maximum.accumulate(np.where(first, idx, 0))
        step = idx - start_idx
        key = tile * 2048 + step
        order = np.argsort(key, kind="stable")
        k, lvs = key[order], lv[order]
        bounds = np.flatnonzero(np.diff(k)) + 1
        wave_max = np.maximum.reduceat(lvs, np.concatenate([[0], bounds]))
        wave_sum_lanes = np.add.reduceat(lvs, np.concatenate([[0], bounds]))
        print(f"{name:42s} levels/find {lv.mean():5.2f}   wave-level: sum of per-step max {wave_max.sum() / (W * H / 64):7.1f} per wave,"
              f" lane utilisation {wave_sum_lanes.sum() / (wave_max.sum() * 64):.2f}, steps per wave {len(wave_max) / (W * H / 64):.1f}")


if __name__ == "__main__":
    main()
