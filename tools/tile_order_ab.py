"""Does the order in which a launch's tiles start matter?  Records the ticks each 8x8 tile took (ORDERED flavour of
the default kernel), then times the same frame with the tiles started heaviest-first (LPT), lightest-first, and in
the natural order, next to the plain kernel. Pixels are compared with the plain kernel's."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vrt_import  # noqa: E402

V = vrt_import.vrt()
import torch  # noqa: E402

shd = importlib.import_module("voxel-raytracer_amd.sharding")


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    W, H, steps = 1920, 1080, 400
    w = V.World()
    assert w.load_vox(os.path.join(root, "tests/golden/maps/dragon.vox"))
    tex, dim = w.flatten()
    ctx = V.Context(0)
    ctx.upload_octree(tex, dim)
    ip, iv, cp, _ = V.camera_block((63.5, 60.5, 140.5), -90.0, -10.0, W, H)
    ctx.set_camera(ip, iv, cp)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    plan = shd.ShardPlan(W, H, 8, 0, 1)
    buf = plan.local_buffer(dev)
    p = plan.pointers(buf)
    L = ctx._L
    L.vrt_set_tile_order.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    n_tiles = (W // 8) * (H // 8)
    n_wg = n_tiles // 4

    def run(label, order, ref=None, sched=0):
        ctx.set_tile_scheduling(sched)
        L.vrt_set_tile_order(ctx._h, 1 if order is not None else 0, order.data_ptr() if order is not None else None, None)
        buf.zero_()
        for _ in range(40):
            ctx.dispatch_shard(W, H, 8, 0, 1, 0, p[0], p[1], st.cuda_stream)
        torch.cuda.synchronize()
        ctx.set_profiling(steps, every=4 if sched != 4 else 3)
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.dispatch_shard(W, H, 8, 0, 1, 0, p[0], p[1], st.cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e6
        k = ctx.profile_read(steps)
        ctx.set_profiling(0)
        same = None if ref is None else bool(torch.equal(buf, ref))
        print("%-34s %7.2f us/frame back to back, kernel avg %7.2f us (min %7.2f)  same pixels: %s" %
              (label, dt, k.mean() * 1e3, k.min() * 1e3, same), flush=True)
        return buf.clone()

    ref = run("plain kernel", None)
    cost = torch.zeros(n_tiles, dtype=torch.int32, device=dev)
    L.vrt_set_tile_order(ctx._h, 1, None, cost.data_ptr())
    ctx.dispatch_shard(W, H, 8, 0, 1, 0, p[0], p[1], st.cuda_stream)
    torch.cuda.synchronize()
    c = cost.cpu().numpy().astype(np.int64) & 0xffffffff
    print("tile ticks: min %d median %d mean %.0f p99 %d max %d; sum/7168 slots = %.0f ticks" %
          (c.min(), np.median(c), c.mean(), np.percentile(c, 99), c.max(), c.sum() / 7168))
    wg = c.reshape(-1, 4).max(1)
    dev_order = lambda o: torch.from_numpy(np.ascontiguousarray(o, dtype=np.int32)).to(dev)
    run("explicit order: identity", dev_order(np.arange(n_wg)), ref)
    run("explicit order: heaviest first", dev_order(np.argsort(-wg, kind="stable")), ref)
    run("explicit order: lightest first", dev_order(np.argsort(wg, kind="stable")), ref)
    run("explicit order: heaviest SUM first", dev_order(np.argsort(-c.reshape(-1, 4).sum(1), kind="stable")), ref)
    run("explicit order: heaviest MIN first", dev_order(np.argsort(-c.reshape(-1, 4).min(1), kind="stable")), ref)
    # heaviest first, but the very heaviest 2 % held back a little so they do not all start on the same CUs
    o = np.argsort(-wg, kind="stable"); k = len(o) // 50
    run("explicit order: top 2 % interleaved", dev_order(np.concatenate([np.stack([o[:k], o[k:2 * k]], 1).reshape(-1), o[2 * k:]])), ref)
    rng = np.random.default_rng(1)
    run("explicit order: random", dev_order(rng.permutation(n_wg)), ref)
    hw = wg >= np.median(wg)
    run("explicit order: heavy half first", dev_order(np.concatenate([np.nonzero(hw)[0], np.nonzero(~hw)[0]])), ref)
    for period in (16, 4, 64, 1):
        run("scheduler, period %d" % period, None, ref, sched=period)
        o = ctx.sched_order(st.cuda_stream)
        assert o.size == n_wg and np.array_equal(np.sort(o), np.arange(n_wg)), "not a permutation"
    run("plain kernel again", None, ref)
    np.save(os.path.join(root, "gpurun_out", "tile_cost.npy"), c)


if __name__ == "__main__":
    main()
