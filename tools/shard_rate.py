"""One rank's share of an N-rank run on one GPU: launches shard 0 of N back to back the way bench.py does and reports
the time per step next to the host's enqueue time per step (what bounds the step once the shard kernel is short)."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vrt_import  # noqa: E402

V = vrt_import.vrt()
import torch  # noqa: E402

shd = importlib.import_module("voxel-raytracer_amd.sharding")


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    W, H, steps = 1920, 1080, 2000
    w = V.World()
    assert w.load_vox(os.path.join(root, "tests/golden/maps/dragon.vox"))
    tex, dim = w.flatten()
    ctx = V.Context(0)
    ctx.upload_octree(tex, dim)
    if os.environ.get("SCHED_PERIOD"):   # feedback tile scheduling: 0 = off (default: the library's 16)
        ctx.set_tile_scheduling(int(os.environ["SCHED_PERIOD"]))
    ip, iv, cp, _ = V.camera_block((63.5, 60.5, 140.5), -90.0, -10.0, W, H)
    ctx.set_camera(ip, iv, cp)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for world in (1, 2, 4, 8):
        plan = shd.ShardPlan(W, H, 8, 0, world)
        bufs = [plan.local_buffer(dev) for _ in range(4)]
        ptrs = [plan.pointers(b) for b in bufs]
        side = [torch.cuda.Stream(dev) for _ in range(4)]
        cam = (ip, iv, cp)
        bufs4 = [plan.local_buffer(dev) for _ in range(8)]
        ptrs4 = [plan.pointers(b) for b in bufs4]
        for mode in ("one stream", "two streams", "three streams", "four streams", "2 views/launch", "4 views/launch"):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if mode.endswith("launch"):
                f = int(mode[0])
                sets = [V.make_views([cam + ptrs4[(g * f + j) % 8] for j in range(f)]) for g in range(8 // f)]
                for i in range(steps // f):
                    ctx.dispatch_views(W, H, 8, 0, world, 0, sets[i % len(sets)], stream)
            else:
                ns = {"one": 1, "two": 2, "three": 3, "four": 4}[mode.split()[0]]
                for i in range(steps):
                    p = ptrs[i % max(2, ns)]
                    ctx.dispatch_shard(W, H, 8, 0, world, 0, p[0], p[1],
                                       stream if ns == 1 else side[i % ns].cuda_stream)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print("N=%d shard, %-14s: %7.2f us/step total, host enqueue %7.2f us/step" %
                  (world, mode, (t2 - t0) / steps * 1e6, (t1 - t0) / steps * 1e6))


if __name__ == "__main__":
    main()
