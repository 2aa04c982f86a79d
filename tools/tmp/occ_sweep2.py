import importlib, os, sys, time, numpy as np
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
import vrt_import
V = vrt_import.vrt()
import torch
shd = importlib.import_module("voxel-raytracer_amd.sharding")
W, H, steps = 1920, 1080, 300
w = V.World(); assert w.load_vox(os.path.join(ROOT, "tests/golden/maps/dragon.vox"))
tex, dim = w.flatten()
ctx = V.Context(0); ctx.upload_octree(tex, dim)
ip, iv, cp, _ = V.camera_block((63.5, 60.5, 140.5), -90.0, -10.0, W, H); ctx.set_camera(ip, iv, cp)
dev = torch.device("cuda", 0); st = torch.cuda.Stream(dev)
plan = shd.ShardPlan(W, H, 8, 0, 1); buf = plan.local_buffer(dev); p = plan.pointers(buf)
ref = {}
for rep in range(2):
    for mode in (0, 1, 2):
        for variant, name in [(0, "default (64 thr)"), (14, "256 thr, 6 waves"), (19, "64 thr, 7 waves")]:
            for sched in (0, 16):
                ctx.set_variant(variant); ctx.set_tile_scheduling(sched)
                buf.zero_()
                n = steps if mode < 2 else 100
                for _ in range(40):
                    ctx.dispatch_shard(W, H, 8, 0, 1, mode, p[0], p[1], st.cuda_stream)
                torch.cuda.synchronize()
                ctx.set_profiling(n, every=7)
                t0 = time.perf_counter()
                for _ in range(n):
                    ctx.dispatch_shard(W, H, 8, 0, 1, mode, p[0], p[1], st.cuda_stream)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n * 1e6
                k = ctx.profile_read(n); ctx.set_profiling(0)
                if mode not in ref: ref[mode] = buf.clone()
                print("mode %d %-18s sched %2d: %7.2f us/frame, kernel avg %7.2f  same=%s" % (mode, name, sched, dt, k.mean() * 1e3, bool(torch.equal(buf, ref[mode]))), flush=True)
