import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, vrt_import, bench
V = vrt_import.vrt()
for name, W, H, mode in (("dragon", 1920, 1080, 0), ("dragon", 1920, 1080, 1), ("dragon", 1920, 1080, 2), ("monu9", 1280, 720, 0), ("nature", 3840, 2160, 1)):
    w = bench.load_world(V, name)
    tex, dim = w.flatten()
    p = bench.POSES[name]
    ip, iv, cp, _ = V.camera_block(p[:3], p[3], p[4], W, H)
    ctx = V.Context(0)
    ctx.upload_octree(tex, dim)
    ctx.set_camera(ip, iv, cp)
    d_rgba = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    d_id = torch.zeros((H, W, 2), dtype=torch.int32, device="cuda")
    res = {}
    for label, wmin, wmax in (("reference world", (-1023,)*3, (1024,)*3), ("octant only", (0,)*3, (1024,)*3)):
        prm = ctx.default_params()
        prm.world_min[:] = wmin; prm.world_max[:] = wmax
        ctx.set_params(prm)
        for _ in range(300): ctx.dispatch_rows(W, H, 0, H, mode, d_rgba.data_ptr(), d_id.data_ptr())
        torch.cuda.synchronize()
        ms = float(np.median(ctx.dispatch_timed(W, H, 0, H, mode, d_rgba.data_ptr(), d_id.data_ptr(), 200)))
        torch.cuda.synchronize()
        res[label] = (ms, d_rgba.cpu().numpy().copy(), d_id.cpu().numpy().copy())
        print(name, W, H, "mode", mode, label, "%.4f ms" % ms, flush=True)
    a, b = res["reference world"], res["octant only"]
    same_rgba = np.array_equal(a[1], b[1]); same_id = np.array_equal(a[2][..., 0], b[2][..., 0])
    dist_diff = np.count_nonzero(a[2][..., 1] != b[2][..., 1])
    print("   same rgba", same_rgba, "same id", same_id, "dist words differing", dist_diff, "(sky pixels carry the world size)")
    ctx.close()
