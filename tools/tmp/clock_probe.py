import ctypes as C, importlib, os, sys, numpy as np
ROOT = "/root/repo" if os.path.exists("/root/repo/tools") else os.getcwd()
sys.path.insert(0, ROOT)
import vrt_import
V = vrt_import.vrt()
V.HIP_LIB = os.path.join(ROOT, "tools/tmp/libvrt_hip_clock.so")
import torch
shd = importlib.import_module("voxel-raytracer_amd.sharding")
W, H = 1920, 1080
w = V.World(); assert w.load_vox(os.path.join(ROOT, "tests/golden/maps/dragon.vox"))
tex, dim = w.flatten()
ctx = V.Context(0); ctx.upload_octree(tex, dim)
ip, iv, cp, _ = V.camera_block((63.5, 60.5, 140.5), -90.0, -10.0, W, H); ctx.set_camera(ip, iv, cp)
dev = torch.device("cuda", 0); st = torch.cuda.Stream(dev)
plan = shd.ShardPlan(W, H, 8, 0, 1); buf = plan.local_buffer(dev); p = plan.pointers(buf)
L = ctx._L
L.vrt_debug_set_tile_order.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
n = (W // 8) * (H // 8)
cost = torch.zeros(5 * n, dtype=torch.int32, device=dev)
ctx.set_tile_scheduling(0)
L.vrt_debug_set_tile_order(ctx._h, 1, None, cost.data_ptr())
for rep in range(3):
    for _ in range(30):
        ctx.dispatch_shard(W, H, 8, 0, 1, 0, p[0], p[1], st.cuda_stream)
    torch.cuda.synchronize()
    ctx.set_profiling(1, every=1)
    ctx.dispatch_shard(W, H, 8, 0, 1, 0, p[0], p[1], st.cuda_stream)
    torch.cuda.synchronize()
    k = ctx.profile_read(1); ctx.set_profiling(0)
    a = cost.cpu().numpy().view(np.uint32).reshape(5, n).astype(np.int64)
    t_end, t_beg, wall, hw = a[0], a[1], a[2], a[3]
    base = t_beg.min()
    te = (t_end - base) & 0xffffffff; tb = (t_beg - base) & 0xffffffff
    wl = (wall - wall.min()) & 0xffffffff
    span = te.max()
    i0, i1 = np.argmin(te), np.argmax(te)
    clock = (te[i1] - te[i0]) / max(1, (wl[i1] - wl[i0])) * 100e6
    res = (te - tb).sum()
    print("kernel %.2f us by events; span %d ticks; clock from s_memtime/s_memrealtime %.3f GHz -> span %.2f us" % (k[0] * 1e3, span, clock / 1e9, span / clock * 1e6))
    print("  residency sum %d ticks = %.1f%% of span x 7168 slots; distinct hw ids %d" % (res, 100 * res / (span * 7168.0), len(np.unique(hw & 0xffff))))
    # occupancy over time: number of resident waves per 2 us bin
    bins = np.linspace(0, span, 41)
    occ = [(np.minimum(te, b1) - np.maximum(tb, b0)).clip(0).sum() / (b1 - b0) for b0, b1 in zip(bins[:-1], bins[1:])]
    print("  resident waves per 1/40 of the span:", " ".join("%d" % o for o in occ))
    print("  first tile begins at %d, last begins at %d ticks; tile begin->end mean %d" % (tb.min(), tb.max(), (te - tb).mean()))
np.save(os.path.join(ROOT, "gpurun_out", "clock_probe.npy"), a)
