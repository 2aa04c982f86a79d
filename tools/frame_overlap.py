"""Displayed frames (full path tracer + display pass) with one, two and three frames in flight: frame i's display pass on one stream while
frame i + 1 is traced on the next. The tracer is bound by vector issue, the display pass waits on LDS and memory: side by side they fill
each other's gaps. Every stream has its own three images; the last frame of each is compared with the oracle's committed displayed frame."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vrt_import
V = vrt_import.vrt()
import torch

name, W, H = "dragon", 1920, 1080
g = json.load(open(os.path.join(ROOT, "tests/golden/frames.json")))["frames"]["dragon_1080p_full/mode2"]
w = V.World(); assert w.load_vox(os.path.join(ROOT, "tests/golden/maps", name + ".vox"))
ctx = V.Context(0); ctx.upload_octree(*w.flatten())
p = g["pose"]
ip, iv, cp, _ = V.camera_block(p[:3], p[3], p[4], W, H); ctx.set_camera(ip, iv, cp)
dev = torch.device("cuda", 0)
for n_streams in (1, 2, 3, 4):
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    bufs = [(torch.zeros((H, W), dtype=torch.int32, device=dev), torch.zeros((H, W, 2), dtype=torch.int32, device=dev), torch.zeros((H, W), dtype=torch.int32, device=dev))
            for _ in range(n_streams)]
    def frame(k):
        s = streams[k % n_streams]; r, i, o = bufs[k % n_streams]
        ctx.dispatch_rows(W, H, 0, H, V.MODE_FULL, r.data_ptr(), i.data_ptr(), s.cuda_stream)
        ctx.denoise_device(W, H, r.data_ptr(), i.data_ptr(), o.data_ptr(), s.cuda_stream)
    for k in range(400): frame(k)
    torch.cuda.synchronize()
    n = 1000
    t0 = time.perf_counter()
    for k in range(n): frame(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    ok = all("%016x" % V.fnv1a64(b[2].cpu().numpy().view(np.uint8).reshape(H, W, 4)) == g["shown_fnv1a64"] for b in bufs)
    print("%d frame(s) in flight: %.4f ms per displayed frame, %.0f per second; frames equal the oracle's: %s" % (n_streams, dt * 1e3, 1.0 / dt, ok), flush=True)
