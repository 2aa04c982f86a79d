"""Sums what tools/denoise_phases.sh's builds left in the tiles' first pixels: ticks of one phase of the display pass, first wave of every tile.
Usage (GPU box): VRT_HIP_LIB=.../libvrt_hip_phase<k>.so python tools/denoise_phases.py <k> [map W H]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vrt_import
V = vrt_import.vrt()
POSES = {"dragon": (63.5, 60.5, 140.5, -90.0, -10.0), "monu9": (48.5, 60.5, 170.5, -90.0, -12.0), "nature": (60.5, 80.5, 200.5, -90.0, -20.0)}
k = int(sys.argv[1])
name = sys.argv[2] if len(sys.argv) > 2 else "dragon"
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
w = V.World(); assert w.load_vox(os.path.join(ROOT, "tests/golden/maps", name + ".vox"))
ctx = V.Context(0); ctx.upload_octree(*w.flatten())
p = POSES[name]
ip, iv, cp, _ = V.camera_block(p[:3], p[3], p[4], W, H); ctx.set_camera(ip, iv, cp)
ctx.set_tile_scheduling(0)
rgba, idd = ctx.dispatch(W, H, 2)
for _ in range(3): out = ctx.denoise(rgba, idd)
t = out.view(np.uint32).reshape(H, W)[0::16, 0::32].astype(np.int64)
sky = (idd[..., 0].reshape(-1, W)[:H // 16 * 16].reshape(H // 16, 16, W)[:, :, :W // 32 * 32].reshape(H // 16, 16, W // 32, 32) != 0).any(axis=(1, 3))
t = t[:H // 16, :W // 32]
print("phase %d: tiles with work %d of %d; ticks per such tile mean %.0f  median %.0f  max %d" % (k, sky.sum(), t.size, t[sky].mean(), np.median(t[sky]), t[sky].max()))
