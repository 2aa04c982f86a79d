"""Parity fuzz of the part-tile waves (VRT_OPT_HEAVY_TILES): random poses in and around the reference's room (and random translucent
worlds), VRT_MODE_FULL with the feedback scheduler re-measuring every other launch, every frame against the same context's
unscheduled frame (same kernel, whole tiles, row-major starts; that frame is held to the oracle by the parity suite)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vrt_import
V = vrt_import.vrt()
import conftest

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_poses = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(seed)
t0 = time.time()
frames = split_frames = 0
def run(ctx, W, H, label):
    global frames, split_frames
    d_rgba = ctx.device_alloc(W * H * 4); d_id = ctx.device_alloc(W * H * 8)
    ctx.set_tile_scheduling(0)
    ctx.dispatch_rows(W, H, 0, H, V.MODE_FULL, d_rgba, d_id)
    ref = (ctx.device_read(d_rgba, (H, W), np.uint32), ctx.device_read(d_id, (H, W, 2), np.int32))
    ctx.set_tile_scheduling(2)
    for k in range(5):
        ctx.device_write(d_rgba, np.zeros((H, W), np.uint32)); ctx.device_write(d_id, np.zeros((H, W, 2), np.int32))
        ctx.dispatch_rows(W, H, 0, H, V.MODE_FULL, d_rgba, d_id)
        got = (ctx.device_read(d_rgba, (H, W), np.uint32), ctx.device_read(d_id, (H, W, 2), np.int32))
        n = ctx.sched_split_count()
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), (label, k, n)
        frames += 1
        split_frames += 1 if (n > 0 and k >= 2) else 0
    ctx.device_free(d_rgba); ctx.device_free(d_id)

room = conftest.room_world(V).flatten()
ctx = V.Context(0)
ctx.upload_octree(*room)
for i in range(n_poses):
    W, H = [(960, 540), (1280, 720), (1024, 512), (1016, 520)][i % 4]
    inside = rng.random() < 0.6
    pos = rng.uniform((4, 22, 4), (44, 44, 44)) if inside else rng.uniform((-40, 10, -40), (110, 70, 110))
    yaw, pitch = rng.uniform(-180, 180), rng.uniform(-50, 30)
    ip, iv, cp, _ = V.camera_block(tuple(float(x) for x in pos), float(yaw), float(pitch), W, H)
    ctx.set_camera(ip, iv, cp)
    run(ctx, W, H, ("room", i, tuple(pos), yaw, pitch))
    if i % 25 == 24: print("room poses %d, frames %d (part-tile waves on %d), %.0f s" % (i + 1, frames, split_frames, time.time() - t0), flush=True)
ctx.close()
# random worlds with glass: blobs of translucent voxels around opaque cores
for i in range(max(4, n_poses // 10)):
    w = V.World()
    n = int(rng.integers(2000, 12000))
    xyz = rng.integers(8, 56, size=(n, 3), dtype=np.int32)
    palette = np.array([0x50b43cff, 0x644628ff, 0x3c64dc96, 0xc8dcff50, 0xffd2d240, 0xa0a0a0ff], np.uint32)
    w.insert_many(xyz, palette[rng.integers(0, len(palette), size=n)], refraction=float(rng.choice([1.33, 1.5, 2.4])), illumination=0.0, k=0.0)
    ctx = V.Context(0)
    ctx.upload_octree(*w.flatten())
    for j in range(4):
        W, H = 960, 540
        pos = rng.uniform((-20, 10, -20), (84, 70, 84))
        yaw, pitch = rng.uniform(-180, 180), rng.uniform(-40, 20)
        ip, iv, cp, _ = V.camera_block(tuple(float(x) for x in pos), float(yaw), float(pitch), W, H)
        ctx.set_camera(ip, iv, cp)
        run(ctx, W, H, ("world", i, j))
    ctx.close()
print("seed %d: %d scheduled frames equal to their unscheduled frames, part-tile waves active on %d of them, %.0f s" % (seed, frames, split_frames, time.time() - t0))
