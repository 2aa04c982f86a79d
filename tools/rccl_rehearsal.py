"""Rehearsal of bench.py's N > 1 code path on ONE GPU with the real backend: torch.distributed is initialised with
backend "nccl" (= RCCL) and world_size 1, and the frame pipeline is told to run its collectives anyway, so every
call the 8-GPU run makes (asynchronous gather into a list of device tensors, wait, barrier, all_reduce of a device
double, side streams around them) goes through RCCL once before the driver's scaling run does."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import vrt_import  # noqa: E402

V = vrt_import.vrt()
shd = importlib.import_module("voxel-raytracer_amd.sharding")


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    W, H, steps = 1920, 1080, int(os.environ.get("VRT_REHEARSAL_STEPS", "40"))
    w = V.World()
    assert w.load_vox(os.path.join(ROOT, "tests/golden/maps/dragon.vox"))
    tex, dim = w.flatten()
    ctx = V.Context(0)
    ctx.upload_octree(tex, dim)
    ip, iv, cp, _ = V.camera_block((63.5, 60.5, 140.5), -90.0, -10.0, W, H)
    ctx.set_camera(ip, iv, cp)
    g = json.load(open(os.path.join(ROOT, "tests/golden/frames.json")))["frames"]["dragon_1080p/mode0"]
    plan = shd.ShardPlan(W, H, 8, 0, 1)
    for mode in ("final", "frame"):
        for streams in (1, 2, 4):
            pipe = shd.FramePipeline(plan, dev, gather=mode, streams=streams, collective=True)
            dist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                k, p_rgba, p_id = pipe.slot()
                ctx.dispatch_shard(W, H, 8, 0, 1, 0, p_rgba, p_id, pipe.stream_handle(k))
                pipe.submit(k)
            pipe.drain()
            dist.barrier()
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            fr, fi = pipe.frame_views()
            ok = ("%016x" % V.fnv1a64(fr.cpu().numpy().view("uint8").reshape(H, W, 4)) == g["rgba_fnv1a64"] and
                  "%016x" % V.fnv1a64(fi.cpu().numpy()) == g["id_dist_fnv1a64"])
            print("rccl world=1  gather=%-5s streams=%d  %.4f ms/step  pixels_match=%s" % (mode, streams, float(t.item()) / steps * 1e3, ok),
                  flush=True)
            assert ok
    dist.destroy_process_group()
    print("rccl rehearsal ok")


if __name__ == "__main__":
    main()
