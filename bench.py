#!/usr/bin/env python3
"""Headline benchmark of the MI355X ray-casting path.

One "step" = one 1920x1080 frame of primary rays over maps/dragon.vox (BASELINE.json config 3, the
configuration its metric is quoted on), octree and camera already resident in HBM. With N GPUs the
frame's rows are dealt in 8-row tiles to the ranks (one process per GPU), each rank traces its
tiles, and the finished rows are gathered to rank 0 over RCCL -- the gather is inside the timed step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode primary|primary_shadow]
                    [--map dragon|monu9|nature] [--width 1920 --height 1080] [--variant V]

Rank 0 prints ONE JSON line (contract in the repository README / DESIGN.md section "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

POSES = {  # framing poses of SURVEY.md 8(d): x, y, z, yaw, pitch
    "dragon": (63.5, 60.5, 140.5, -90.0, -10.0),
    "monu9": (48.5, 60.5, 170.5, -90.0, -12.0),
    "nature": (60.5, 80.5, 200.5, -90.0, -20.0),
    "terrain": (512.5, 420.5, 1000.5, -90.0, -20.0),  # config 4: procedural 1024^2 heightfield shell
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# every n-th launch of the timed region carries events (odd: no beat with the scheduler's measuring launches, every 16th)
PROFILE_EVERY = int(os.environ.get("VRT_BENCH_PROFILE_EVERY", "7"))


def metric_name():
    """BASELINE.json's metric string, verbatim."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except (OSError, KeyError, ValueError):
        return "Mrays/s at 1920×1080 primary rays; achieved HBM GB/s vs peak"


def cpu_baseline(args, tex, dim, cam, budget_s=10.0):
    """Times the CPU restatement (oracle/, a scalar single-thread port of the same traversal) on whole
    frames of the same workload until ~budget_s of CPU work is done. Reported, never the target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    O.build()
    s = O.make_scene(tex, dim, *cam)
    mode = {"primary": 0, "primary_shadow": 1}[args.mode]
    W, H = args.width, args.height
    band = max(8, H // 8)
    t0 = time.perf_counter()
    rays = 0
    passes = 0
    r = 0
    # bounded sample: successive bands of rows of the bench frame, cycling over the frame
    while time.perf_counter() - t0 < budget_s:
        r0 = r % H
        r1 = min(H, r0 + band)
        O.render(s, W, H, mode, row0=r0, row1=r1)
        rays += (r1 - r0) * W
        r = r1 % H
        passes += 1
    dt = time.perf_counter() - t0
    out = {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
           "sample": f"{rays} primary rays of the same frame ({passes} bands of {band} rows, cycling) in {dt:.1f} s, "
                     f"oracle/rt_oracle.c -O2 -ffp-contract=off, 1 thread of {os.cpu_count()} host cores"}
    # the same port row-parallel: T threads, each tracing its own bands for ~mt_budget seconds
    import threading
    T = max(1, min(16, os.cpu_count() or 1))
    mt_budget = 6.0
    done = [0] * T

    def worker(k):
        tw = time.perf_counter()
        r_ = (k * band) % H
        while time.perf_counter() - tw < mt_budget:
            r1_ = min(H, r_ + band)
            O.render(s, W, H, mode, row0=r_, row1=r1_)
            done[k] += (r1_ - r_) * W
            r_ = (r1_ + (T - 1) * band) % H
    t1 = time.perf_counter()
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(T)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt_mt = time.perf_counter() - t1
    out["row_parallel"] = {"value": round(sum(done) / dt_mt / 1e6, 3), "unit": "Mrays/s", "cores": T,
                           "sample": f"{sum(done)} rays in {dt_mt:.1f} s on {T} threads"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", default="primary", choices=["primary", "primary_shadow"])
    ap.add_argument("--map", default="dragon", choices=sorted(POSES))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tile-rows", type=int, default=8)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket each launch with hipEvents (roofline is then omitted); for measuring their cost")
    ap.add_argument("--gather", default="final", choices=["final", "frame"],
                    help="final: traced frames stay sharded in their ranks' HBM (as a one-GPU run keeps them resident); the "
                         "last frame is gathered to rank 0 and assembled inside the timed region. frame: EVERY frame is "
                         "gathered to rank 0 (double-buffered) -- bounded by 12 B/pixel into one GPU's xGMI links")
    ap.add_argument("--streams", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="HIP streams the frames rotate through (the drain of one launch overlaps the start of the next "
                         "ones). 0 = 1 at one GPU, where the per-launch duration feeds the roofline and must not be "
                         "inflated by a neighbour, and 4 with more ranks, where a launch is a fraction of a frame (an "
                         "eighth of a frame: 25.7 us per launch on one stream, 13.7 on two, 9.2 on four)")
    ap.add_argument("--extras", action="store_true",
                    help="one GPU: after the timed region also time the same frames rotating through four streams "
                         "(overlapped_frames) and four per launch (batched_views). Off by default so that a rocprofv3 "
                         "kernel trace of the default command holds the timed region's launches only")
    ap.add_argument("--sched-period", type=int, default=-1,
                    help="feedback tile scheduling (vrt_set_tile_scheduling): every n-th launch of a shape measures its tiles "
                         "and the following ones start them heaviest first; 0 = off (row-major starts). Default: the "
                         "library's 16 up to two ranks; off from four ranks on, where a launch is a quarter of a frame or "
                         "less and four overlapping streams already fill its tail (tools/shard_rate.py: 17.6 vs 17.9 us at "
                         "a quarter, 8.9 vs 9.2 us at an eighth)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the measured path); gloo only rehearses the N>1 code path on a "
                         "box whose ranks share one GPU (collective staged through host memory)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import vrt_import
    V = vrt_import.vrt()
    shd = __import__("importlib").import_module("voxel-raytracer_amd.sharding")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.sched_period < 0:
        args.sched_period = 16 if world <= 2 else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ray-casting path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= n_dev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible (one process per GPU)")
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    via_host = args.backend == "gloo"

    # host side of the path: .vox -> octree -> texel stream; camera block (all through the product library)
    wld = V.World()
    if args.map == "terrain":
        wld.fill_terrain(1024, 1337)
    elif not wld.load_vox(os.path.join(ROOT, "tests", "golden", "maps", args.map + ".vox")):
        raise SystemExit("cannot load the scene fixture")
    tex, dim = wld.flatten()
    pose = POSES[args.map]
    W, H = args.width, args.height
    ip, iv, cp, _ = V.camera_block(pose[:3], pose[3], pose[4], W, H)
    mode = V.MODES[args.mode]

    ctx = V.Context(dev_index)
    ctx.set_variant(args.variant)
    ctx.set_tile_scheduling(args.sched_period)
    ctx.upload_octree(tex, dim)
    ctx.set_camera(ip, iv, cp)

    plan = shd.ShardPlan(W, H, args.tile_rows, rank, world)
    n_streams = args.streams or (1 if world == 1 else 4)
    pipe = shd.FramePipeline(plan, dev, stage_through_host=via_host, gather=args.gather, streams=n_streams)

    def step():
        k, p_rgba, p_id = pipe.slot()
        # on torch-owned streams, so a gather (and the final assembly) orders itself after the trace
        ctx.dispatch_shard(W, H, args.tile_rows, rank, world, mode, p_rgba, p_id, pipe.stream_handle(k))
        pipe.submit(k)

    def fence():
        pipe.drain()  # the newest frame (--gather frame: every frame) is gathered to rank 0 and assembled there
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    # every 7th launch of the timed region carries a hipEvent pair on its stream (a pair around EVERY launch keeps
    # consecutive launches from overlapping and costs ~6 % of the frame rate)
    ctx.set_profiling(0 if args.no_kernel_events else args.steps, every=PROFILE_EVERY)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = ctx.profile_read(args.steps)
    ctx.set_profiling(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if via_host else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    torch.cuda.synchronize(dev)

    # several GPUs: SURVEY 8(e) asks for the rate with AND without the per-frame gather. The timed region above is
    # the one without (frames stay sharded, or whatever --gather says); here the same K frames are each delivered to
    # rank 0, double-buffered. Reported beside the headline, not instead of it.
    delivered = None
    if world > 1 and args.gather == "final" and not os.environ.get("VRT_BENCH_NO_FRAME_GATHER"):
        pipe_f = shd.FramePipeline(plan, dev, stage_through_host=via_host, gather="frame", streams=n_streams)
        for it in range(args.warmup + args.steps):
            if it == args.warmup:
                pipe_f.drain()
                dist.barrier()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
            k, p_rgba, p_id = pipe_f.slot()
            ctx.dispatch_shard(W, H, args.tile_rows, rank, world, mode, p_rgba, p_id, pipe_f.stream_handle(k))
            pipe_f.submit(k)
        pipe_f.drain()
        dist.barrier()
        torch.cuda.synchronize(dev)
        ef = time.perf_counter() - t0
        t = torch.tensor([ef], dtype=torch.float64, device="cpu" if via_host else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ef = float(t.item())
        same = None
        if rank == 0:
            fr, fi = pipe_f.frame_views()
            same = bool(torch.equal(fr, pipe.frame_views()[0]) and torch.equal(fi, pipe.frame_views()[1]))
        delivered = {"gather": "frame", "value": round(W * H * args.steps / ef / 1e6, 2), "unit": "Mrays/s",
                     "ms_per_step": round(ef / args.steps * 1e3, 5), "same_pixels": same}

    # one GPU, informational: the same K frames rotating through four streams (no per-launch events; the figure the
    # headline would become if overlapped launches were allowed to blur the per-kernel duration the roofline uses)
    overlapped = None
    if args.extras and world == 1 and n_streams == 1:
        pipe2 = shd.FramePipeline(plan, dev, gather=args.gather, streams=4)
        for it in range(args.warmup + args.steps):
            if it == args.warmup:
                pipe2.drain()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
            k, p_rgba, p_id = pipe2.slot()
            ctx.dispatch_shard(W, H, args.tile_rows, rank, world, mode, p_rgba, p_id, pipe2.stream_handle(k))
            pipe2.submit(k)
        pipe2.drain()
        torch.cuda.synchronize(dev)
        e2 = time.perf_counter() - t0
        f2_rgba, f2_id = pipe2.frame_views()
        same = bool(torch.equal(f2_rgba, pipe.frame_views()[0]) and torch.equal(f2_id, pipe.frame_views()[1]))
        overlapped = {"streams": 4, "value": round(W * H * args.steps / e2 / 1e6, 2), "unit": "Mrays/s",
                      "ms_per_step": round(e2 / args.steps * 1e3, 5), "same_pixels": same}

    # one GPU, informational: four frames per launch (vrt_dispatch_views). The launch duration is again a clean
    # per-kernel figure (no neighbour on the GPU), so the roofline arithmetic of the contract applies to it as is
    batched = None
    if args.extras and world == 1 and n_streams == 1 and not args.no_kernel_events:
        F = 4
        bufs = [plan.local_buffer(dev) for _ in range(2 * F)]
        sets = [V.make_views([(ip, iv, cp) + plan.pointers(bufs[g * F + j]) for j in range(F)]) for g in range(2)]
        stream0 = torch.cuda.current_stream(dev).cuda_stream
        n_launch = (args.steps + F - 1) // F
        for g in range(max(2, args.warmup // F)):
            ctx.dispatch_views(W, H, args.tile_rows, rank, world, mode, sets[g & 1], stream0)
        torch.cuda.synchronize(dev)
        ctx.set_profiling(n_launch, every=2)
        t0 = time.perf_counter()
        for g in range(n_launch):
            ctx.dispatch_views(W, H, args.tile_rows, rank, world, mode, sets[g & 1], stream0)
        torch.cuda.synchronize(dev)
        e3 = time.perf_counter() - t0
        launch_ms = ctx.profile_read(n_launch)
        ctx.set_profiling(0)
        last = bufs[((n_launch - 1) & 1) * F + F - 1]
        same = bool(torch.equal(last, pipe.local[(pipe.frame - 1) % pipe.n_buf].view(-1)[: last.numel()]))
        batched = {"frames_per_launch": F, "value": round(W * H * n_launch * F / e3 / 1e6, 2), "unit": "Mrays/s",
                   "ms_per_step": round(e3 / (n_launch * F) * 1e3, 5), "same_pixels": same,
                   "launch_avg_ms": round(float(launch_ms.mean()), 5) if len(launch_ms) else None}

    if rank == 0:
        frames = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))["frames"]
        key = {"dragon": "dragon_1080p", "monu9": "monu9_720p", "nature": "nature_4k"}.get(args.map, "-") + f"/mode{mode}"
        g = frames.get(key)
        known = g is not None and (g["width"], g["height"]) == (W, H)
        frame_rgba, frame_id = pipe.frame_views()
        rgba_host = frame_rgba.cpu().numpy().view("uint8").reshape(H, W, 4)
        id_host = frame_id.cpu().numpy()
        check = None
        if known:
            check = ("%016x" % V.fnv1a64(rgba_host) == g["rgba_fnv1a64"] and
                     "%016x" % V.fnv1a64(id_host) == g["id_dist_fnv1a64"])
        rays = W * H
        ms_per_step = elapsed / args.steps * 1e3
        value = rays * args.steps / elapsed / 1e6
        roofline = None
        if known and len(kernel_ms):
            # algorithmic bytes of THIS rank's launch: 4 B per texel fetch the reference algorithm issues for
            # the rows it traces + 12 B per pixel written; exact per-row counts are committed
            if "row_fetches" in g:
                f_local = sum(g["row_fetches"][r] for r in plan.rows_of[0])
            else:
                f_local = g["fetches"] * plan.rows_local // H
            b_algo = 4 * f_local + 12 * W * plan.rows_local
            avg_ms = float(kernel_ms.mean())
            achieved = b_algo / (avg_ms * 1e-3) / 1e9
            # HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc.sh);
            # counters cannot be read from inside this process, so the committed figure for this exact
            # workload/variant is quoted, else null
            traffic = None
            try:
                tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                traffic = tr.get(f"{args.map}/{W}x{H}/{args.mode}/variant{args.variant}/gpus{world}", {}).get("bytes")
            except OSError:
                pass
            roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "kernel": "trace_kernel", "kernel_avg_ms": round(avg_ms, 5),
                        "algorithmic_bytes_per_launch": b_algo,
                        "compulsory_bytes_per_launch": int(tex.size + 12 * W * plan.rows_local)}
        if batched and roofline and batched["launch_avg_ms"]:
            b4 = batched["frames_per_launch"] * roofline["algorithmic_bytes_per_launch"]
            batched["roofline_frac"] = round(b4 / (batched["launch_avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out = {
            "metric": metric_name(),
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32",
            "data": (f"tests/golden/maps/{args.map}.vox scene fixture" if args.map != "terrain" else
                     "procedural 1024x1024 heightfield (vrth_world_fill_terrain, seed 1337)") + ", fixed synthetic camera pose",
            "config": {"workload": f"{args.map}.vox {W}x{H} {args.mode} rays, pose {pose}", "mode": args.mode,
                       "sharding": f"interleaved {args.tile_rows}-row tiles over {world} rank(s); " + (
                           "frames stay sharded in HBM, no collective per step; the last frame is gathered to rank 0 and "
                           "assembled inside the timed region" if args.gather == "final" else
                           "every frame gathered to rank 0 inside the timed region, double-buffered (gather of frame i "
                           "overlaps trace of frame i+1)"),
                       "gather": args.gather, "streams": n_streams, "tile_scheduling_period": args.sched_period,
                       "variant": args.variant, "collective_backend": args.backend if world > 1 else None},
            "roofline": roofline,
            "pixels_match_oracle_golden": check,
            "every_frame_delivered": delivered,
            "overlapped_frames": overlapped,
            "batched_views": batched,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, tex, dim, (ip, iv, cp))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
