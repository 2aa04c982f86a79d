#!/usr/bin/env python3
"""Headline benchmark of the MI355X ray-casting path.

One "step" = one 1920x1080 frame of primary rays over maps/dragon.vox (BASELINE.json config 3, the
configuration its metric is quoted on), octree and camera already resident in HBM. With N GPUs the
frame's rows are dealt in 8-row tiles to the ranks (one process per GPU), each rank traces its
tiles, and EVERY frame is delivered to rank 0 over RCCL inside the timed region (double-buffered);
the rate with the frames left sharded in the ranks' HBM is printed beside it (`sharded_resident`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode primary|primary_shadow|full]
                    [--map dragon|monu9|nature|terrain|terrain_full|room] [--width 1920 --height 1080] [--variant V]

`--gpus N` with N > 1 and no torchrun environment starts the N ranks itself (a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...`, before this process touches the
GPU) and relays rank 0's line; started under torchrun it is one of the ranks. Rank 0 prints ONE
JSON line (contract in the repository README / DESIGN.md section "Measurement").

Timing: W warm-up frames, fence, K timed frames, fence. That region runs twice: first as the
process's first GPU work (reported as `cold_start`), then -- the headline `value` -- behind 512
untimed frames that bring the GPU out of its idle clocks (`config.preroll_launches`;
VRT_BENCH_PREROLL=0 makes the first region the headline). `roofline` is the contract's HBM line
(reference-requested bytes: cache-served, may pass 1); `issue_roofline` is the bound that binds.
With N > 1 the headline is the faster verified of two ways of delivering every frame to rank 0
(`config.delivery`: an RCCL gather per frame, or the kernels' own stores through IPC mappings of
rank 0's frame); both, the rotating-root form and the sharded-resident rate are in the line, and so is
`whole_frame_per_gpu`: N whole frames per step, one per GPU, kept where they were traced (weak scaling).

At one GPU the line also carries `configs`: every other BASELINE.json configuration (monu9 720p, the config-4 terrain at its
full extent and as the window the texel format holds, nature 4K + shadow rays), the full path tracer, the displayed frame
and the reference's translucent room, each timed the same way (pre-roll, W warm-up, K timed frames, hipEvent pairs on the
kernel) and checked against the oracle's golden hashes, and config 1's CPU ray cast (the host library's octree_ray_cast,
256 x 256). `--no-configs` leaves them out (profiling runs of the headline kernel).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

POSES = {  # framing poses of SURVEY.md 8(d): x, y, z, yaw, pitch
    "dragon": (63.5, 60.5, 140.5, -90.0, -10.0),
    "monu9": (48.5, 60.5, 170.5, -90.0, -12.0),
    "nature": (60.5, 80.5, 200.5, -90.0, -20.0),
    "terrain": (512.5, 420.5, 1000.5, -90.0, -20.0),  # config 4: tests/golden/terrain.json
    "terrain_full": (512.5, 420.5, 1000.5, -90.0, -20.0),   # config 4 at its named extent (records path, no texel stream)
    "room": (14.5, 30.5, 16.5, 32.0, -10.0),                # tests/golden/room.json "inside"
}
GOLDEN_KEY = {"dragon": "dragon_1080p", "monu9": "monu9_720p", "nature": "nature_4k", "terrain": "terrain_1080p",
              "terrain_full": "terrain_full_1080p", "room": "room_inside_1080p"}
# the BASELINE.json configurations beside the headline one (configs[2]) and the rows SURVEY 8(f) added, in the order timed:
# (name, map, width, height, mode); mode "shown" = full path tracer + display pass (what the reference puts on screen)
CONFIGS = [
    ("config2_monu9_720p_primary", "monu9", 1280, 720, "primary"),
    ("config3_dragon_1080p_primary_shadow", "dragon", 1920, 1080, "primary_shadow"),
    ("dragon_1080p_full_path_tracer", "dragon", 1920, 1080, "full"),
    ("dragon_1080p_displayed_frame", "dragon", 1920, 1080, "shown"),
    ("config5_nature_4k_primary_shadow", "nature", 3840, 2160, "primary_shadow"),
    ("config4_terrain_window_1080p_primary", "terrain", 1920, 1080, "primary"),
    ("config4_terrain_full_1080p_primary", "terrain_full", 1920, 1080, "primary"),
    ("config4_terrain_full_1080p_primary_shadow", "terrain_full", 1920, 1080, "primary_shadow"),
    ("room_inside_1080p_full_path_tracer", "room", 1920, 1080, "full"),
]
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
KERNEL_SAMPLES = 16    # launches of the timed region that carry a hipEvent pair (every max(2, steps // 16)-th: >= 10 of 20)


def metric_name():
    """BASELINE.json's metric string, verbatim."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except (OSError, KeyError, ValueError):
        return "Mrays/s at 1920×1080 primary rays; achieved HBM GB/s vs peak"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", default="primary", choices=["primary", "primary_shadow", "full"])
    ap.add_argument("--map", default="dragon", choices=sorted(POSES))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tile-rows", type=int, default=8)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="one GPU: leave the `configs` array (the other BASELINE configurations) out of the line")
    ap.add_argument("--no-ray-tables", action="store_true",
                    help="A/B: every launch runs the shader's own ray-generation prologue (vrt_set_option(VRT_OPT_RAY_TABLES, 0))")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket launches with hipEvents (roofline is then omitted); for measuring their cost")
    ap.add_argument("--gather", default="auto", choices=["auto", "final", "frame"],
                    help="what the HEADLINE region does with a traced frame. frame: EVERY frame is gathered to rank 0 "
                         "(double-buffered: the gather of frame i overlaps the trace of frame i+1) -- the default with "
                         "several ranks. final: frames stay sharded in their ranks' HBM (as a one-GPU run keeps them "
                         "resident) and only the last one is gathered and assembled, inside the timed region -- the "
                         "default at one rank, where both are the same thing. With several ranks the other mode is timed "
                         "too and printed beside the headline")
    ap.add_argument("--streams", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="HIP streams the frames rotate through (the drain of one launch overlaps the start of the next "
                         "ones). 0 = 1 at one GPU, where the per-launch duration feeds the roofline and must not be "
                         "inflated by a neighbour, and 4 with more ranks, where a launch is a fraction of a frame")
    ap.add_argument("--extras", action="store_true",
                    help="one GPU: after the timed region also time the same frames rotating through four streams "
                         "(overlapped_frames) and four per launch (batched_views). Off by default so that a rocprofv3 "
                         "kernel trace of the default command holds the timed region's launches only")
    ap.add_argument("--sched-period", type=int, default=-1,
                    help="feedback tile scheduling (vrt_set_tile_scheduling): every n-th launch of a shape measures its tiles "
                         "and the following ones start them heaviest first; 0 = off (row-major starts). Default: the "
                         "library's 16 up to two ranks; off from four ranks on (a launch is a quarter of a frame or less "
                         "and four overlapping streams already fill its tail)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the measured path); gloo only rehearses the N>1 code path on a "
                         "box whose ranks share one GPU (collective staged through host memory)")
    ap.add_argument("--dry-plan", action="store_true",
                    help="no GPU work: start/join the ranks exactly as a real run does, build each rank's shard plan, check "
                         "it across ranks (with --backend gloo through a real process group) and print the plan as JSON")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args, argv):
    """--gpus N > 1 outside torchrun: this process has touched no GPU; it starts the N ranks as a CHILD process tree and
    relays rank 0's JSON line and the exit code (never exec: a process that has initialised the GPU must not be replaced)."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, VRT_BENCH_SPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for line in p.stdout:
        if line.startswith("{"):
            lines.append(line)
        else:
            sys.stderr.write(line)
    rc = p.wait()
    for line in lines:
        sys.stdout.write(line)
    sys.stdout.flush()
    return rc


def host_threads():
    """threads the row-parallel CPU leg runs: every core open to this process -- the affinity mask, cut to the cgroup's CPU quota
    where one is set (a box may show 256 logical CPUs to a container that may use 16)"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(args, tex, dim, cam, budget_s=10.0):
    """Times the CPU restatement (oracle/, a scalar port of the same traversal, -O3 -ffp-contract=off) on bands of rows of the
    same frame: one thread for ~budget_s, then row-parallel on every core open to the process. Reported, never the target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctypes as C
    import threading
    import numpy as np
    import oracle_py as O
    O.build()
    L = O.lib()
    s = O.make_scene(tex, dim, *cam)
    mode = {"primary": 0, "primary_shadow": 1, "full": 2}[args.mode]
    W, H = args.width, args.height
    band = max(8, H // 8)

    def trace_bands(first_row, stride_rows, seconds):
        """bands of `band` rows starting at first_row, advancing by stride_rows (cycling over the frame), into this thread's own
        images (allocated once: a fresh 25 MB frame per call would measure the page faults, not the traversal)"""
        rgba = np.zeros((H, W, 4), np.uint8)
        idd = np.zeros((H, W, 2), np.int32)
        st = O.Stats()
        t_end = time.perf_counter() + seconds
        rays, passes, r = 0, 0, first_row % H
        while time.perf_counter() < t_end:
            r1 = min(H, r + band)
            L.o_render(C.byref(s), W, H, r, r1, mode, rgba.ctypes.data, idd.ctypes.data, None, C.byref(st))
            rays += (r1 - r) * W
            passes += 1
            r = (r1 + stride_rows) % H if r1 < H else stride_rows % H
        return rays, passes

    t0 = time.perf_counter()
    rays, passes = trace_bands(0, 0, budget_s)
    dt = time.perf_counter() - t0
    out = {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
           "sample": f"{rays} primary rays of the same frame ({passes} bands of {band} rows, cycling) in {dt:.1f} s, "
                     f"oracle/rt_oracle.c -O3 -ffp-contract=off, 1 thread; the box shows {os.cpu_count()} logical CPUs, "
                     f"{host_threads()} open to this process (affinity and cgroup quota)"}
    # the same port row-parallel on every core the process may use
    T = host_threads()
    mt_budget = 6.0
    done = [0] * T

    def worker(k):
        done[k] = trace_bands(k * band, (T - 1) * band, mt_budget)[0]
    t1 = time.perf_counter()
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(T)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt_mt = time.perf_counter() - t1
    out["row_parallel"] = {"value": round(sum(done) / dt_mt / 1e6, 3), "unit": "Mrays/s", "cores": T,
                           "sample": f"{sum(done)} rays in {dt_mt:.1f} s on {T} threads (every core open to the process)"}
    return out


def load_world(V, name):
    """host side of the path through the product library: .vox (or the config-4 height field) -> octree -> texel stream"""
    wld = V.World()
    import numpy as np
    if name == "room":   # the reference's translucent room (src/main.cpp:505-633) as its ordered insert list
        d = np.load(os.path.join(ROOT, "tests", "golden", "room.npz"))
        mats = json.load(open(os.path.join(ROOT, "tests", "golden", "room.json")))["materials"]
        for (x, y, z), c, m in zip(d["xyz"], d["color"], d["material"]):
            mm = mats[int(m)]
            wld.insert(int(x), int(y), int(z), int(c), mm["refraction"], mm["illumination"], mm["k"])
    elif name in ("terrain", "terrain_full"):
        tj = json.load(open(os.path.join(ROOT, "tests", "golden", "terrain.json")))
        wd = tj["window"] if name == "terrain" else {"x0": 0, "z0": 0, "nx": 1024, "nz": 1024}
        wld.fill_heights(np.load(os.path.join(ROOT, "tests", "golden", "terrain_heights.npz"))["heights"],
                         wd["x0"], wd["z0"], wd["nx"], wd["nz"], tj["band"], tj["floor"])
    elif not wld.load_vox(os.path.join(ROOT, "tests", "golden", "maps", name + ".vox")):
        raise SystemExit("cannot load the scene fixture")
    return wld


def upload_world(ctx, wld, name):
    """-> (bytes of the reference's texel stream for this tree, tex_dim). The full config-4 field is beyond the stream's 2^23
    texels: it goes up as records (vrth_world_records -> vrt_upload_records)."""
    if name == "terrain_full":
        rec, dim = wld.records()
        ctx.upload_records(rec, dim)
        return 4 * wld.texel_count(), dim
    tex, dim = wld.flatten()
    ctx.upload_octree(tex, dim)
    return int(tex.size), dim


# Untimed frames before the W warm-up steps of the headline region, to take the GPU out of its idle power state: a
# default driver run (W = 5, K = 20) is 1.6 ms of GPU work in all, and the kernel keeps getting faster for ~30 ms after
# the first launch (profiles/r02_f_preroll.txt: 0.0638 ms per launch right away, 0.0612 after 64 frames, 0.0582 after
# 512, 0.0578 after 4096). Disclosed in the JSON line (config.preroll_launches), with the region WITHOUT them beside it
# (`cold_start`); VRT_BENCH_PREROLL=0 makes that region the headline.
PREROLL = int(os.environ.get("VRT_BENCH_PREROLL", "512"))


def lib_stamp(V):
    """What the quoted PMC figures are tied to: the compiled tracer kernels -- a hash of the three launch objects the shipped library is
    linked from (csrc/build/vrt_launch_{primary,shadow,full}.o; hipcc output is reproducible, comments and the display pass do not
    enter). None when the objects are not there (a library built some other way): nothing is dropped then."""
    import hashlib
    h = hashlib.sha256()
    build = os.path.join(ROOT, "voxel-raytracer_amd", "csrc", "build")
    try:
        for f in ("vrt_launch_primary.o", "vrt_launch_shadow.o", "vrt_launch_full.o"):
            h.update(open(os.path.join(build, f), "rb").read())
    except OSError:
        return None
    return h.hexdigest()[:16]


def quoted(name):
    """profiles/<name> as JSON, or {} -- counters cannot be read from inside this process, so per-launch PMC figures are QUOTED
    from the committed collection of this exact workload (tools/pmc_issue.sh -> tools/issue_model.py)"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return {}


def issue_roofline(V, workload_key, kernel_ms_mean, share=1.0):
    """The bound that binds: vector-instruction issue. A gfx950 SIMD issues one wave64 vector instruction per `slot` cycles
    whatever its kind (tools/micro/valu_rate on SIMDs that verifiably held 8 waves: 2.25-2.35 cycles for full-rate streams and,
    per instruction, for full + half-rate mixes; the half-rate kinds also occupy a second pipe for 4.1-4.3 cycles each, which
    binds only above a ~55 % share; MI355X_MICROARCH.md quotes 2 cycles). With N vector instructions per launch (PMC
    SQ_INSTS_VALU), S = 1,024 SIMDs and C = the cycles the launch's waves were on the chip (SQ_BUSY_CYCLES / 32 shader engines):
        frac = N * slot / (S * C)       -- counters only, no clock: the share of the launch the vector issue port is busy.
    `frac_this_run` prices the same N at the clock the kernel ran at under the collection against THIS run's measured duration.
    Inputs: profiles/r03_issue_model.json (tools/issue_model.py from profiles/r03_pmc_summary.json, r03_valu_rate.json, isa_cost.py)."""
    m = quoted("r03_issue_model.json")
    e = m.get("workloads", {}).get(workload_key)
    if not e or not e.get("busy_cycles"):
        return None
    if m.get("lib_stamp") and lib_stamp(V) and m["lib_stamp"] != lib_stamp(V):
        return {"stale": True, "note": "profiles/r03_issue_model.json was collected on different kernel sources; not quoted"}
    n_valu = e["valu_insts_per_launch"] * share
    t_slot = n_valu * m["issue_slot_cycles"] / m["simds"] / (e["clock_ghz"] * 1e9) * 1e3
    return {"bound": "valu-issue", "frac": e["frac_under_pmc"], "frac_at_2_cycles": e["frac_at_2_cycles_under_pmc"],
            "frac_half_rate_pipe": e["half_pipe_frac_under_pmc"], "frac_scalar_port": e["scalar_port_frac_under_pmc"],
            "frac_this_run": round(t_slot / kernel_ms_mean, 4),
            "frac_is": "counters only (SQ_INSTS_VALU x issue slot / (1024 SIMDs x SQ_BUSY_CYCLES / 32)), same launch, profiled run",
            "valu_insts_per_launch": int(n_valu), "salu_insts_per_launch": int(e["salu_insts_per_launch"] * share),
            "busy_cycles_per_launch": e["busy_cycles"], "half_rate_share": e["half_rate_share"],
            "issue_slot_cycles": m["issue_slot_cycles"], "half_pipe_cycles": m["half_pipe_cycles"], "salu_cycles": m["salu_cycles"],
            "clock_ghz_under_pmc": e["clock_ghz"], "kernel_ms_under_pmc": e.get("kernel_ms_under_pmc"), "simds": m["simds"],
            "issue_time_ms_at_pmc_clock": round(t_slot, 5),
            "quoted_from": "profiles/r03_issue_model.json (tools/issue_model.py: rocprofv3 --pmc of this workload, tools/micro/valu_rate, "
                           "tools/isa_cost.py class shares)"}


def roofline_block(V, workload_key, b_algo, compulsory, kernel_ms, n_gpus=1):
    """The contract's HBM line. `achieved` = the bytes this algorithm must move per launch (the tree once + 12 B per pixel:
    the wide layout is cache resident, nothing else reaches HBM) / the kernel's duration; `traffic` = measured HBM bytes (PMC,
    quoted). The reference-requested figure of SURVEY 8(d) (4 B x every texel fetch raytracing.comp issues + 12 B/pixel) is
    reported beside it as `reference_fetch_rate`: those fetches are served from cache, so that rate passes the HBM peak and is
    not a fraction of any bound."""
    import numpy as np
    ms = np.asarray(kernel_ms, dtype=np.float64)
    avg_ms = float(ms.mean())
    tr = quoted("pmc_traffic.json").get(workload_key + f"/gpus{n_gpus}", {})
    traffic = tr.get("bytes")
    achieved = compulsory / (avg_ms * 1e-3) / 1e9
    ref_rate = b_algo / (avg_ms * 1e-3) / 1e9 if b_algo else None
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_quoted_from": tr.get("source") if traffic else None,
            "traffic_frac_of_peak": round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
            "bytes_are": "compulsory bytes of this algorithm: the uploaded tree once + 12 B per pixel written; the tree is cache "
                         "resident (L1 99 % hits), so the kernel is NOT HBM-bound -- see issue_roofline for the bound that binds",
            "algorithmic_bytes_per_launch": int(compulsory),
            "reference_fetch_rate": None if not b_algo else {
                "bytes_per_launch": int(b_algo), "rate_gbs": round(ref_rate, 1), "ratio_to_hbm_peak": round(ref_rate / HBM_PEAK_GBS, 4),
                "what": "SURVEY 8(d) B_algo = 4 B x texel fetches the REFERENCE shader issues (oracle count, frames.json) + 12 B/pixel; "
                        "cache-served requests the wide layout never issues: above 1 it only says the kernel outruns streaming them from HBM"},
            "kernel": "trace_kernel", "kernel_avg_ms": round(avg_ms, 5), "kernel_median_ms": round(float(np.median(ms)), 5),
            "kernel_min_ms": round(float(ms.min()), 5), "kernel_samples": int(ms.size)}


def time_configs(args, V, ctx, torch, dev, worlds):
    """The other BASELINE configurations on this GPU, each timed like the headline: pre-roll, W warm-up frames, fence, K timed
    frames, fence; hipEvent pairs on every max(2, K/16)-th launch; the last frame checked against the oracle's golden hash."""
    import numpy as np
    frames = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))["frames"]
    out = []
    loaded = None
    stream = torch.cuda.current_stream(dev).cuda_stream
    preroll = min(PREROLL, 256)
    for name, mp, W, H, mode_name in CONFIGS:
        t_cfg = time.perf_counter()
        try:
            if loaded != mp:
                if mp not in worlds:
                    worlds[mp] = load_world(V, mp)
                stream_bytes, _ = upload_world(ctx, worlds[mp], mp)
                loaded = mp
            pose = POSES[mp]
            ip, iv, cp, _ = V.camera_block(pose[:3], pose[3], pose[4], W, H)
            ctx.set_camera(ip, iv, cp)
            shown = mode_name == "shown"
            mode = 2 if shown else V.MODES[mode_name]
            rgba = torch.zeros(H * W, dtype=torch.int32, device=dev)
            idd = torch.zeros(H * W * 2, dtype=torch.int32, device=dev)
            disp = torch.zeros(H * W, dtype=torch.int32, device=dev) if shown else None

            def step():
                ctx.dispatch_rows(W, H, 0, H, mode, rgba.data_ptr(), idd.data_ptr(), stream)
                if shown:
                    ctx.denoise_device(W, H, rgba.data_ptr(), idd.data_ptr(), disp.data_ptr(), stream)

            for _ in range(preroll + args.warmup):
                step()
            torch.cuda.synchronize(dev)
            every = max(2, args.steps // KERNEL_SAMPLES) if args.steps >= 4 else 1
            ctx.set_profiling(args.steps, every=every)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
            kms = ctx.profile_read(args.steps)
            ctx.set_profiling(0)
            key = GOLDEN_KEY[mp] + ("_full" if mode == 2 else "") + f"/mode{mode}"
            g = frames.get(key)
            ok = None
            if g is not None and (g["width"], g["height"]) == (W, H):
                ok = ("%016x" % V.fnv1a64(rgba.cpu().numpy()) == g["rgba_fnv1a64"] and
                      "%016x" % V.fnv1a64(idd.cpu().numpy()) == g["id_dist_fnv1a64"])
                if shown:
                    ok = ok and "%016x" % V.fnv1a64(disp.cpu().numpy()) == g.get("shown_fnv1a64")
            wkey = f"{mp}/{W}x{H}/{'full' if shown else mode_name}/variant0"
            e = {"name": name, "workload": f"{mp} {W}x{H} {mode_name}, pose {pose}", "ms_per_step": round(el / args.steps * 1e3, 5),
                 "value": round(W * H * args.steps / el / 1e6, 2), "unit": "Mrays/s (primary rays per second)" if not shown else "Mpixels/s displayed",
                 "pixels_match_oracle_golden": ok, "preroll_launches": preroll, "steps": args.steps, "warmup": args.warmup}
            if len(kms):
                b_algo = g["b_algo_bytes"] if g else None
                e["roofline"] = roofline_block(V, wkey, b_algo, stream_bytes + 12 * W * H, kms)
                e["issue_roofline"] = issue_roofline(V, wkey, float(np.mean(kms)))
                if shown:
                    e["kernel_note"] = "kernel_avg_ms is the path tracer's launch; ms_per_step also holds the display pass (vrt_denoise)"
            out.append(e)
            del rgba, idd, disp
        except Exception as ex:  # noqa: BLE001 -- one configuration must not take the line with it
            out.append({"name": name, "error": f"{type(ex).__name__}: {ex}"[:300]})
        out[-1]["wall_s_incl_setup"] = round(time.perf_counter() - t_cfg, 2)
    return out


def config1_cpu_cast(V):
    """BASELINE config 1: custom.vox stand-in (the reference's is missing), 256 x 256, the host library's
    octree_ray_cast (reference src/octree.cpp:405-485) once per pixel, single thread, no GPU."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import config1_cpu as c1
    w = V.World()
    ok, n = w.load_vox_bytes(V.make_custom_vox())
    W = H = 256
    pos = (32.5, 40.5, 150.5)
    ip, iv, _, _ = V.camera_block(pos, -90.0, -8.0, W, H)
    dirs = c1.world_dirs(ip, iv, W, H).reshape(-1, 3)
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        hit, _ = w.ray_cast_many(pos, dirs)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return {"name": "config1_custom_vox_256x256_cpu_octree_ray_cast", "workload": "synthesized custom.vox 64^3 (SURVEY 8(d)), 256x256, "
            "host library octree_ray_cast per pixel (vrth_world_ray_cast_many), 1 thread", "voxels": n, "ms_per_step": round(best * 1e3, 3),
            "value": round(W * H / best / 1e6, 3), "unit": "Mrays/s", "hit_fraction": round(float((hit == 1).mean()), 4), "cores": 1,
            "checked_by": "tests/test_host.py::test_cpu_ray_cast_matches_oracle_config1 (every pixel vs the oracle)"}


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    in_torchrun = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if not in_torchrun and args.gpus > 1:
        sys.exit(spawn_ranks(args, argv))   # BEFORE anything touches the GPU or loads libvrt_hip.so
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a line for a different job size",
                  file=sys.stderr)
        sys.exit(2)
    if args.sched_period < 0:
        args.sched_period = 16 if world <= 2 else 0
    gather = args.gather if args.gather != "auto" else ("final" if world == 1 else "frame")
    launcher = "self-spawned" if os.environ.get("VRT_BENCH_SPAWNED") else ("torchrun" if in_torchrun else "single")

    import torch
    import torch.distributed as dist
    import vrt_import
    V = vrt_import.vrt()
    shd = __import__("importlib").import_module("voxel-raytracer_amd.sharding")
    W, H = args.width, args.height
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

    if args.dry_plan:
        plan = shd.ShardPlan(W, H, args.tile_rows, rank, world)
        checked = None
        if world > 1 and args.backend == "gloo":
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            t = torch.tensor([plan.rows_local, rank], dtype=torch.int64)
            dist.all_reduce(t)
            checked = bool(int(t[0]) == H and int(t[1]) == world * (world - 1) // 2)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_plan": True, "n_gpus": world, "collective_backend": args.backend if world > 1 else None,
                              "launcher": launcher, "gather": gather, "streams": args.streams or (1 if world == 1 else 4),
                              "tile_scheduling_period": args.sched_period, "rows_per_rank": [len(r) for r in plan.rows_of],
                              "rows_sum_checked_across_ranks": checked}), flush=True)
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ray-casting path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= n_dev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible (one process per GPU)")
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    via_host = args.backend == "gloo"
    rank_census = None
    if world > 1:   # who is really there: every rank's device ordinal and the device's UUID
        try:
            me = {"rank": rank, "local_rank": local_rank, "device": dev_index, "uuid": str(torch.cuda.get_device_properties(dev_index).uuid)}
        except Exception:  # noqa: BLE001
            me = {"rank": rank, "local_rank": local_rank, "device": dev_index, "uuid": None}
        allr = [None] * world
        dist.all_gather_object(allr, me)
        rank_census = {"ranks": len(allr), "distinct_devices": len({(r or {}).get("uuid") or f"dev{(r or {}).get('device')}" for r in allr}),
                       "devices": allr}

    wld = load_world(V, args.map)
    pose = POSES[args.map]
    ip, iv, cp, _ = V.camera_block(pose[:3], pose[3], pose[4], W, H)
    mode = V.MODES[args.mode]

    ctx = V.Context(dev_index)
    ctx.set_variant(args.variant)
    if args.no_ray_tables:
        ctx.set_ray_tables(False)
    if os.environ.get("VRT_BENCH_NO_ROOT0_ONLY"):   # A/B: 1 = rays that leave wide root 0 walk the empty octants' records; 2 = the shortcut
        ctx.set_root0_only({"1": 0, "2": 2}.get(os.environ["VRT_BENCH_NO_ROOT0_ONLY"], 0))   # without the tighter root
    if os.environ.get("VRT_BENCH_TWO_PASS") is not None:   # A/B: vrt_set_option(VRT_OPT_FULL_OPAQUE, value): 0 the general full path tracer, 6 (default) the stack-free kernel for opaque scenes, 1 two kernels
        ctx.set_option(V.OPT_FULL_OPAQUE, int(os.environ["VRT_BENCH_TWO_PASS"]))
    ctx.set_tile_scheduling(args.sched_period)
    stream_bytes, dim = upload_world(ctx, wld, args.map)
    ctx.set_camera(ip, iv, cp)

    plan = shd.ShardPlan(W, H, args.tile_rows, rank, world)
    n_streams = args.streams or (1 if world == 1 else 4)

    def timed_region(gather_mode, profile, preroll=0, whole=False):
        """[preroll untimed frames,] W warm-up frames, fence, K timed frames, fence; MAX over ranks.
        Returns (seconds, pipeline, kernel ms samples). whole: every rank traces WHOLE frames of its own and keeps them
        (a one-rank plan per GPU, no exchange at all); the fences and the MAX over ranks stay."""
        if whole:
            pipe = shd.FramePipeline(shd.ShardPlan(W, H, args.tile_rows, 0, 1), dev, gather="final", streams=1, collective=False)
        else:
            pipe = shd.FramePipeline(plan, dev, stage_through_host=via_host, gather=gather_mode, streams=n_streams)
        s_rank, s_world = (0, 1) if whole else (rank, world)

        def step():
            k, p_rgba, p_id = pipe.slot()
            # on torch-owned streams, so a gather (and the final assembly) orders itself after the trace
            ctx.dispatch_shard(W, H, args.tile_rows, s_rank, s_world, mode, p_rgba, p_id, pipe.stream_handle(k))
            pipe.submit(k)

        def fence():
            pipe.drain()  # the newest frame (gather "frame": every frame) is on rank 0, assembled
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)

        for _ in range(preroll):   # see PREROLL
            step()
        for _ in range(args.warmup):
            step()
        fence()
        # a hipEvent pair rides on every n-th launch of the timed region (a pair on EVERY launch of a long run keeps
        # consecutive launches from overlapping and costs a few percent of the frame rate)
        every = max(2, args.steps // KERNEL_SAMPLES) if args.steps >= 4 else 1
        ctx.set_profiling(args.steps if profile else 0, every=every)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        el = time.perf_counter() - t0
        ms = ctx.profile_read(args.steps) if profile else []
        ctx.set_profiling(0)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cpu" if via_host else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        torch.cuda.synchronize(dev)
        return el, pipe, ms

    # the contract's region as the first GPU work of the process (W warm-up frames, K timed ones: 1.6 ms of GPU time at the
    # driver's W = 5, K = 20, while the clocks are still coming up), reported as `cold_start`; then the headline region
    # behind PREROLL untimed frames
    cold = None
    if PREROLL > 0:
        ec, pipe_c, _ = timed_region(gather, False)
        cold = {"value": round(W * H * args.steps / ec / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(ec / args.steps * 1e3, 5),
                "what": "the same W warm-up + K timed frames as the process's first GPU work, no untimed frames before them"}
        del pipe_c
    elapsed, pipe, kernel_ms = timed_region(gather, not args.no_kernel_events, PREROLL)

    # several GPUs: SURVEY 8(e) asks for the rate with AND without the per-frame gather: the other mode, same K frames
    other = None
    if world > 1 and not os.environ.get("VRT_BENCH_ONE_REGION"):
        o_mode = "final" if gather == "frame" else "frame"
        eo, pipe_o, _ = timed_region(o_mode, False)
        same = None
        if rank == 0:
            fr, fi = pipe_o.frame_views()
            same = bool(torch.equal(fr, pipe.frame_views()[0]) and torch.equal(fi, pipe.frame_views()[1]))
        other = {"gather": o_mode, "value": round(W * H * args.steps / eo / 1e6, 2), "unit": "Mrays/s",
                 "ms_per_step": round(eo / args.steps * 1e3, 5), "same_pixels": same}

    # several GPUs, the weak-scaling figure beside the strong one: every GPU traces whole frames of its own (N viewers of one
    # scene: the tree is replicated anyway) and keeps them in its HBM; nothing crosses a link. The one-GPU configuration
    # (one stream, feedback scheduling, pre-roll) on every rank at once; every rank checks its frame against the golden hashes.
    replicas = None
    if world > 1 and not os.environ.get("VRT_BENCH_ONE_REGION"):
        ctx.set_tile_scheduling(16)
        er, pipe_r, _ = timed_region("final", False, PREROLL, whole=True)
        ctx.set_tile_scheduling(args.sched_period)
        fg = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))["frames"].get(
            GOLDEN_KEY[args.map] + ("_full" if mode == 2 else "") + f"/mode{mode}")
        ok = None
        if fg is not None and (fg["width"], fg["height"]) == (W, H):
            fr, fi = pipe_r.frame_views()
            ok = ("%016x" % V.fnv1a64(fr.cpu().numpy()) == fg["rgba_fnv1a64"] and
                  "%016x" % V.fnv1a64(fi.cpu().numpy()) == fg["id_dist_fnv1a64"])
        oks = [None] * world
        dist.all_gather_object(oks, ok)
        replicas = {"value": round(world * W * H * args.steps / er / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(er / args.steps * 1e3, 5),
                    "scaling": "weak", "frames_per_step": world,
                    "what": "every GPU traces a whole frame per step and keeps it resident (N viewers of one scene); no exchange",
                    "frames_match_oracle_golden": (all(oks) if all(o is not None for o in oks) else None)}
        del pipe_r

    # several GPUs: the same delivery with NO collective -- every rank's kernels store their tiles straight into the root's
    # frame through an IPC mapping (xGMI peer stores), ordered by stream flags (sharding.PeerFramePipeline): to rank 0,
    # and with the root rotating over the ranks (frame f assembled on rank f % N). A failure of the mappings is reported,
    # never fatal: the RCCL regions above stand on their own.
    peer = {}
    stuck_ranks = []    # ranks whose device-side wait never returned (gpu_hang)
    peer_stuck = False   # a device-side wait of the peer path never returned on some rank: the process ends through os._exit
    if (world > 1 or os.environ.get("VRT_BENCH_PEER_AT_ONE")) and not os.environ.get("VRT_BENCH_NO_PEER"):
        frames_g = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))["frames"]
        gkey = GOLDEN_KEY[args.map] + ("_full" if mode == 2 else "") + f"/mode{mode}"
        gg = frames_g.get(gkey)
        def agree(err):
            """the first error text any rank reports, the same answer on every rank (ranks must leave a region together)"""
            if world == 1:
                return err
            out = [None] * world
            dist.all_gather_object(out, err)
            return next((e for e in out if e), None)

        PEER_TIMEOUT_S = 60.0   # a flag hand-shake that never completes must not take the line with it
        for name, rotate in (("peer_store_rank0", False), ("peer_store_rotating_root", True), ("peer_store_rank0_colour_only", False)):
            colour_only = name.endswith("colour_only")   # reported, never the headline: 4 of the 12 bytes per pixel travel
            if peer_stuck:
                peer[name] = {"error": "skipped: an earlier peer region left a device-side wait blocked"}
                continue
            pp = None
            try:
                pp = shd.PeerFramePipeline(ctx, plan, n_buf=4, rotate=rotate, colour_only=colour_only)   # raises on every rank or on none
            except Exception as ex:  # noqa: BLE001 -- any failure of the IPC path is a report line, not the end of the bench
                peer[name] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
                continue
            try:
                why = pp.rehearse()                                              # the same text on every rank
                if why:
                    peer[name] = {"error": why}
                    continue

                def tiles(d_rgba, d_id, stream):
                    ctx.dispatch_tiles(W, H, args.tile_rows, rank, world, mode, d_rgba, d_id, stream)

                def run(n_frames):
                    err = None
                    try:
                        for _ in range(n_frames):
                            pp.step(tiles)
                        if not pp.drain(PEER_TIMEOUT_S):
                            err = f"rank {rank}: frames did not complete within {PEER_TIMEOUT_S} s"
                    except Exception as ex:  # noqa: BLE001
                        err = f"rank {rank}: {type(ex).__name__}: {ex}"
                    return agree(err)

                why = run(args.warmup)
                if why:
                    peer[name] = {"error": why[:300]}
                    continue
                if world > 1:
                    dist.barrier()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                why = run(args.steps)
                if why:
                    peer[name] = {"error": why[:300]}
                    continue
                if world > 1:
                    dist.barrier()
                torch.cuda.synchronize(dev)
                ep = time.perf_counter() - t0
                if world > 1:
                    t = torch.tensor([ep], dtype=torch.float64, device="cpu" if via_host else dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    ep = float(t.item())
                ok = None
                last = pp.last_frame()
                if last is not None and gg is not None and (gg["width"], gg["height"]) == (W, H):
                    ok = "%016x" % V.fnv1a64(last[0]) == gg["rgba_fnv1a64"]
                    if not (colour_only and world > 1):   # colour only: the root holds the (voxelID, dist) rows it traced itself
                        ok = ok and "%016x" % V.fnv1a64(last[1]) == gg["id_dist_fnv1a64"]
                oks = [ok]
                if world > 1:
                    oks = [None] * world
                    dist.all_gather_object(oks, ok)
                checked = [o for o in oks if o is not None]
                peer[name] = {"value": round(W * H * args.steps / ep / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(ep / args.steps * 1e3, 5),
                              "frames_match_oracle_golden": (all(checked) if checked else None), "roots_checked": len(checked),
                              "slots": pp.n_buf}
                if colour_only:
                    peer[name]["bytes_per_pixel_to_root"] = 4
            except Exception as ex:  # noqa: BLE001
                peer[name] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
            finally:
                stuck = [pp.stuck]
                if world > 1:
                    stuck = [None] * world
                    try:
                        dist.all_gather_object(stuck, pp.stuck)
                    except Exception:  # noqa: BLE001
                        stuck = [True]
                peer_stuck = peer_stuck or any(stuck)
                stuck_ranks = sorted(set(stuck_ranks) | {i for i, v in enumerate(stuck) if v})
                try:
                    pp.close()
                except Exception:  # noqa: BLE001
                    pass

    # The COMPLETE frame of the reference's loop -- full path tracer, then the display pass -- sharded by row bands with a
    # 20-row halo, only the displayed image (4 B/pixel) delivered to rank 0 (sharding.ShownFramePipeline). Reported, never
    # the headline (the metric is primary rays); at one GPU only with --extras.
    shown = None
    peer_failed = any("error" in v for v in peer.values())   # the same IPC mappings and stream flags: do not try them again
    if (world > 1 or args.extras) and not peer_stuck and not peer_failed and not os.environ.get("VRT_BENCH_NO_SHOWN"):
        sp = None
        try:
            sp = shd.ShownFramePipeline(ctx, W, H, rank, world, 2, n_buf=3)
        except Exception as ex:  # noqa: BLE001
            shown = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        if sp is not None:
            try:
                def agree2(err):
                    if world == 1:
                        return err
                    out = [None] * world
                    dist.all_gather_object(out, err)
                    return next((e for e in out if e), None)

                def run2(n_frames):
                    err = None
                    try:
                        for _ in range(n_frames):
                            sp.step()
                        if not sp.drain(60.0):
                            err = f"rank {rank}: frames did not complete within 60 s"
                    except Exception as ex:  # noqa: BLE001
                        err = f"rank {rank}: {type(ex).__name__}: {ex}"
                    return agree2(err)

                why = run2(max(2, args.warmup))
                if not why:
                    if world > 1:
                        dist.barrier()
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    why = run2(args.steps)
                    if world > 1:
                        dist.barrier()
                    es = time.perf_counter() - t0
                if why:
                    shown = {"error": why[:300]}
                else:
                    if world > 1:
                        t = torch.tensor([es], dtype=torch.float64, device="cpu" if via_host else dev)
                        dist.all_reduce(t, op=dist.ReduceOp.MAX)
                        es = float(t.item())
                    same = golden_ok = None
                    if rank == 0:   # against the one-GPU route (whole frame traced, then filtered) and the oracle's committed hash
                        got = sp.last_shown()
                        ref_rgba, ref_id = ctx.dispatch(W, H, 2)
                        same = bool((ctx.denoise(ref_rgba, ref_id) == got).all())
                        gs = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))["frames"].get(
                            GOLDEN_KEY[args.map] + "_full/mode2", {})
                        if gs.get("shown_fnv1a64") and (gs.get("width"), gs.get("height")) == (W, H):
                            golden_ok = "%016x" % V.fnv1a64(got) == gs["shown_fnv1a64"]
                    shown = {"frames_per_s": round(args.steps / es, 1), "ms_per_frame": round(es / args.steps * 1e3, 5),
                             "what": "full path tracer + display pass per frame, row bands with a 20-row halo, the displayed image "
                                     "(4 B/pixel) delivered to rank 0 through IPC mappings", "rows_traced_per_rank": sp.h1 - sp.h0,
                             "rows_shown_per_rank": sp.b1 - sp.b0, "same_pixels_as_one_gpu": same,
                             "matches_oracle_golden": golden_ok}
            except Exception as ex:  # noqa: BLE001
                shown = {"error": f"{type(ex).__name__}: {ex}"[:300]}
            finally:
                stuck = [sp.stuck]
                if world > 1:
                    stuck = [None] * world
                    try:
                        dist.all_gather_object(stuck, sp.stuck)
                    except Exception:  # noqa: BLE001
                        stuck = [True]
                peer_stuck = peer_stuck or any(stuck)
                stuck_ranks = sorted(set(stuck_ranks) | {i for i, v in enumerate(stuck) if v})
                try:
                    sp.close()
                except Exception:  # noqa: BLE001
                    pass

    # one GPU, informational: the same K frames rotating through four streams (no per-launch events; the figure the
    # headline would become if overlapped launches were allowed to blur the per-kernel duration the roofline uses)
    overlapped = None
    if args.extras and world == 1 and n_streams == 1:
        pipe2 = shd.FramePipeline(plan, dev, gather=gather, streams=4)
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
        import numpy as np
        frames = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))["frames"]
        key = GOLDEN_KEY[args.map] + ("_full" if mode == 2 else "") + f"/mode{mode}"
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
        # several ranks: two ways deliver every frame to rank 0 -- the RCCL gather timed above and the peer stores; the
        # headline is the faster one whose frames were verified, the other stays beside it
        delivery_used = None
        rccl_frame = None
        if world > 1 and gather == "frame":
            rccl_frame = {"value": round(value, 2), "unit": "Mrays/s", "ms_per_step": round(ms_per_step, 5)}
            delivery_used = "rccl_gather"
        # the headline stays on the region its per-launch figures (kernel_ms, roofline, issue_roofline) were measured in -- the RCCL
        # gather --; the peer-store regions are reported beside it, and the faster verified delivery is NAMED, not swapped in
        best_delivery = None
        if world > 1 and gather == "frame":
            best_delivery = {"name": "rccl_gather", "value": round(value, 2), "ms_per_step": round(ms_per_step, 5)}
            for nm in ("peer_store_rank0",):
                pd = peer.get(nm) or {}
                if pd.get("frames_match_oracle_golden") and pd.get("value", 0) > best_delivery["value"]:
                    best_delivery = {"name": nm, "value": pd["value"], "ms_per_step": pd["ms_per_step"]}
        roofline = None
        issue = None
        if known and len(kernel_ms):
            # "reference-requested bytes": 4 B per texel fetch the REFERENCE algorithm issues for the rows this rank traces
            # + 12 B per pixel written; exact per-row counts are committed. These fetches are served
            # from cache (the tree is L1/L2 resident), so this figure may pass the HBM peak: it is the contract's
            # metric, not the kernel's limiter -- that one is `issue_roofline`.
            if "row_fetches" in g:
                f_local = sum(g["row_fetches"][r] for r in plan.rows_of[0])
            else:
                f_local = g["fetches"] * plan.rows_local // H
            b_algo = 4 * f_local + 12 * W * plan.rows_local
            wkey = f"{args.map}/{W}x{H}/{args.mode}/variant{args.variant}"
            roofline = roofline_block(V, wkey, b_algo, stream_bytes + 12 * W * plan.rows_local, kernel_ms, world)
            issue = issue_roofline(V, wkey, roofline["kernel_avg_ms"], plan.rows_local / H)   # a shard issues its rows' share
        if batched and roofline and batched["launch_avg_ms"]:
            b4 = batched["frames_per_launch"] * roofline["algorithmic_bytes_per_launch"]
            batched["roofline_frac"] = round(b4 / (batched["launch_avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        # the other BASELINE configurations (one GPU): after everything that needs the headline scene on the device
        configs = None
        if world == 1 and not args.no_configs:
            configs = time_configs(args, V, ctx, torch, dev, {args.map: wld})
            try:
                configs.append(config1_cpu_cast(V))
            except Exception as ex:  # noqa: BLE001
                configs.append({"name": "config1_custom_vox_256x256_cpu_octree_ray_cast", "error": f"{type(ex).__name__}: {ex}"[:300]})
        delivery = {"final": "one rank: every frame is complete where it was traced (rgba8 image, then (voxelID, dist) image, in HBM); "
                             "nothing to assemble" if world == 1 else
                             "frames stay sharded in HBM, no collective per step; the last frame is gathered to rank 0 and "
                             "assembled inside the timed region",
                    "frame": "every frame gathered to rank 0 inside the timed region, double-buffered (gather of frame i "
                             "overlaps trace of frame i+1)"}[gather]
        out = {
            "metric": metric_name(),
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32",
            "data": (f"tests/golden/maps/{args.map}.vox scene fixture" if args.map != "terrain" else
                     "FastNoiseLite(1337) Perlin height field fixture (tests/golden/terrain.json), reference terrain generator") +
                    ", fixed synthetic camera pose",
            "config": {"workload": (f"{args.map}.vox" if args.map != "terrain" else "terrain height field (BASELINE config 4)") +
                                   f" {W}x{H} {args.mode} rays, pose {pose}", "mode": args.mode,
                       "sharding": f"interleaved {args.tile_rows}-row tiles over {world} rank(s); " + (
                           delivery),
                       "gather": gather, "delivery": delivery_used, "streams": n_streams, "tile_scheduling_period": args.sched_period,
                       "variant": args.variant, "ray_tables": not args.no_ray_tables, "preroll_launches": PREROLL, "collective_backend": args.backend if world > 1 else None,
                       "launcher": launcher},
            "roofline": roofline,
            "issue_roofline": issue,
            "pixels_match_oracle_golden": check,
            "cold_start": cold,
            ("sharded_resident" if gather == "frame" else "every_frame_delivered"): other,
            "rccl_gather_every_frame": rccl_frame,
            "best_verified_delivery_to_rank0": best_delivery,
            "whole_frame_per_gpu": replicas,
            "peer_delivery": peer or None,
            "shown_frame_pipeline": shown,
            "overlapped_frames": overlapped,
            "batched_views": batched,
            "configs": configs,
        }
        if world > 1:
            # which figure answers BASELINE's ">= 6x at 8 GPUs": the headline delivers 12 B/pixel of every primary-ray frame into ONE GPU
            # and is bound by that GPU's xGMI links from N = 2; the pipelines that shard without funnelling are
            # `shown_frame_pipeline` (the frame the reference displays: 4 B/pixel delivered, both kernels sharded) and
            # `whole_frame_per_gpu` (weak scaling, nothing on a link)
            out["scaling_target_answered_by"] = "shown_frame_pipeline"
            out["ranks_observed"] = rank_census
        if peer_stuck:
            out["gpu_hang"] = {"flag": "peer_stuck", "ranks": stuck_ranks,
                               "what": "a device-side wait of the peer-store / shown-frame path never returned on these ranks; the process "
                                       "leaves through os._exit(3) without synchronising the blocked streams"}
        if world == 1 and not args.no_cpu_baseline:
            if args.map == "terrain_full":   # beyond the texel stream the port reads (its wide-pointer reader is a test-only extension)
                out["cpu_baseline"] = None
            else:
                tex, tdim = wld.flatten()
                out["cpu_baseline"] = cpu_baseline(args, tex, tdim, (ip, iv, cp))
        print(json.dumps(out), flush=True)
    if peer_stuck:      # streams that will never drain: no barrier, no teardown that would wait for them; a GPU hang is NOT rc 0
        sys.stdout.flush()
        sys.stderr.write(f"bench.py: rank {rank}: gpu_hang (peer_stuck): leaving with exit code 3\n")
        sys.stderr.flush()
        os._exit(3)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
