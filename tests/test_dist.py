"""The N>1 path on CPU: two gloo ranks shard a frame by interleaved row tiles and gather it to rank 0.

The device kernel cannot run here, so each rank fills its compact shard buffer from the oracle's rows
(test infrastructure standing in for the GPU); what is under test is the product's sharding plan and
the gather/un-interleave code that bench.py runs over RCCL."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys, importlib
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
    import vrt_import, oracle_py as O
    V = vrt_import.vrt()
    shd = importlib.import_module("voxel-raytracer_amd.sharding")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = V.World(); assert w.load_vox(os.path.join({root!r}, "tests/golden/maps/monu9.vox"))
    tex, dim = w.flatten()
    for (W, H, tile_rows) in [(96, 54, 8), (64, 37, 5), (32, 8, 8)]:
        ip, iv, cp, _ = V.camera_block((48.5, 60.5, 170.5), -90.0, -12.0, W, H)
        s = O.make_scene(tex, dim, ip, iv, cp)
        full_rgba, full_id, _, _ = O.render(s, W, H, 1)
        plan = shd.ShardPlan(W, H, tile_rows, rank, world)
        assert plan.rows_of[rank] == V.shard_row_indices(H, tile_rows, rank, world)
        assert plan.rows_local == V.shard_rows(H, tile_rows, rank, world)
        assert sorted(r for rows in plan.rows_of for r in rows) == list(range(H))
        local = plan.local_buffer("cpu")
        rows = plan.rows_of[rank]
        if rows:
            rm = plan.rows_max
            local[: rm * W].view(rm, W)[: len(rows)] = torch.from_numpy(full_rgba[rows].copy().view(np.int32).reshape(len(rows), W))
            local[rm * W:].view(rm, W, 2)[: len(rows)] = torch.from_numpy(full_id[rows].copy())
        gathered = plan.gather_buffer("cpu") if rank == 0 else None
        store = plan.frame_store("cpu") if rank == 0 else None
        idx = plan.scatter_index("cpu") if rank == 0 else None
        shd.gather_frame(plan, local, gathered, store, idx)
        if rank == 0:
            frame_rgba, frame_id = plan.frame_views(store)
            assert np.array_equal(frame_rgba.numpy().view(np.uint8).reshape(H, W, 4), full_rgba)
            assert np.array_equal(frame_id.numpy(), full_id)
        # the pipelines bench.py uses: frames through two to four buffers (one per stream); "frame" gathers each
        # one asynchronously, "final" keeps them sharded and gathers the last one on drain
        for mode, n_streams, n_frames in (("frame", 1, 5), ("final", 1, 5), ("frame", 4, 7), ("final", 4, 7), ("final", 3, 2)):
            pipe = shd.FramePipeline(plan, "cpu", gather=mode, streams=n_streams)
            assert pipe.n_buf == max(2, n_streams)
            for f in range(n_frames):
                k, _, _ = pipe.slot()
                buf = pipe.local[k]
                buf.zero_()
                if rows:
                    rm = plan.rows_max
                    buf[: rm * W].view(rm, W)[: len(rows)] = torch.from_numpy(full_rgba[rows].copy().view(np.int32).reshape(len(rows), W)) + f
                    buf[rm * W:].view(rm, W, 2)[: len(rows)] = torch.from_numpy(full_id[rows].copy()) - f
                pipe.submit(k)
                if f == 1:
                    pipe.drain()          # a fence in mid-stream (bench.py has one after warm-up)
            pipe.drain()
            pipe.drain()                  # idempotent
            if rank == 0:
                fr, fi = pipe.frame_views()
                assert np.array_equal(fr.numpy(), full_rgba.view(np.int32).reshape(H, W) + (n_frames - 1)), (mode, n_streams)
                assert np.array_equal(fi.numpy(), full_id - (n_frames - 1)), (mode, n_streams)
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.write("rank %d ok\\n" % rank); sys.stdout.flush()
""")


def test_two_rank_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` outside torchrun must become a two-rank job by itself (it printed an
    n_gpus: 1 line): the plan-only mode runs the launcher, the rendezvous and the shard plan without a GPU."""
    import json
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-plan"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["launcher"] == "self-spawned" and d["collective_backend"] == "gloo"
    assert d["gather"] == "frame" and d["rows_per_rank"] == [544, 536] and d["rows_sum_checked_across_ranks"] is True
    # one rank: no launcher, no collective
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-plan"], env=env, capture_output=True, text=True, timeout=300)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["launcher"] == "single" and d["gather"] == "final" and d["rows_per_rank"] == [1080]
    # started under a launcher whose world size disagrees with --gpus: refuse instead of printing a line for another job
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-plan"],
                       env=dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
