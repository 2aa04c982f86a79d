"""The N > 1 path on real devices. The build box and the round-end test box have ONE GPU: what needs two skips there and
enables itself wherever torch.cuda.device_count() >= 2 (the scaling node), so that the first multi-GPU box that runs the
suite also checks pixels across real xGMI links -- peer mappings between distinct devices, IPC mappings across processes on
different devices, RCCL with more than one rank -- before any number is taken there."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

import torch

N_DEV = torch.cuda.device_count()   # does not initialise the GPU
needs2 = pytest.mark.skipif(N_DEV < 2, reason=f"needs two GPUs; this box has {N_DEV}")


def _env():
    return dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")


def test_rccl_collectives_of_the_frame_pipeline_world_size_one():
    """bench.py's N > 1 code path through the real backend on whatever this box has: torch.distributed "nccl" (= RCCL) with one rank,
    the frame pipeline told to run its collectives anyway -- asynchronous gather into device tensors, barrier, all_reduce of a
    device double, side streams -- frames checked against the golden hashes (tools/rccl_rehearsal.py, a child process)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_rehearsal.py")], env=dict(_env(), VRT_REHEARSAL_STEPS="8"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl rehearsal ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


@needs2
def test_vrt_multi_on_two_devices(V, golden, product_scenes):
    """vrt_create_multi([0, 1]): the frame assembled on device 0 by device 1's kernel stores through a peer mapping
    (hipDeviceEnablePeerAccess between DISTINCT devices) and by the pull kernel, against the oracle's hashes; the displayed frame in
    two row bands with halos against the one-context route."""
    tex, dim = product_scenes["dragon"]
    for key in ("dragon_256x144/mode1", "dragon_256x144/mode2", "dragon_1080p/mode0", "dragon_1080p_full/mode2"):
        g = golden["frames"]["frames"][key]
        W, H = g["width"], g["height"]
        ip, iv, cp, _ = V.camera_block(g["pose"][:3], g["pose"][3], g["pose"][4], W, H)
        for devices in ([0, 1], [1, 0], list(range(min(N_DEV, 4)))):
            m = V.Multi(devices)
            try:
                m.upload_octree(tex, dim)
                m.set_camera(ip, iv, cp)
                d_rgba, d_id = m.frame_alloc(W, H)
                c0 = m.context(0)
                for delivery in (V.DELIVER_PEER_STORE, V.DELIVER_GATHER):
                    c0.device_write(d_rgba, np.zeros(W * H, np.uint32), m.stream())
                    for _ in range(3):
                        m.dispatch(W, H, 8, g["mode"], delivery, d_rgba, d_id)
                    rgba = c0.device_read(d_rgba, (H, W, 4), np.uint8, m.stream())
                    idd = c0.device_read(d_id, (H, W, 2), np.int32, m.stream())
                    assert "%016x" % V.fnv1a64(rgba) == g["rgba_fnv1a64"], (key, devices, delivery)
                    assert "%016x" % V.fnv1a64(idd) == g["id_dist_fnv1a64"], (key, devices, delivery)
                if "shown_fnv1a64" in g:
                    for _ in range(3):
                        m.dispatch_frame(W, H, 2, d_rgba)
                    shown = c0.device_read(d_rgba, (H, W, 4), np.uint8, m.stream())
                    assert "%016x" % V.fnv1a64(shown) == g["shown_fnv1a64"], (key, devices, "displayed frame in row bands")
                m.synchronize()
                m.frame_free(d_rgba, d_id)
            finally:
                m.close()


@needs2
@pytest.mark.parametrize("n", [2] + ([4] if N_DEV >= 4 else []))
def test_bench_ranks_on_distinct_devices(n):
    """bench.py --gpus n as the driver runs it (it starts its own ranks): one process per GPU over RCCL, every frame gathered to
    rank 0; the peer-store pipelines through IPC mappings across devices (rank 0 root and rotating roots); the displayed frame in
    row bands. Every region's frames must reproduce the oracle's hashes, the ranks must sit on distinct devices, and the line
    must not report a hang."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "8", "--warmup", "3"], env=_env(),
                       capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(lines[-1])
    assert d["n_gpus"] == n and d["pixels_match_oracle_golden"] is True
    assert d["ranks_observed"]["ranks"] == n and d["ranks_observed"]["distinct_devices"] == n, d["ranks_observed"]
    assert "gpu_hang" not in d
    assert d["whole_frame_per_gpu"]["frames_match_oracle_golden"] is True
    for name, p in (d.get("peer_delivery") or {}).items():
        assert "error" not in p, (name, p)
        assert p["frames_match_oracle_golden"] is True, (name, p)
    s = d["shown_frame_pipeline"]
    assert s and "error" not in s and s["matches_oracle_golden"] is True and s["same_pixels_as_one_gpu"] is True, s
    assert d["scaling_target_answered_by"] == "shown_frame_pipeline"
