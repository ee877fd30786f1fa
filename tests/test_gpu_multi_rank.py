"""Multi-rank runs of the Z-slab pipeline on the GPU box: real HIP operators and composite / hand-over kernels under
real collectives.  The box has one GPU, so the ranks share it and the transport is gloo (RCCL refuses two ranks on one
device); everything else is the code path `bench.py --gpus N` takes.  See tests/mp_slab_gpu.py."""
import os
import socket
import subprocess
import sys

import pytest

import kfx_testlib as T

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,halo,raycast,mode", [(2, "recompute", "composite", ""), (2, "exchange", "exact", ""), (3, "exchange", "composite", ""),
                                                     (3, "recompute", "exact", ""), (2, "recompute", "exact", "tracking"),
                                                     (2, "exchange", "composite", "tracking"),
                                                     # the frame as ONE kfx_slab_frame_step call per rank (collectives through torch.distributed)
                                                     (2, "recompute", "exact", "cframe"), (3, "exchange", "exact", "cframe"),
                                                     (2, "recompute", "composite", "cframe"), (3, "exchange", "composite", "cframe")])
def test_gpu_slab_pipeline_ranks_sharing_one_gpu(world, halo, raycast, mode):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(T.ROOT, "tests", "mp_slab_gpu.py"), halo, raycast] + ([mode] if mode else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=T.ROOT)
    assert out.returncode == 0 and out.stdout.count("MP_OK") == world, out.stdout[-3000:] + out.stderr[-3000:]


def _bench(extra, env_extra=None, launcher=True, timeout=900):
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["KFX_BENCH_BACKEND"] = "gloo"
    env.update(env_extra or {})
    base = [os.path.join(T.ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--res", "128", "--no-cpu-baseline", "--prime", "6"]
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + base + list(extra)
    else:
        cmd = [sys.executable] + base + list(extra)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=T.ROOT, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, (json.loads(lines[0]) if len(lines) == 1 else None)


@pytest.mark.parametrize("raycast", ["exact", "composite"])
def test_gpu_bench_two_ranks_smoke(raycast):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per process), with the transport
    switched to gloo because both ranks share the box's single GPU: one JSON line, whole-job value, ranks agree.  The default
    driver: one kfx_slab_frame_step call per frame and rank (here with its collectives routed through the process group)."""
    out, d = _bench(["--raycast", raycast])
    assert out.returncode == 0 and d is not None, out.stdout[-2000:] + out.stderr[-3000:]
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["ranks_agree"] is True and d["roofline"]["bound"] == "hbm" and "cpu_baseline" not in d
    assert d["config"]["driver"].startswith("c:") and d["config"]["raycast_mode"] == raycast
    assert ("bit-identical" in d["config"]["raycast_parity"]) == (raycast == "exact")
    # the N = 1 point with the same kernels, measured in the same run: what a strong-scaling efficiency is computed against
    assert d["strong_scaling_baseline_fps"] > 0 and "kfx_raycast_sdf" in d["strong_scaling_baseline"]["note"]
    # what the first run on real links has to show without a second attempt (round-3 verdict item 8)
    comm = d["config"]["communicator"]
    assert comm["n_ranks"] == 2 and comm["backend"] == "gloo" and "rccl_version" in comm and "kfx_slab_frame_step" in comm["frame_collectives"]
    pr = d["per_rank"]
    assert isinstance(pr, list) and [r["rank"] for r in pr] == [0, 1]
    for r in pr:
        assert r["sdf_fuse_ms"] > 0 and r["raycast_sdf_plus_merge_ms"] > 0 and r["halo_exchange_ms"] > 0 and r["preprocess_ms"] > 0 and r["frame_events_ms"] > 0
        # ghost planes per side: 2, or -- exact raycast, recomputed ghosts (the default) -- kfx_slab_exact_ghost's width: 7 here
        g = d["config"]["ghost_planes"]
        assert g == (7 if raycast == "exact" else 2)
        assert r["halo_bytes_received_per_fuse"] == g * 128 * 128 * 8          # the ghost planes of 128 x 128 cells from the one neighbour
        assert r["planes_owned"] == 64 and r["voxels_stored"] == 128 * 128 * (64 + g)
        assert (r["composite_merge_ms"] is not None and r["composite_merge_ms"] > 0) == (raycast == "composite")
    v = d["multi_gpu_variants"]
    assert ("raycast_exact_pipelined_with_last_stage_fps" in v) == (raycast == "exact")
    for key in ("as_configured_fps", "raycast_exact_fps", "raycast_exact_pipelined_fps", "halo_exchange+raycast_exact_pipelined_fps",
                "raycast_exact_pipelined_tiles_1_fps", "halo_exchange+raycast_exact_fps", "raycast_exact_tiles_1_fps", "raycast_exact_tiles_8_fps",
                "raycast_composite_fps", "raycast_composite_overlapped_fps", "raycast_composite_merge_allreduce_fps", "halo_exchange_fps",
                "inputs_broadcast_fps", "driver_python_fps"):
        assert v.get(key, 0) > 0, (key, v)
    if raycast == "composite":   # direct-send merge, overlapped with the next frame where nothing else communicates
        assert "all_to_all" in d["config"]["partition"] and "overlapped" in d["config"]["partition"]
    else:   # the exact default: frames pipelined across the ranks (the final exchange on the side stream / second communicator)
        assert "tokens over image row-tiles" in d["config"]["partition"] and "overlapped" not in d["config"]["partition"]
        assert d["config"]["frames_pipelined"] is True and d["config"]["pipeline_note"] is None and "frames pipelined" in d["config"]["partition"]
    assert d["kernels_ms"]["sdf_fuse"] == pr[0]["sdf_fuse_ms"] and d["kernels_ms"]["host_gap"] is not None


def test_gpu_bench_unpipelined_exact_two_ranks():
    """--no-overlap (or KFX_BENCH_PIPELINE=0): the exact hand-over with its final exchange on the launch stream, frame by frame."""
    out, d = _bench(["--raycast", "exact", "--no-overlap"])
    assert out.returncode == 0 and d is not None, out.stdout[-2000:] + out.stderr[-3000:]
    assert d["config"]["frames_pipelined"] is False and "frames pipelined" not in d["config"]["partition"] and d["config"]["ranks_agree"] is True
    assert d["multi_gpu_variants"]["raycast_exact_pipelined_fps"] > 0


def test_gpu_bench_python_driver_two_ranks():
    """--driver python: SlabPipeline issues operators and torch.distributed collectives one by one (the cross-check of the C call)."""
    out, d = _bench(["--driver", "python", "--raycast", "composite"])
    assert out.returncode == 0 and d is not None, out.stdout[-2000:] + out.stderr[-3000:]
    assert d["config"]["driver"].startswith("python") and d["config"]["ranks_agree"] is True
    pr = d["per_rank"]
    for r in pr:
        assert r["composite_merge_ms"] > 0 and r["composite_merge_allreduce_ms"] > 0   # both merges, by themselves
    v = d["multi_gpu_variants"]
    assert v["merge_allreduce_fps"] > 0 and v["overlap_off_fps"] > 0 and v["halo_exchange_fps"] > 0 and v["images_root_fps"] > 0


def test_gpu_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts the two ranks itself (child processes of
    a parent that never touches the GPU) and the line says n_gpus = 2.  gloo, because the box has a single GPU."""
    out, d = _bench(["--raycast", "composite", "--no-overlap"], launcher=False)
    assert out.returncode == 0 and d is not None, out.stdout[-2000:] + out.stderr[-3000:]
    assert d["n_gpus"] == 2 and d["config"]["ranks_agree"] is True and "overlapped" not in d["config"]["partition"]
    v = d["multi_gpu_variants"]
    assert v["as_configured_fps"] > 0 and v["halo_exchange_fps"] > 0 and v["raycast_composite_overlapped_fps"] > 0


def test_gpu_bench_overlapped_merge_two_ranks():
    """--overlap: frame k's composite runs on the frame's side stream under frame k+1's SdfFuse; ranks still agree on the images."""
    out, d = _bench(["--raycast", "composite", "--overlap", "--steps", "6"], launcher=False)
    assert out.returncode == 0 and d is not None, out.stdout[-2000:] + out.stderr[-3000:]
    assert d["n_gpus"] == 2 and d["config"]["ranks_agree"] is True and "overlapped" in d["config"]["partition"]
    # the overlapped merge next to the ghost-plane exchange would interleave collectives in rank-dependent order: refused
    bad, _ = _bench(["--raycast", "composite", "--overlap", "--halo", "exchange"], launcher=False, timeout=300)
    assert bad.returncode != 0 and "--overlap needs" in bad.stderr + bad.stdout


def test_gpu_bench_refuses_more_ranks_than_gpus():
    """With the RCCL backend every rank needs its own GPU: on this 1-GPU box `--gpus 2` must fail loudly, not report
    a 1-GPU number."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KFX_BENCH_BACKEND")}
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    cmd = [sys.executable, os.path.join(T.ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--res", "64", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=T.ROOT, env=env)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")], out.stdout[-2000:]
    assert "needs 2 GPUs" in out.stderr + out.stdout
