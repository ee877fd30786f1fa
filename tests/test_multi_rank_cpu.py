"""World-size-2 (and 3) gloo tests of the Z-slab pipeline on CPU: the host-side partitioning and
compositing logic of kangaroo_amd/pipeline.py, with the oracle standing in for the HIP operators
(tests/oracle_ops.py).  The GPU box runs the same code over RCCL with kangaroo_amd.roo."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import kfx_testlib as T
from kangaroo_amd import scenes
from kangaroo_amd.pipeline import FramePipeline, SlabPipeline, slab_range

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

N, W, H, FRAMES = 48, 96, 72, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, halo, raycast="composite", inputs="replicate", images="all", merge="direct"):
    import oracle_ops as ops
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bmin, bmax, near, far = scenes.SCENES["room"]
    pipe = SlabPipeline(ops, dist, (N, N, N), bmin, bmax, W, H, halo=halo, raycast=raycast, near=near, far=far, inputs=inputs, images=images,
                        merge=merge)
    K = pipe.K
    for i in range(FRAMES):
        T_wc = scenes.orbit_pose(i, 8)
        raw = scenes.render_depth("room", W, H, T_wc, K)
        if inputs == "broadcast" and rank != 0:   # only rank 0 sees the sensor: the others must get the maps from it
            raw = np.full_like(raw, np.nan)
        pipe.raw.MemcpyFromHost(raw)
        pipe.step(T_wc)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), rounds=getattr(pipe, "rounds", 0), vol=pipe.vol.data, s0=pipe.s0, s1=pipe.s1, z0=pipe.z0, z1=pipe.z1,
             depth=pipe.ray_d.data, norm=pipe.ray_n.data, img=pipe.ray_i.data)
    dist.barrier()
    dist.destroy_process_group()


def test_slab_range_partitions_every_plane_once():
    for d in (8, 48, 100, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [slab_range(d, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == d
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_slab_extent_reproduces_the_reference_extents_of_the_whole_volume():
    """Dimensions that are not multiples of 8: roo::SdfFuse skips the trailing dim % 8 voxels per axis (quirk Q1).  Slabs
    integrated with full_extent="slab" leave exactly those voxels untouched, whatever the partition."""
    import oracle
    dims, w, h = (20, 17, 29), 80, 60
    bmin, bmax = scenes.SCENES["full"][:2]
    K = scenes.intrinsics(w, h)
    f, vbo, nrm = T.preprocess_oracle(scenes.render_depth("full", w, h, None, K), K)
    tr = scenes.trunc_dist(bmin, bmax, dims)
    whole = oracle.Volume(*dims, bmin, bmax)
    oracle.sdf_reset(whole, float("nan"))
    n_ref = oracle.sdf_fuse(whole, f, nrm, scenes.identity_pose(), K, tr, 1000.0, 0.1)
    assert n_ref == 16 * 16 * 24
    for world in (2, 3, 5):
        parts = oracle.Volume(*dims, bmin, bmax)
        oracle.sdf_reset(parts, float("nan"))
        n = 0
        for r in range(world):
            z0, z1 = slab_range(dims[2], r, world)
            sub = oracle.KfoVolume()
            sub.pitch, sub.img_pitch, sub.w, sub.h, sub.d = parts.pitch, parts.img_pitch, dims[0], dims[1], z1 - z0
            sub.ptr = parts.raw.ctypes.data + z0 * parts.img_pitch
            for k in range(3):
                sub.boxmin[k], sub.boxmax[k] = bmin[k], bmax[k]
            view = oracle.SubVolume(parts, sub)
            n += oracle.sdf_fuse(view, f, nrm, scenes.identity_pose(), K, tr, 1000.0, 0.1, full_extent="slab",
                                 slab=(dims[2], z0, bmin[2], bmax[2]))
        assert n == n_ref and T.nan_equal(parts.data, whole.data)


@pytest.mark.parametrize("world,halo,inputs", [(2, "exchange", "replicate"), (2, "recompute", "replicate"), (3, "exchange", "replicate"),
                                              (2, "exchange", "broadcast")])
def test_slab_pipeline_matches_single_volume(tmp_path, world, halo, inputs):
    import oracle_ops as ops
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), halo, "composite", inputs), nprocs=world, join=True)
    # single-process reference: the same frames through the monolithic pipeline
    bmin, bmax, near, far = scenes.SCENES["room"]
    ref = FramePipeline(ops, (N, N, N), bmin, bmax, W, H, near=near, far=far)
    for i in range(FRAMES):
        T_wc = scenes.orbit_pose(i, 8)
        ref.raw.MemcpyFromHost(scenes.render_depth("room", W, H, T_wc, ref.K))
        ref.preprocess()
        # the monolithic reference with roo::SdfFuse's own extents: the slabs integrate exactly those voxels (full_extent="slab")
        ops.SdfFuse(ref.vol, ref.filtered, ref.normals, scenes.se3_inverse(T_wc), ref.K, ref.trunc, ref.max_w, ref.mincostheta)
        ref.raycast(T_wc)
    full = ref.vol.data
    ranks = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]

    # (1) fused slabs: every stored plane (owned + ghosts) is BIT-IDENTICAL to the same plane of the
    #     monolithic volume (the slab entry point evaluates voxel positions with the full volume's expression)
    covered = np.zeros(N, bool)
    for r in ranks:
        s0, s1, z0, z1 = int(r["s0"]), int(r["s1"]), int(r["z0"]), int(r["z1"])
        assert s0 <= z0 < z1 <= s1 and r["vol"].shape[0] == s1 - s0
        covered[z0:z1] = True
        assert T.nan_equal(r["vol"], full[s0:s1]), T.mismatch_report(r["vol"], full[s0:s1])
    assert covered.all()

    # (2) every rank holds the same composite image
    for r in ranks[1:]:
        assert T.nan_equal(r["depth"], ranks[0]["depth"]) and T.nan_equal(r["norm"], ranks[0]["norm"])
        assert T.nan_equal(r["img"], ranks[0]["img"])

    # (3) the composite agrees with the single-volume raycast: same hit mask up to a few boundary pixels,
    #     depth within a fraction of a voxel (the march restarts at each slab entry)
    d_ref, d_got = ref.ray_d.data, ranks[0]["depth"]
    hit_ref, hit_got = np.isfinite(d_ref), np.isfinite(d_got)
    assert (hit_ref != hit_got).mean() < 0.01
    both = hit_ref & hit_got
    voxel = (bmax[0] - bmin[0]) / (N - 1)
    err = np.abs(d_ref[both] - d_got[both])
    assert np.median(err) < 0.02 * voxel and np.quantile(err, 0.99) < 0.5 * voxel
    n_ref, n_got = ref.ray_n.data[both], ranks[0]["norm"][both]
    assert (np.abs(n_ref - n_got).max(axis=1) < 0.05).mean() > 0.98
    assert (ranks[0]["norm"][~hit_got] == 0).all() and (ranks[0]["img"][~hit_got] == 0).all()


@pytest.mark.parametrize("world,mode", [(2, "exact"), (3, "exact"), (4, "exact"), (3, "exact_allreduce")])
def test_exact_slab_raycast_is_bit_identical_to_single_volume(tmp_path, world, mode):
    """raycast="exact": the march state travels with the ray across the slabs (neighbour exchanges, world + 1 stages, no
    host check in between), so depth, normals and shade equal the single-volume RaycastSdf bit for bit on every rank;
    "exact_allreduce" is its cross-check (all-reduce + termination test per round)."""
    import oracle_ops as ops
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "exchange", mode), nprocs=world, join=True)
    bmin, bmax, near, far = scenes.SCENES["room"]
    ref = FramePipeline(ops, (N, N, N), bmin, bmax, W, H, near=near, far=far)
    for i in range(FRAMES):
        T_wc = scenes.orbit_pose(i, 8)
        ref.raw.MemcpyFromHost(scenes.render_depth("room", W, H, T_wc, ref.K))
        ref.preprocess()
        ops.SdfFuse(ref.vol, ref.filtered, ref.normals, scenes.se3_inverse(T_wc), ref.K, ref.trunc, ref.max_w, ref.mincostheta)
        ref.raycast(T_wc)
    assert np.isfinite(ref.ray_d.data).mean() > 0.3
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert int(got["rounds"]) == world + 1 if mode == "exact" else 1 < int(got["rounds"]) <= world + 3
        assert T.nan_equal(got["depth"], ref.ray_d.data), T.mismatch_report(got["depth"], ref.ray_d.data)
        assert T.nan_equal(got["norm"], ref.ray_n.data), T.mismatch_report(got["norm"], ref.ray_n.data)
        assert T.nan_equal(got["img"], ref.ray_i.data), T.mismatch_report(got["img"], ref.ray_i.data)


def _tracking_worker(rank, world, port, out_dir, raycast, images="all"):
    import oracle_ops as ops
    from kangaroo_amd.pipeline import TrackingSlabPipeline
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bmin, bmax, near, far = scenes.SCENES["room"]
    pipe = TrackingSlabPipeline(ops, dist, (N, N, N), bmin, bmax, 160, 120, halo="exchange", raycast=raycast, near=near, far=far, images=images)
    poses = []
    for i in range(4):
        T_true = scenes.orbit_pose(i, 60)
        pipe.raw.MemcpyFromHost(scenes.render_depth("room", 160, 120, T_true, pipe.K))
        poses.append(pipe.step(T_wl_init=T_true if i == 0 else None).copy())
        assert pipe.tracking_good
    np.savez(os.path.join(out_dir, "track%d.npz" % rank), poses=np.array(poses), vol=pipe.vol.data, s0=pipe.s0, s1=pipe.s1)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("raycast", ["exact", "composite"])
def test_tracked_kinectfusion_on_slabs(tmp_path, raycast):
    """TrackingSlabPipeline (world 2, gloo, oracle operators): every rank derives the same poses without a broadcast;
    with the exact march they -- and the fused slabs -- equal the single-process TrackingPipeline bit for bit, with
    the nearest-hit composite they stay within a fraction of a millimetre of it."""
    import oracle_ops as ops
    from kangaroo_amd.pipeline import TrackingPipeline
    world = 2
    mp.spawn(_tracking_worker, args=(world, _free_port(), str(tmp_path), raycast), nprocs=world, join=True)
    bmin, bmax, near, far = scenes.SCENES["room"]
    ref = TrackingPipeline(ops, (N, N, N), bmin, bmax, 160, 120, near=near, far=far)
    want = []
    for i in range(4):
        T_true = scenes.orbit_pose(i, 60)
        ref.raw.MemcpyFromHost(scenes.render_depth("room", 160, 120, T_true, ref.K))
        want.append(ref.step(T_wl_init=T_true if i == 0 else None).copy())
    want = np.array(want)
    ranks = [np.load(os.path.join(str(tmp_path), "track%d.npz" % r)) for r in range(world)]
    assert np.array_equal(ranks[0]["poses"], ranks[1]["poses"])
    if raycast == "exact":
        assert np.array_equal(ranks[0]["poses"], want)
        for r in ranks:
            s0, s1 = int(r["s0"]), int(r["s1"])
            assert T.nan_equal(r["vol"], ref.vol.data[s0:s1])
    else:
        assert np.abs(ranks[0]["poses"][:, :3, 3] - want[:, :3, 3]).max() < 5e-4


@pytest.mark.parametrize("world,images", [(2, "all"), (3, "all"), (3, "root"), (8, "all")])
def test_direct_send_merge_equals_the_all_reduce_merge(tmp_path, world, images):
    """merge="direct" (strips to their owners by all-to-all, nearest hit per pixel, strips back by all-gather / gather) against
    merge="allreduce" (MIN of keys + SUM of payloads): the same winner per pixel, so the same images (a -0 component of a winning
    normal comes back as -0 from the direct merge and as +0 from the sum: compared as values)."""
    a, b = tmp_path / "direct", tmp_path / "allreduce"
    a.mkdir(); b.mkdir()
    mp.spawn(_worker, args=(world, _free_port(), str(a), "recompute", "composite", "replicate", images, "direct"), nprocs=world, join=True)
    mp.spawn(_worker, args=(world, _free_port(), str(b), "recompute", "composite", "replicate", images, "allreduce"), nprocs=world, join=True)
    for r in range(world if images == "all" else 1):
        x, y = np.load(str(a / ("rank%d.npz" % r))), np.load(str(b / ("rank%d.npz" % r)))
        assert np.isfinite(x["depth"]).mean() > 0.3
        for k in ("depth", "norm", "img"):
            assert np.array_equal(x[k], y[k], equal_nan=True), (r, k)


def test_composite_to_root_and_pose_broadcast(tmp_path):
    """images="root": the payload of the composite is reduced to rank 0 instead of all-reduced -- rank 0 ends up with exactly
    the images of images="all"; in the tracked loop rank 0 alone solves and broadcasts the pose, and every rank integrates its
    slab at it: same poses, same slabs as with every rank solving (world 3, gloo, oracle operators)."""
    world = 3
    a, b = tmp_path / "all", tmp_path / "root"
    a.mkdir(); b.mkdir()
    mp.spawn(_worker, args=(world, _free_port(), str(a), "recompute", "composite", "replicate", "all"), nprocs=world, join=True)
    mp.spawn(_worker, args=(world, _free_port(), str(b), "recompute", "composite", "replicate", "root"), nprocs=world, join=True)
    ra, rb = np.load(str(a / "rank0.npz")), np.load(str(b / "rank0.npz"))
    for k in ("depth", "norm", "img"):
        assert T.nan_equal(ra[k], rb[k]), k
    assert np.isfinite(rb["depth"]).mean() > 0.3
    for r in range(world):   # the volume never depended on the images
        assert T.nan_equal(np.load(str(a / ("rank%d.npz" % r)))["vol"], np.load(str(b / ("rank%d.npz" % r)))["vol"])
    ta, tb = tmp_path / "t_all", tmp_path / "t_root"
    ta.mkdir(); tb.mkdir()
    mp.spawn(_tracking_worker, args=(2, _free_port(), str(ta), "composite", "all"), nprocs=2, join=True)
    mp.spawn(_tracking_worker, args=(2, _free_port(), str(tb), "composite", "root"), nprocs=2, join=True)
    for r in range(2):
        x, y = np.load(str(ta / ("track%d.npz" % r))), np.load(str(tb / ("track%d.npz" % r)))
        assert np.array_equal(x["poses"], y["poses"]) and T.nan_equal(x["vol"], y["vol"])
