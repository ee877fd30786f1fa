"""Two (or more) ranks of the Z-slab pipeline on ONE GPU, real HIP operators, collectives over gloo (RCCL needs one GPU per
rank; the data path, the composite kernels and the march hand-over are the same code).  Launched by
tests/test_gpu_multi_rank.py through torch.distributed.run; every rank checks its result against the single-volume pipeline
it runs itself and prints MP_OK.

    python -m torch.distributed.run --nproc-per-node 2 tests/mp_slab_gpu.py <halo> <raycast> [tracking]
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kangaroo_amd import roo, scenes  # noqa: E402
from kangaroo_amd.pipeline import FramePipeline, SlabPipeline, TrackingPipeline, TrackingSlabPipeline  # noqa: E402
import kfx_testlib as T  # noqa: E402

halo, raycast = sys.argv[1], sys.argv[2]
tracking = len(sys.argv) > 3 and sys.argv[3] == "tracking"
cframe = len(sys.argv) > 3 and sys.argv[3] == "cframe"   # the frame as ONE kfx_slab_frame_step call per rank, collectives through Comm.torch(dist)
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
N, W, H, FRAMES, scene = 96, 160, 120, 3, "room"
bmin, bmax, near, far = scenes.SCENES[scene]


def same(a, b):
    return T.nan_equal(a, b)


if cframe:
    # kfx_slab_frame (include/kfx_slab.h) under real collectives: the C call's bits against the operator-by-operator SlabPipeline
    # (composite: the same restart semantics, so the same images) and against the single volume (exact: bit-identical, any tile count)
    # exact + overlap: frames pipelined across the ranks (the final exchange of frame k on the side stream through a second process
    # group -- kfx_comm::dup -- under frame k + 1; 2 and 3 image sets in flight over 7 frames)
    variants = ([dict(tiles=1), dict(tiles=4), dict(tiles=7), dict(tiles=4, overlap=True), dict(tiles=1, overlap=True, pipeline=2)] if raycast == "exact" else
                [dict(merge="direct"), dict(merge="allreduce"), dict(merge="direct", overlap=True)])
    if halo == "recompute" and raycast == "exact":
        variants.append(dict(tiles=4, ghost=2))   # (the default here is kfx_slab_exact_ghost's width: the hand-over without its last stage)
    if halo == "exchange":
        variants = [v for v in variants if not (v.get("overlap") and raycast == "composite")] + [dict(inputs="broadcast")] + ([dict(inputs="broadcast", overlap=True)] if raycast == "exact" else [])
    FRAMES = 7
    py = SlabPipeline(roo, dist, (N, N, N), bmin, bmax, W, H, halo=halo, raycast=raycast, near=near, far=far)
    ref = FramePipeline(roo, (N, N, N), bmin, bmax, W, H, near=near, far=far)
    cs = [SlabPipeline(roo, dist, (N, N, N), bmin, bmax, W, H, halo=halo, raycast=raycast, near=near, far=far, driver="c", **v) for v in variants]
    for i in range(FRAMES):
        T_wc = scenes.orbit_pose(i, 8)
        depth = scenes.render_depth(scene, W, H, T_wc, py.K)
        for p_ in [py, ref] + cs:
            p_.raw.MemcpyFromHost(depth)
        py.step(T_wc)
        ref.preprocess()
        roo.SdfFuse(ref.vol, ref.filtered, ref.normals, scenes.se3_inverse(T_wc), ref.K, ref.trunc, ref.max_w, ref.mincostheta)
        ref.raycast(T_wc)
        for c_ in cs:
            c_.step(T_wc)
    for c_ in cs:
        c_.wait_composite()   # (pipelined frames: the trailing final exchanges; ray_d / ray_n / ray_i = the last frame's set)
        c_.sframe.sync()
    torch.cuda.synchronize()
    full = ref.vol.MemcpyToHost()
    want = [x.MemcpyToHost() for x in ((ref.ray_d, ref.ray_n, ref.ray_i) if raycast == "exact" else (py.ray_d, py.ray_n, py.ray_i))]
    for v, c_ in zip(variants, cs):
        assert same(c_.vol.MemcpyToHost(), full[c_.s0:c_.s1]), "rank %d %r: slab planes differ from the single volume" % (rank, v)
        got = [x.MemcpyToHost() for x in (c_.ray_d, c_.ray_n, c_.ray_i)]
        for a_, b_ in zip(got, want):
            assert same(a_, b_), "rank %d %r: the C frame's images differ" % (rank, v)
        if raycast == "exact":
            T_ = c_.sframe.last_steps
            # world + tiles - 1 token steps, + the normals' stage unless the ghost planes are wide enough to do without (recomputed ghosts)
            assert T_ == world + v["tiles"] - 1 + (1 if c_.GHOST <= 2 else 0) if "tiles" in v else T_ > 0, (v, T_, c_.GHOST)
        t = c_.sframe.timings(c_.sframe.count - 2, 2)
        assert t.shape == (2, 6) and np.isfinite(t[:, :3]).all() and (t[:, :3] >= 0).all() and np.isfinite(t[0, 5]), t
        assert np.isfinite(t[:, 3]).all() == (raycast == "composite" or bool(v.get("overlap")))
elif not tracking:
    pipe = SlabPipeline(roo, dist, (N, N, N), bmin, bmax, W, H, halo=halo, raycast=raycast, near=near, far=far)
    ref = FramePipeline(roo, (N, N, N), bmin, bmax, W, H, near=near, far=far)
    for i in range(FRAMES):
        T_wc = scenes.orbit_pose(i, 8)
        depth = scenes.render_depth(scene, W, H, T_wc, pipe.K)
        pipe.raw.MemcpyFromHost(depth)
        pipe.step(T_wc)
        ref.raw.MemcpyFromHost(depth)
        ref.preprocess()
        roo.SdfFuse(ref.vol, ref.filtered, ref.normals, scenes.se3_inverse(T_wc), ref.K, ref.trunc, ref.max_w, ref.mincostheta)
        ref.raycast(T_wc)
    torch.cuda.synchronize()
    # (1) every stored plane (owned + ghost) equals the same plane of the single volume, bit for bit
    full = ref.vol.MemcpyToHost()
    mine = pipe.vol.MemcpyToHost()
    assert same(mine, full[pipe.s0:pipe.s1]), "rank %d: slab planes differ from the single volume" % rank
    d, n, im = pipe.ray_d.MemcpyToHost(), pipe.ray_n.MemcpyToHost(), pipe.ray_i.MemcpyToHost()
    rd, rn, ri = ref.ray_d.MemcpyToHost(), ref.ray_n.MemcpyToHost(), ref.ray_i.MemcpyToHost()
    if raycast == "exact":   # (2a) the handed-over march reproduces RaycastSdf exactly
        assert same(d, rd) and same(n, rn) and same(im, ri), "rank %d: exact slab march differs from RaycastSdf" % rank
        assert 1 <= pipe.rounds <= world + 3
    else:                    # (2b) nearest-hit composite: same hits up to the slab-entry resampling
        hit, rhit = np.isfinite(d), np.isfinite(rd)
        assert (hit != rhit).mean() < 0.01, (hit != rhit).mean()
        both = hit & rhit
        voxel = (bmax[0] - bmin[0]) / (N - 1)
        err = np.abs(d[both] - rd[both])   # the march restarts at each slab entry: a fraction of a voxel (as tests/test_multi_rank_cpu.py)
        assert both.mean() > 0.3 and np.median(err) < 0.02 * voxel and np.quantile(err, 0.99) < 0.5 * voxel, (np.median(err) / voxel, np.quantile(err, 0.99) / voxel)
        assert (np.abs(n[both] - rn[both]).max(axis=1) < 0.05).mean() > 0.98
        assert (n[hit][:, 3] == 1).all() and (n[~hit] == 0).all() and (im[~hit] == 0).all()
    # (2c) several levels rendered and merged at once (one launch, one pair of all-reduces) = level by level
    lv = (0, 1, 2)
    Ks = [scenes.intrinsics_level(pipe.K, l) for l in lv]
    mk = lambda: [(roo.Image(W >> l, H >> l), roo.Image(W >> l, H >> l, "f32x4"), roo.Image(W >> l, H >> l)) for l in lv]
    a, b = mk(), mk()
    T_last = scenes.orbit_pose(FRAMES - 1, 8)
    for (dd, nn, ii), Kl in zip(a, Ks):
        pipe.raycast_into(dd, nn, ii, Kl, T_last)
    pipe.raycast_levels_into(b, Ks, T_last)
    for x, y in zip(a, b):
        for p_, q_ in zip(x, y):
            assert same(p_.MemcpyToHost(), q_.MemcpyToHost()), "rank %d: merged levels differ from level-by-level" % rank
    assert same(a[0][0].MemcpyToHost(), d)
    # (2d) the direct-send merge (strips: all-to-all + all-gather, the default) = the key / payload merge (two all-reduces), for one
    # image and for several levels at once
    if raycast == "composite":
        assert pipe.merge == "direct"
        pipe.merge = "allreduce"
        c, e = mk(), mk()
        for (dd, nn, ii), Kl in zip(c, Ks):
            pipe.raycast_into(dd, nn, ii, Kl, T_last)
        pipe.raycast_levels_into(e, Ks, T_last)
        pipe.merge = "direct"
        for x, y, z in zip(a, c, e):
            for p_, q_, r_ in zip(x, y, z):
                assert same(p_.MemcpyToHost(), q_.MemcpyToHost()) and same(p_.MemcpyToHost(), r_.MemcpyToHost()), "rank %d: direct merge differs from the all-reduce merge" % rank
    # (3) all ranks hold the same images
    chk = torch.tensor(np.nan_to_num(d, nan=-1.0).view(np.int32).astype(np.int64).sum()).reshape(1)
    both = torch.cat([chk, -chk])
    dist.all_reduce(both, op=dist.ReduceOp.MAX)
    assert int(both[0]) == -int(both[1]), "ranks hold different images"
else:
    pipe = TrackingSlabPipeline(roo, dist, (N, N, N), bmin, bmax, 320, 240, halo=halo, raycast=raycast, near=near, far=far)
    ref = TrackingPipeline(roo, (N, N, N), bmin, bmax, 320, 240, near=near, far=far)
    for i in range(4):
        T_true = scenes.orbit_pose(i, 30)
        depth = scenes.render_depth(scene, 320, 240, T_true, pipe.K)
        pipe.raw.MemcpyFromHost(depth)
        ref.raw.MemcpyFromHost(depth)
        Ta = pipe.step(T_wl_init=T_true if i == 0 else None)
        Tb = ref.step(T_wl_init=T_true if i == 0 else None)
        if raycast == "exact":
            assert np.array_equal(Ta, Tb), "rank %d frame %d: tracked pose differs from the single-GPU loop" % (rank, i)
        else:
            assert np.abs(Ta - Tb).max() < 2e-3, np.abs(Ta - Tb).max()
        assert pipe.tracking_good
    torch.cuda.synchronize()
    if raycast == "exact":
        assert same(pipe.vol.MemcpyToHost(), ref.vol.MemcpyToHost()[pipe.s0:pipe.s1])
dist.barrier()
print("MP_OK rank %d of %d (%s, %s%s)" % (rank, world, halo, raycast, ", tracking" if tracking else ""), flush=True)
dist.destroy_process_group()
