#!/usr/bin/env python3
"""Per-rank kernel time of the Z-slab pipeline for world = 1, 2, 4, 8 at 512^3 / 640x480, measured on ONE GPU by running
each rank's slab work in turn (SdfFuse of its planes + ghosts through kfx_sdf_fuse_slab, RaycastSdf of its slab, the
kernels of the composite merge: strips pack / merge / unpack of the direct-send merge, and pack / select / unpack of the all-reduce
merge).  No collectives are included: this is the compute side of the strong-scaling
curve the 8-GPU driver run measures (max over ranks = the slowest rank's kernels).  Usage: python scripts/slab_kernel_scaling.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from kangaroo_amd import roo, scenes  # noqa: E402
from kangaroo_amd.pipeline import slab_range  # noqa: E402

N, w, h, G, scene = 512, 640, 480, 2, "full"
K = scenes.intrinsics(w, h)
bmin, bmax, near, far = scenes.SCENES[scene]
tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
roo.set_math_mode("fast")
f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
raw = roo.Image(w, h).MemcpyFromHost(scenes.render_depth(scene, w, h, None, K))
roo.BilateralFilter(f, raw, **scenes.BILATERAL)
roo.DepthToVbo(vbo, f, K)
roo.NormalsFromVbo(nrm, vbo)
key = torch.empty(w * h, dtype=torch.int64, device="cuda")
payload = torch.empty(w * h * 5, dtype=torch.float32, device="cuda")


def timed(fn, reps=12):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


for world in (1, 2, 4, 8):
    per_rank = []
    for r in range(world):
        z0, z1 = slab_range(N, r, world)
        s0, s1 = max(z0 - G, 0), min(z1 + G, N)
        f32 = np.float32
        sz = f32(bmax[2]) - f32(bmin[2])
        lo = (bmin[0], bmin[1], float(f32(bmin[2]) + sz * f32(s0) / f32(N - 1)))
        hi = (bmax[0], bmax[1], float(f32(bmin[2]) + sz * f32(s1 - 1) / f32(N - 1)))
        v = roo.BoundedVolume(N, N, s1 - s0, lo, hi)
        roo.SdfReset(v, float("nan"))
        rd, rn, ri = roo.Image(w, h, pitch=w * 4), roo.Image(w, h, "f32x4", pitch=w * 16), roo.Image(w, h, pitch=w * 4)
        T_cw, T_wc = scenes.se3_inverse(scenes.identity_pose()), scenes.identity_pose()
        fuse = lambda: roo.SdfFuse(v, f, nrm, T_cw, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, full_extent=True, slab=(N, s0, bmin[2], bmax[2]))
        ray = lambda: roo.RaycastSdf(rd, rn, ri, v, T_wc, K, near, far, tr, True)

        def comp_allreduce():
            roo.CompositePack(rd, rn, ri, key, r)
            roo.CompositeSelect(rd, rn, ri, key, payload, r)
            roo.CompositeUnpack(rd, rn, ri, key, payload)
        S = roo.CompositeStripPixels(w, h, world)
        send = torch.zeros((world, roo.STRIP_PLANES, S), device="cuda")
        recv = torch.zeros((world, roo.STRIP_PLANES, S), device="cuda")
        merged = torch.zeros((roo.STRIP_PLANES, S), device="cuda")

        def comp():
            roo.CompositeStripsPack(rd, rn, ri, send, world)
            roo.CompositeStripsMerge(recv, merged, S, world)
            roo.CompositeStripsUnpack(rd, rn, ri, recv, world)
        for _ in range(2):
            fuse()
        per_rank.append((timed(fuse), timed(ray), timed(comp) if world > 1 else 0.0, timed(comp_allreduce) if world > 1 else 0.0))
        del v
        torch.cuda.empty_cache()
    tot = [a + b + c for a, b, c, _ in per_rank]
    worst = int(np.argmax(tot))
    print("world %d: slowest rank %d: fuse %.3f + raycast %.3f + merge kernels %.3f (all-reduce merge's: %.3f) = %.3f ms  (preprocess 0.03 ms and the "
          "merge's two collectives come on top); mean over ranks %.3f ms" % (world, worst, *per_rank[worst], tot[worst], float(np.mean(tot))), flush=True)
