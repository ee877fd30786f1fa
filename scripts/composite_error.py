#!/usr/bin/env python3
"""How far the nearest-hit composite of per-slab marches is from the single-volume RaycastSdf, 512^3 in 8 Z-slabs
emulated on one GPU (the slab volumes are separate allocations with 2 ghost planes, fused through kfx_sdf_fuse_slab).
Usage: python scripts/composite_error.py [N] [world]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from kangaroo_amd import roo, scenes  # noqa: E402
from kangaroo_amd.pipeline import slab_range  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w, h, G = 640, 480, 2
K = scenes.intrinsics(w, h)
for scene in ("room", "full"):
    bmin, bmax, near, far = scenes.SCENES[scene]
    tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    full = roo.BoundedVolume(N, N, N, bmin, bmax)
    roo.SdfReset(full, float("nan"))
    spans = [slab_range(N, r, world) for r in range(world)]
    stored = [(max(a - G, 0), min(b + G, N)) for a, b in spans]
    f32 = np.float32
    sz = f32(bmax[2]) - f32(bmin[2])
    slabs = []
    for s0, s1 in stored:
        lo = (bmin[0], bmin[1], float(f32(bmin[2]) + sz * f32(s0) / f32(N - 1)))
        hi = (bmax[0], bmax[1], float(f32(bmin[2]) + sz * f32(s1 - 1) / f32(N - 1)))
        v = roo.BoundedVolume(N, N, s1 - s0, lo, hi)
        roo.SdfReset(v, float("nan"))
        slabs.append(v)
    f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
    for i in range(3):
        T_wc = scenes.orbit_pose(i, 30)
        raw = roo.Image(w, h).MemcpyFromHost(scenes.render_depth(scene, w, h, T_wc, K))
        roo.BilateralFilter(f, raw, **scenes.BILATERAL)
        roo.DepthToVbo(vbo, f, K)
        roo.NormalsFromVbo(nrm, vbo)
        T_cw = scenes.se3_inverse(T_wc)
        roo.SdfFuse(full, f, nrm, T_cw, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
        for v, (s0, s1) in zip(slabs, stored):
            roo.SdfFuse(v, f, nrm, T_cw, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, full_extent=True, slab=(N, s0, bmin[2], bmax[2]))
    rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
    roo.RaycastSdf(rd, rn, ri, full, T_wc, K, near, far, tr, True)
    want_d, want_n = rd.tensor().clone(), rn.tensor().clone()
    best = torch.full((h, w), float("inf"), device="cuda")
    best_n = torch.zeros((h, w, 4), device="cuda")
    for v in slabs:
        roo.RaycastSdf(rd, rn, ri, v, T_wc, K, near, far, tr, True)
        d = torch.where(torch.isfinite(rd.tensor()), rd.tensor(), torch.full_like(best, float("inf")))
        take = d < best
        best = torch.where(take, d, best)
        best_n = torch.where(take.unsqueeze(-1), rn.tensor(), best_n)
    got_hit, want_hit = torch.isfinite(best), torch.isfinite(want_d)
    both = got_hit & want_hit
    err = (best[both] - want_d[both]).abs()
    voxel = (bmax[0] - bmin[0]) / (N - 1)
    ang = (best_n[both][:, :3] * want_n[both][:, :3]).sum(-1).clamp(-1, 1).acos() * 180 / np.pi
    print("S_%s %d^3 in %d slabs: hit masks differ on %d of %d pixels; depth |diff| median %.2e m (%.4f voxel), 99 %% %.2e m (%.3f voxel), "
          "max %.2e m (%.2f voxel); identical depth bits on %.1f %% of the hits; normal angle median %.4f deg, 99 %% %.3f deg" % (
              scene, N, world, int((got_hit != want_hit).sum()), w * h, float(err.median()), float(err.median()) / voxel,
              float(err.quantile(0.99)), float(err.quantile(0.99)) / voxel, float(err.max()), float(err.max()) / voxel,
              100.0 * float((err == 0).float().mean()), float(ang.median()), float(ang.quantile(0.99))), flush=True)
    del full, slabs
    torch.cuda.empty_cache()
