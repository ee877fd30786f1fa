#!/usr/bin/env python3
"""Where SdfFuse spends its time along z: per 64-slice range of the 512^3 volume, the launch time (slab entry point, so
voxel positions are those of the whole volume), the pixels-per-voxel ratio at the range's centre and the updated fraction.
Usage: python scripts/c3_zprofile.py [w h] [scene]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from kangaroo_amd import roo, scenes  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1280, 960)
scene = sys.argv[3] if len(sys.argv) > 3 else "room"
N, step = 512, 64
bmin, bmax, near, far = scenes.SCENES[scene]
K = scenes.intrinsics(w, h)
tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
vol = roo.BoundedVolume(N, N, N, bmin, bmax)
roo.SdfReset(vol, float("nan"))
raw = roo.Image(w, h).MemcpyFromHost(scenes.render_depth(scene, w, h, None, K))
f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
roo.BilateralFilter(f, raw, **scenes.BILATERAL)
roo.DepthToVbo(vbo, f, K)
roo.NormalsFromVbo(nrm, vbo)
Tid = scenes.identity_pose()
voxel = (bmax[0] - bmin[0]) / (N - 1)
print("%dx%d scene %s, env: %s" % (w, h, scene, {k: v for k, v in os.environ.items() if k.startswith("KFX_")}))
for mode in ("fast", "exact"):
    roo.set_math_mode(mode)
    tot = 0.0
    for z0 in range(0, N, step):
        slab = vol.ZSlab(z0, z0 + step)
        ms = []
        for i in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            roo.SdfFuse(slab, f, nrm, Tid, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, slab=(N, z0, bmin[2], bmax[2]))
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        t = sorted(ms[2:])[len(ms[2:]) // 2]
        tot += t
        zc = bmin[2] + (bmax[2] - bmin[2]) * (z0 + step / 2) / (N - 1)
        upd = float((~torch.isnan(slab.tensor()[..., 0])).float().mean())
        print("  %-5s z %3d..%3d  Z=%.2f m  r=%.2f px/voxel  %.4f ms  updated %.2f  -> %.0f GB/s algorithmic" % (
            mode, z0, z0 + step, zc, K[0] * voxel / zc, t, upd, 16.0 * upd * N * N * step / (t * 1e-3) / 1e9))
    print("  %s sum of ranges %.4f ms" % (mode, tot))
