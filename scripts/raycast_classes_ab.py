#!/usr/bin/env python3
"""RaycastSdf at 512^3 / 640x480 after a few tracked frames: plain march against the class-table march (environment knobs are
read once per process, so run it once per configuration: KFX_RAYCAST_SUMMARY, KFX_RAYCAST_CLASS_KB).
Prints kernel times (median of 20 launches, HIP events), tracked / untracked SdfFuse times and the image differences."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from kangaroo_amd import roo, scenes  # noqa: E402

N, w, h = 512, 640, 480
frames = int(os.environ.get("AB_FRAMES", "8"))
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in ("KFX_RAYCAST_SUMMARY", "KFX_RAYCAST_CLASS_KB") if k in os.environ) or "defaults"
for scene in sys.argv[1:] or ("full", "room"):
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    for math in os.environ.get("AB_MATH", "fast,exact").split(","):
        roo.set_math_mode(math)
        vol = roo.BoundedVolume(N, N, N, bmin, bmax)
        summ = roo.SdfSummary(vol)
        roo.SdfReset(vol, float("nan"), summary=summ)
        f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
        for i in range(frames):
            T_wc = scenes.orbit_pose(i, 30)
            roo.BilateralFilter(f, roo.Image(w, h).MemcpyFromHost(scenes.render_depth(scene, w, h, T_wc, K)), **scenes.BILATERAL)
            roo.DepthToVbo(vbo, f, K)
            roo.NormalsFromVbo(nrm, vbo)
            roo.SdfFuse(vol, f, nrm, scenes.se3_inverse(T_wc), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, summary=summ)
        T_wc = scenes.orbit_pose(frames - 1, 30)
        out = {}
        for name, kw in (("plain", {}), ("classes", {"summary": summ})):
            imgs = [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
            ms = []
            for i in range(24):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                roo.RaycastSdf(*imgs, vol, T_wc, K, near, far, tr, True, **kw)
                b.record()
                torch.cuda.synchronize()
                ms.append(a.elapsed_time(b))
            out[name] = (float(np.median(ms[4:])), [im.MemcpyToHost() for im in imgs])
        da, db = out["plain"][1][0], out["classes"][1][0]
        both = np.isfinite(da) & np.isfinite(db)
        same = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(out["plain"][1], out["classes"][1]))
        print("[%s] %-5s %-5s raycast plain %.4f ms, classes %.4f ms; identical images %s, hit flips %d, max |ddepth| %.3g (hits %d)" % (
            tag, scene, math, out["plain"][0], out["classes"][0], same, int((np.isfinite(da) != np.isfinite(db)).sum()),
            float(np.abs(da[both] - db[both]).max()) if both.any() else 0.0, int(both.sum())), flush=True)
        del vol, summ
        torch.cuda.empty_cache()
