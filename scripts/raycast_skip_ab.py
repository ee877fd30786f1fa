#!/usr/bin/env python3
"""RaycastSdf with and without the brick summary at 512^3 / 640x480 (both scenes, both numerics modes): kernel time after a few
tracked frames, fraction of bricks the march may skip, and the depth difference to the plain march."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from kangaroo_amd import roo, scenes  # noqa: E402

N, w, h = 512, 640, 480
for scene in ("full", "room"):
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    for math in ("fast", "exact"):
        roo.set_math_mode(math)
        vol = roo.BoundedVolume(N, N, N, bmin, bmax)
        summ = roo.SdfSummary(vol)
        roo.SdfReset(vol, float("nan"), summary=summ)
        f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
        fuse_t, fuse_u = [], []
        for i in range(6):
            T_wc = scenes.orbit_pose(i, 30)
            roo.BilateralFilter(f, roo.Image(w, h).MemcpyFromHost(scenes.render_depth(scene, w, h, T_wc, K)), **scenes.BILATERAL)
            roo.DepthToVbo(vbo, f, K)
            roo.NormalsFromVbo(nrm, vbo)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            roo.SdfFuse(vol, f, nrm, scenes.se3_inverse(T_wc), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, summary=summ)
            b.record()
            torch.cuda.synchronize()
            fuse_t.append(a.elapsed_time(b))
        out = {}
        for name, kw in (("plain", {}), ("summary", {"summary": summ})):
            d, n, im = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
            ms = []
            for i in range(12):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                roo.RaycastSdf(d, n, im, vol, scenes.orbit_pose(5, 30), K, near, far, tr, True, **kw)
                b.record()
                torch.cuda.synchronize()
                ms.append(a.elapsed_time(b))
            out[name] = (sorted(ms[2:])[5], d.MemcpyToHost())
        import ctypes as C
        from kangaroo_amd import _lib
        L = _lib.load()
        _lib.load_debug().kfx_debug_summary_export.restype = C.c_int
        _lib.load_debug().kfx_debug_summary_export.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
        dims = (C.c_int * 9)()
        Dall = torch.empty(64 * 64 * 64 + 16 * 16 * 16 + 64, dtype=torch.float32, device="cuda")
        R = torch.empty((64, 64, 64, 4), dtype=torch.float32, device="cuda")
        _lib.load_debug().kfx_debug_summary_export(summ.handle, 1e-5 if math == "fast" else 0.0, C.c_void_p(R.data_ptr()), C.c_void_p(Dall.data_ptr()), dims, None)
        torch.cuda.synchronize()
        D = Dall[:64 ** 3].view(64, 64, 64)
        st = R[..., 2].contiguous().view(torch.int32)
        rel = ((R[..., 1] - R[..., 0]) / R[..., 1].abs().clamp_min(1e-30))[st == 0]
        print("      bricks: D uniform %.3f, D nan %.3f, D sample %.3f | R states: values %.3f, nan %.3f, mixed %.3f | value bricks with rel. spread <= 1e-5: %.3f, median spread %.2g" % (
            float((D > 0).float().mean()), float(torch.isnan(D).float().mean()), float((D == -2).float().mean()),
            float((st == 0).float().mean()), float((st == 1).float().mean()), float((st == 2).float().mean()),
            float((rel <= 1e-5).float().mean()) if rel.numel() else 0.0, float(rel.median()) if rel.numel() else 0.0))
        D2 = Dall[64 ** 3: 64 ** 3 + 16 ** 3].view(16, 16, 16)
        D3 = Dall[64 ** 3 + 16 ** 3:]
        cls = lambda t: "uniform %.3f nan %.3f descend %.3f sample %.3f" % (float((t > 0).float().mean()), float(torch.isnan(t).float().mean()), float((t == -1).float().mean()), float((t == -2).float().mean()))
        print("      level 2:", cls(D2), "| level 3:", cls(D3))
        print("      level 2 uniform per z-layer:", [int((D2[z] > 0).sum()) for z in range(16)])
        da, db = out["plain"][1], out["summary"][1]
        both = np.isfinite(da) & np.isfinite(db)
        print("%-5s %-5s raycast plain %.4f ms, with summary %.4f ms; tracked fuse %.4f ms; hit flips %d, max |ddepth| %.3g (hits %d)" % (
            scene, math, out["plain"][0], out["summary"][0], sorted(fuse_t[1:])[2], int((np.isfinite(da) != np.isfinite(db)).sum()),
            float(np.abs(da[both] - db[both]).max()) if both.any() else 0.0, int(both.sum())), flush=True)
        del vol, summ
        torch.cuda.empty_cache()
