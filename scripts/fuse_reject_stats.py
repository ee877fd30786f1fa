#!/usr/bin/env python3
"""Where SdfFuse's work goes in a scene that does not update every voxel: the update predicate of cu_sdffusion.cu:26-44 evaluated
with torch (fp32, not bit-exact: statistics only) for every voxel of a 512^3 volume, summarised per wave-slice of the tiled
kernel (64 x 2 voxels of one slice for the 64 x 8 x 16 brick; 32 x 4 for the 32 x 8 x 16 brick) and per brick.
Usage: python scripts/fuse_reject_stats.py [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from kangaroo_amd import roo, scenes  # noqa: E402

N = 512
out = {}
for scene, w, h, bx, by in (("room", 640, 480, 64, 2), ("room", 1280, 960, 32, 4), ("full", 640, 480, 64, 2)):
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    T_wc = scenes.orbit_pose(3, 30)
    T_cw = torch.tensor(scenes.se3_inverse(T_wc), device="cuda")
    raw = roo.Image(w, h).MemcpyFromHost(scenes.render_depth(scene, w, h, T_wc, K))
    f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
    roo.BilateralFilter(f, raw, **scenes.BILATERAL)
    roo.DepthToVbo(vbo, f, K)
    roo.NormalsFromVbo(nrm, vbo)
    D, Nn = f.tensor(), nrm.tensor()[..., :3]
    ax = [torch.linspace(float(bmin[i]), float(bmax[i]), N, device="cuda") for i in range(3)]
    cls = torch.empty((N, N, N), dtype=torch.uint8, device="cuda")   # 0 updated, 1 out of the image band, 2 no depth / normal, 3 behind (sd <= -trunc), 4 grazing (costheta <= mincos)
    dd = torch.empty((N, N, N), dtype=torch.float16, device="cuda")  # (Z - md) / trunc
    for z0 in range(0, N, 32):
        zs = ax[2][z0:z0 + 32]
        X, Y, Z = torch.meshgrid(zs, ax[1], ax[0], indexing="ij")   # (z, y, x) order of the volume
        Pw = torch.stack([Z * 0 + ax[0].view(1, 1, N), Z * 0 + ax[1].view(1, N, 1), X], -1)
        Pc = Pw @ T_cw[:, :3].T + T_cw[:, 3]
        pu = K[2] + K[0] * Pc[..., 0] / Pc[..., 2]
        pv = K[3] + K[1] * Pc[..., 1] / Pc[..., 2]
        inb = (pu >= 2) & (pu < w - 2) & (pv >= 2) & (pv < h - 2) & (Pc[..., 2] > 0)
        ix = pu.floor().clamp(0, w - 2).long()
        iy = pv.floor().clamp(0, h - 2).long()
        fx, fy = (pu - ix).unsqueeze(-1), (pv - iy).unsqueeze(-1)

        def bil(img):
            a, b, c, d = img[iy, ix], img[iy, ix + 1], img[iy + 1, ix], img[iy + 1, ix + 1]
            if a.dim() == 3:
                a, b, c, d = a.unsqueeze(-1), b.unsqueeze(-1), c.unsqueeze(-1), d.unsqueeze(-1)
            top, bot = a + fx * (b - a), c + fx * (d - c)
            return top + fy * (bot - top)
        md = bil(D)[..., 0]
        mdn = bil(Nn)
        cost = -(mdn * Pc).sum(-1) / Pc.norm(dim=-1)
        sd = cost * (md - Pc[..., 2])
        wgt = cost / Pc[..., 2]
        c = torch.zeros_like(md, dtype=torch.uint8)
        c[~(cost > scenes.MIN_COS_THETA)] = 4
        c[sd <= -tr] = 3
        c[~torch.isfinite(md) | ~torch.isfinite(wgt)] = 2
        c[~inb] = 1
        cls[z0:z0 + 32] = c
        dd[z0:z0 + 32] = ((Pc[..., 2] - md) / tr).clamp(-100, 100).to(torch.float16)
    upd = cls == 0
    rep = {"updated_fraction": round(float(upd.float().mean()), 4)}
    for k, name in ((1, "out_of_image"), (2, "no_depth"), (3, "behind_surface"), (4, "grazing")):
        rep[name] = round(float((cls == k).float().mean()), 4)
    # bricks 64 x 8 x 16 (or 32 x 8 x 16): any update?
    BX = 64 if bx == 64 else 32
    ub = upd.view(N // 16, 16, N // 8, 8, N // BX, BX).permute(0, 2, 4, 1, 3, 5).reshape(N // 16, N // 8, N // BX, -1)
    live = ub.any(-1)
    rep["bricks_with_an_update"] = round(float(live.float().mean()), 4)
    rep["voxels_in_such_bricks"] = rep["bricks_with_an_update"]
    rep["updated_within_them"] = round(float(ub[live].float().mean()), 4)
    # wave-slices (bx x by voxels of one slice) inside bricks that have an update: how many have none, and why
    ws = upd.view(N, N // by, by, N // bx, bx).permute(0, 1, 3, 2, 4).reshape(N, N // by, N // bx, -1)
    cs = cls.view(N, N // by, by, N // bx, bx).permute(0, 1, 3, 2, 4).reshape(N, N // by, N // bx, -1)
    ds = dd.view(N, N // by, by, N // bx, bx).permute(0, 1, 3, 2, 4).reshape(N, N // by, N // bx, -1)
    liveb = live.repeat_interleave(16, 0).repeat_interleave(8 // by, 1).repeat_interleave(BX // bx, 2)
    dead = ~ws.any(-1)
    rep["wave_slices_in_live_bricks_without_update"] = round(float((dead & liveb).float().sum() / liveb.float().sum()), 4)
    allb = (cs == 3).all(-1) & liveb
    rep["  of_them_all_behind_surface"] = round(float(allb.float().sum() / liveb.float().sum()), 4)
    rep["  of_them_all_out_of_image"] = round(float(((cs == 1).all(-1) & liveb).float().sum() / liveb.float().sum()), 4)
    # a depth-only test (Z - md >= cut) with cut = trunc / mincos would catch of the all-behind ones:
    for cutf in (10.0, 3.0, 1.5):
        rep["  all_behind_by_%g_trunc" % cutf] = round(float((((ds >= cutf) | (cs == 1) | (cs == 2)).all(-1) & liveb & dead).float().sum() / liveb.float().sum()), 4)
    out["%s_%dx%d" % (scene, w, h)] = rep
    print(scene, w, h, json.dumps(rep), flush=True)
    del cls, dd, upd, ub, ws, cs, ds
    torch.cuda.empty_cache()
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
