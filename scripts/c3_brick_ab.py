#!/usr/bin/env python3
"""SdfFuse at 512^3 with the large (64 x 8 x 16) brick forced, the narrow (32 x 8 x 16) brick forced, and the per-range choice
(default), at 1280x960 (config C3) and 640x480 (C2), both scenes, both numerics modes; volumes compared bit for bit per mode.
Usage: python scripts/c3_brick_ab.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(w, h):
    import torch
    from kangaroo_amd import roo, scenes
    N = 512
    for scene in ("room", "full"):
        bmin, bmax, near, far = scenes.SCENES[scene]
        K = scenes.intrinsics(w, h)
        tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
        raw = roo.Image(w, h).MemcpyFromHost(scenes.render_depth(scene, w, h, None, K))
        f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
        roo.BilateralFilter(f, raw, **scenes.BILATERAL)
        roo.DepthToVbo(vbo, f, K)
        roo.NormalsFromVbo(nrm, vbo)
        for mode in ("fast", "exact"):
            roo.set_math_mode(mode)
            vol = roo.BoundedVolume(N, N, N, bmin, bmax)
            roo.SdfReset(vol, float("nan"))
            ms = []
            for i in range(14):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                roo.SdfFuse(vol, f, nrm, scenes.se3_inverse(scenes.orbit_pose(i, 30)), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
                b.record()
                torch.cuda.synchronize()
                ms.append(a.elapsed_time(b))
            ms = sorted(ms[2:])
            t = vol.tensor().contiguous().view(torch.int32).to(torch.int64)
            chk = (int(t.sum().item()) ^ int((t * torch.arange(1, 3, device=t.device)).sum().item())) & 0xffffffff
            upd = float((~torch.isnan(vol.tensor()[..., 0])).float().mean())
            print("  %s %-5s %.4f ms (chk %08x, %.0f%% observed)" % (scene, mode, ms[len(ms) // 2], chk, 100 * upd), end="")
            del vol, t
            torch.cuda.empty_cache()
    print(flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        for w, h in ((1280, 960), (640, 480)):
            for name, env in (("large brick", {"KFX_FUSE_BRICK": "0"}), ("narrow brick", {"KFX_FUSE_BRICK": "1"}), ("per range  ", {})):
                print("%dx%d %s:" % (w, h, name), flush=True)
                subprocess.run([sys.executable, __file__, "--child", str(w), str(h)], env=dict(os.environ, **env))
