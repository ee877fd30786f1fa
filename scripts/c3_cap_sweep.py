#!/usr/bin/env python3
"""Config C3 (512^3, 1280x960): SdfFuse time against the LDS tile capacity (KFX_FUSE_CAP, texels of 16 B)."""
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
        vol = roo.BoundedVolume(N, N, N, bmin, bmax)
        roo.SdfReset(vol, float("nan"))
        raw = roo.Image(w, h).MemcpyFromHost(scenes.render_depth(scene, w, h, None, K))
        f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
        roo.BilateralFilter(f, raw, **scenes.BILATERAL)
        roo.DepthToVbo(vbo, f, K)
        roo.NormalsFromVbo(nrm, vbo)
        for mode in ("fast", "exact"):
            roo.set_math_mode(mode)
            ms = []
            for i in range(14):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                roo.SdfFuse(vol, f, nrm, scenes.se3_inverse(scenes.orbit_pose(i, 30)), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
                b.record()
                torch.cuda.synchronize()
                ms.append(a.elapsed_time(b))
            ms = sorted(ms[2:])
            print("  %s %s %.4f ms" % (scene, mode, ms[len(ms) // 2]), end="")
        del vol
        torch.cuda.empty_cache()
    print(flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        for w, h in ((1280, 960), (640, 480)):
            for cap in sys.argv[1:] or ["1536", "2048", "3072", "4096", "6144"]:
                print("%dx%d cap %5s:" % (w, h, cap), end="", flush=True)
                subprocess.run([sys.executable, __file__, "--child", str(w), str(h)], env=dict(os.environ, KFX_FUSE_CAP=cap))
