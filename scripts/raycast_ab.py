#!/usr/bin/env python3
"""A/B of the RaycastSdf wave tile shape (KFX_RAYCAST_TILE = log2 of the tile width: 3 = 8x8 ... 6 = 64x1) at
512^3 on both scenes; each variant in its own process, outputs compared by checksum (must be identical)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(N=512, w=640, h=480, reps=30):
    import torch
    from kangaroo_amd import roo, scenes
    out = {}
    for scene in ("full", "room"):
        bmin, bmax, near, far = scenes.SCENES[scene]
        K = scenes.intrinsics(w, h)
        tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
        roo.set_math_mode("fast")
        vol = roo.BoundedVolume(N, N, N, bmin, bmax)
        roo.SdfReset(vol, float("nan"))
        f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
        for i in range(3):
            raw = roo.Image(w, h).MemcpyFromHost(scenes.render_depth(scene, w, h, scenes.orbit_pose(i, 30), K))
            roo.BilateralFilter(f, raw, **scenes.BILATERAL)
            roo.DepthToVbo(vbo, f, K)
            roo.NormalsFromVbo(nrm, vbo)
            roo.SdfFuse(vol, f, nrm, scenes.se3_inverse(scenes.orbit_pose(i, 30)), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
        rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
        ms = []
        for i in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            roo.RaycastSdf(rd, rn, ri, vol, scenes.orbit_pose(i % 30, 30), K, near, far, tr, True)
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        ms.sort()
        roo.RaycastSdf(rd, rn, ri, vol, scenes.orbit_pose(1, 30), K, near, far, tr, True)
        d = torch.nan_to_num(rd.tensor(), nan=-1.0).double().sum().item() + rn.tensor().double().sum().item() + ri.tensor().double().sum().item()
        out[scene] = {"ms_med": round(ms[len(ms) // 2], 4), "ms_min": round(ms[0], 4), "checksum": d}
        del vol
        torch.cuda.empty_cache()
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        for tv in (sys.argv[1:] or ["3", "4", "5", "6", "2"]):
            t, _, g = tv.partition(":")
            env = dict(os.environ, KFX_RAYCAST_TILE=t, KFX_RAYCAST_WG=g or "1")
            p = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            print("tile 2^%s x %d wg %s: %s" % (t, 64 >> int(t), g or "1", line[0][7:] if line else "FAILED " + p.stderr[-800:]), flush=True)
