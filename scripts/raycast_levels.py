#!/usr/bin/env python3
"""RaycastSdf at the pyramid levels the tracking loop renders (main.cpp:280-288: levels with ICP iterations, 0 / 2 / 3 of a
640x480 camera): time per level, their sum, and the single multi-level launch (kfx_raycast_sdf_levels)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from kangaroo_amd import roo, scenes  # noqa: E402

N, w, h = 512, 640, 480
scene = sys.argv[1] if len(sys.argv) > 1 else "room"
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


def timed(fn, reps=30):
    ms = []
    for i in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        fn(i)
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    ms.sort()
    return round(ms[len(ms) // 2], 4)


levels = (0, 2, 3)
outs = {l: (roo.Image(w >> l, h >> l), roo.Image(w >> l, h >> l, "f32x4"), roo.Image(w >> l, h >> l)) for l in levels}
Ks = {l: scenes.intrinsics_level(K, l) for l in levels}
out = {"scene": scene}
for l in levels:
    out["level%d_ms" % l] = timed(lambda i, l=l: roo.RaycastSdf(*outs[l], vol, scenes.orbit_pose(i % 30, 30), Ks[l], near, far, tr, True))
out["three_launches_ms"] = timed(lambda i: [roo.RaycastSdf(*outs[l], vol, scenes.orbit_pose(i % 30, 30), Ks[l], near, far, tr, True) for l in levels])
if hasattr(roo, "RaycastSdfLevels"):
    ref = {}
    for l in levels:
        roo.RaycastSdf(*outs[l], vol, scenes.orbit_pose(1, 30), Ks[l], near, far, tr, True)
        ref[l] = [t.tensor().clone() for t in outs[l]]
    out["one_launch_ms"] = timed(lambda i: roo.RaycastSdfLevels([outs[l] for l in levels], vol, scenes.orbit_pose(i % 30, 30), [Ks[l] for l in levels], near, far, tr, True))
    roo.RaycastSdfLevels([outs[l] for l in levels], vol, scenes.orbit_pose(1, 30), [Ks[l] for l in levels], near, far, tr, True)
    torch.cuda.synchronize()
    out["identical"] = all(torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(t.tensor(), nan=-7.0)) for l in levels for a, t in zip(ref[l], outs[l]))
print(json.dumps(out))
