#!/usr/bin/env python3
"""Minimal SdfFuse-only workload for PMC collection: 512^3, N launches.  Usage: fuse_only.py [scene] [reps] [width] [height] [math]
(640x480 = BASELINE config C2, 1280 960 = C3)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from kangaroo_amd import roo, scenes  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "full"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N = 512
w = int(sys.argv[3]) if len(sys.argv) > 3 else 640
h = int(sys.argv[4]) if len(sys.argv) > 4 else 480
if len(sys.argv) > 5:
    roo.set_math_mode(sys.argv[5])
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
for i in range(reps):
    roo.SdfFuse(vol, f, nrm, scenes.se3_inverse(scenes.orbit_pose(i, 30)), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
for i in range(reps):
    roo.RaycastSdf(rd, rn, ri, vol, scenes.orbit_pose(i, 30), K, near, far, tr, True)
torch.cuda.synchronize()
