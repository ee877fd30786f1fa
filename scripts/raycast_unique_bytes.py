#!/usr/bin/env python3
"""SURVEY 8(d), RaycastSdf: distinct voxels touched (U) and samples taken per frame at 512^3 / 640x480, counted by the
oracle with a bitmap (CPU only, ~1 min).  Algorithmic bytes = 8 B x U + 24 B x w*h; gather volume = 64 B x steps x 4
(four 16-byte corner loads per trilinear sample, each in its own line).  Usage: python scripts/raycast_unique_bytes.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from kangaroo_amd import scenes  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
w, h = 640, 480
K = scenes.intrinsics(w, h)
for scene in ("full", "room"):
    bmin, bmax, near, far = scenes.SCENES[scene]
    tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    vol = oracle.Volume(N, N, N, bmin, bmax)
    oracle.sdf_reset(vol, float("nan"))
    f, vbo, nrm = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h, channels=4)
    for i in range(2):
        T_wc = scenes.orbit_pose(i, 30)
        raw = oracle.Image.from_numpy(scenes.render_depth(scene, w, h, T_wc, K))
        oracle.bilateral(f, raw, nthreads=0, **scenes.BILATERAL)
        oracle.depth_to_vbo(vbo, f, K)
        oracle.normals_from_vbo(nrm, vbo)
        oracle.sdf_fuse(vol, f, nrm, scenes.se3_inverse(T_wc), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, nthreads=0)
    rd, rn, ri = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
    st, U = oracle.raycast_sdf_touch(rd, rn, ri, vol, scenes.orbit_pose(1, 30), K, near, far, tr, True)
    alg = 8 * U + 24 * w * h
    print("S_%s %d^3: rays %d, hits %d, samples %d (%.1f per ray), distinct voxels U = %d (%.1f %% of the volume), "
          "algorithmic bytes %.3f GB, 64-byte gather volume %.3f GB" % (scene, N, st["rays"], st["hits"], st["steps"], st["steps"] / max(st["rays"], 1),
                                                                    U, 100.0 * U / N ** 3, alg / 1e9, 64.0 * 4 * st["steps"] / 1e9), flush=True)
