#!/usr/bin/env python3
"""Timing of the SURVEY 8(f) rows on one MI355X: tracked frame (ICP), colour fusion, colour raycast.
Prints one JSON object; run under rocprofv3 --kernel-trace --stats for the per-kernel view."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from kangaroo_amd import roo, scenes, tracking  # noqa: E402
from kangaroo_amd.pipeline import TrackingPipeline  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
w, h, scene = 640, 480, "room"
bmin, bmax, near, far = scenes.SCENES[scene]
K = scenes.intrinsics(w, h)
out = {"volume": N, "image": [w, h], "scene": scene}


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


# ---- tracked KinectFusion (f-2): raycast 3 levels + 6 ICP iterations with host solves + fuse ----
roo.set_math_mode("fast")
pipe = TrackingPipeline(roo, (N, N, N), bmin, bmax, w, h, near=near, far=far)
frames = [scenes.render_depth(scene, w, h, scenes.orbit_pose(i, 30), K) for i in range(12)]
dev = [roo.Image(w, h).MemcpyFromHost(f) for f in frames]
worst = 0.0
t_frames = []
for i in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T_est = pipe.step(T_wl_init=scenes.orbit_pose(0, 30) if i == 0 else None, raw_image=dev[i])
    torch.cuda.synchronize()
    t_frames.append((time.perf_counter() - t0) * 1e3)
    worst = max(worst, float(np.linalg.norm(T_est[:3, 3] - scenes.orbit_pose(i, 30)[:3, 3])))
out["tracked_frame_ms"] = round(float(np.median(t_frames[2:])), 4)
# the same loop with the refinement iterations chained on the device (kfx_icp_refine)
pipe2 = TrackingPipeline(roo, (N, N, N), bmin, bmax, w, h, near=near, far=far, device_icp=True)
t2 = []
for i in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe2.step(T_wl_init=scenes.orbit_pose(0, 30) if i == 0 else None, raw_image=dev[i])
    torch.cuda.synchronize()
    t2.append((time.perf_counter() - t0) * 1e3)
out["tracked_frame_device_icp_ms"] = round(float(np.median(t2[2:])), 4)
del pipe2
out["tracked_fps"] = round(1e3 / out["tracked_frame_ms"], 1)
out["tracking_worst_position_error_mm"] = round(worst * 1e3, 3)
# one ICP evaluation at full resolution (kernel + block sum + 116-byte blocking readback)
KT = (tracking.k_matrix(K) @ np.eye(4)[:3]).astype(np.float32)
I34 = np.eye(4, dtype=np.float32)[:3]
out["icp_call_ms_640x480"] = round(timed(lambda: roo.PoseRefinementProjectiveIcpPointPlane(
    pipe.kin_v[0], pipe.pyr_v[0], pipe.pyr_n[0], KT, I34, 0.1, pipe.scratch, pipe.debug), reps=50), 4)
del pipe

# ---- colour fusion / colour raycast (f-3) ----
roo.set_math_mode("exact")
vol = roo.BoundedVolume(N, N, N, bmin, bmax)
cvol = roo.BoundedVolume(N, N, N, bmin, bmax, kind="c32")
roo.SdfReset(vol, float("nan"))
roo.ColorReset(cvol)
f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
roo.BilateralFilter(f, dev[0], **scenes.BILATERAL)
roo.DepthToVbo(vbo, f, K)
roo.NormalsFromVbo(nrm, vbo)
rgb = roo.Image(w, h, "u8x3")
rgb.MemcpyFromHost(np.random.default_rng(0).integers(0, 256, (h, w, 3), dtype=np.uint8))
T_cw = scenes.se3_inverse(scenes.orbit_pose(0, 30))
tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
fuse_c = lambda: roo.SdfFuseColor(vol, cvol, f, nrm, T_cw, K, rgb, T_cw, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
out["color_fuse_ms"] = round(timed(fuse_c), 4)
updated = int(roo.SdfFuseCount(vol, f, nrm, T_cw, K, tr, scenes.MIN_COS_THETA))
out["color_fuse_updated_voxels"] = updated
# 16 B SDF read+write + 8 B colour read+write per updated voxel, one pass over depth + normals + rgb
alg = 24 * updated + (4 + 16 + 3) * w * h
out["color_fuse_GBps"] = round(alg / (out["color_fuse_ms"] * 1e-3) / 1e9, 1)
roo.set_math_mode("fast")
out["color_fuse_fast_ms"] = round(timed(fuse_c), 4)
out["color_fuse_fast_GBps"] = round(alg / (out["color_fuse_fast_ms"] * 1e-3) / 1e9, 1)
roo.set_math_mode("exact")
rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
T_wc = scenes.orbit_pose(0, 30)
out["color_raycast_ms"] = round(timed(lambda: roo.RaycastSdfColor(rd, rn, ri, vol, cvol, T_wc, K, near, far, tr, True)), 4)
out["grey_raycast_ms"] = round(timed(lambda: roo.RaycastSdf(rd, rn, ri, vol, T_wc, K, near, far, tr, True)), 4)
out["grey_fuse_exact_ms"] = round(timed(lambda: roo.SdfFuse(vol, f, nrm, T_cw, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)), 4)
# ---- mesh extraction (f-4): count + prefix sum + emit on the fused 512^3 volume, then PCIe + file ----
from kangaroo_amd import mesh  # noqa: E402
mesh.ExtractMesh(vol, cvol)
torch.cuda.synchronize()
t0 = time.perf_counter()
v, n, c = mesh.ExtractMesh(vol, cvol)
torch.cuda.synchronize()
out["mesh_extract_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
out["mesh_triangles"] = int(v.shape[0] // 3)
t0 = time.perf_counter()
mesh.SaveMesh("/tmp/next_rows_mesh", vol, cvol)
out["mesh_save_ply_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
os.remove("/tmp/next_rows_mesh.ply")
print(json.dumps(out))
