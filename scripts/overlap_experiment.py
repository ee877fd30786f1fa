#!/usr/bin/env python3
"""Feasibility: SdfFuse in z-chunks on one stream, the exact slab march (kfx_raycast_sdf_slab, 'available planes' =
the chunks fused so far) following it on a second stream.  Fuse is HBM-bound, the march is a chain of dependent
misses with the GPU mostly idle, so the two should overlap.  Compares time and images with fuse -> RaycastSdf."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from kangaroo_amd import roo, scenes  # noqa: E402

N, w, h = 512, 640, 480
scene = sys.argv[1] if len(sys.argv) > 1 else "full"
nchunks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
bmin, bmax, near, far = scenes.SCENES[scene]
K = scenes.intrinsics(w, h)
tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
roo.set_math_mode("fast")
frames = []
for i in range(6):
    raw = roo.Image(w, h).MemcpyFromHost(scenes.render_depth(scene, w, h, scenes.orbit_pose(i, 30), K))
    f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
    roo.BilateralFilter(f, raw, **scenes.BILATERAL)
    roo.DepthToVbo(vbo, f, K)
    roo.NormalsFromVbo(nrm, vbo)
    frames.append((f, nrm, scenes.orbit_pose(i, 30)))


def sequential(vol, outs, i):
    f, nrm, T_wc = frames[i % 6]
    roo.SdfFuse(vol, f, nrm, scenes.se3_inverse(T_wc), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
    roo.RaycastSdf(*outs, vol, T_wc, K, near, far, tr, True)


sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
state = torch.empty((9, h, w), dtype=torch.float32, device="cuda")
edges = [round(k * N / nchunks) for k in range(nchunks + 1)]


def overlapped(vol, outs, i):
    f, nrm, T_wc = frames[i % 6]
    T_cw = scenes.se3_inverse(T_wc)
    cur = torch.cuda.current_stream()
    start = torch.cuda.Event()
    start.record(cur)
    sA.wait_event(start)
    sB.wait_event(start)
    for c in range(nchunks):
        z0, z1 = edges[c], edges[c + 1]
        sub = vol.SubVolume((0, 0, z0), (N, N, z1 - z0))
        roo.SdfFuse(sub, f, nrm, T_cw, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, full_extent=True,
                    slab=(N, z0, float(bmin[2]), float(bmax[2])), stream=sA.cuda_stream)
        ev = torch.cuda.Event()
        ev.record(sA)
        sB.wait_event(ev)
        avail = vol.SubVolume((0, 0, 0), (N, N, z1))
        roo.RaycastSdfSlab(state, c == 0, avail, (N, 0, float(bmin[2]), float(bmax[2])), 0, N, w, h, T_wc, K, near, far, tr, True,
                           stream=sB.cuda_stream)
    roo.RaycastStateToImages(*outs, state, stream=sB.cuda_stream)
    done = torch.cuda.Event()
    done.record(sB)
    cur.wait_event(done)
    doneA = torch.cuda.Event()
    doneA.record(sA)
    cur.wait_event(doneA)


def run(fn, reps=24):
    vol = roo.BoundedVolume(N, N, N, bmin, bmax)
    roo.SdfReset(vol, float("nan"))
    outs = (roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h))
    for i in range(6):
        fn(vol, outs, i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(vol, outs, i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, vol, outs


t_seq, v1, o1 = run(sequential)
t_ovl, v2, o2 = run(overlapped)
same = all(torch.equal(torch.nan_to_num(x.tensor(), nan=-7.0), torch.nan_to_num(y.tensor(), nan=-7.0)) for x, y in zip(o1, o2))
same_vol = torch.equal(torch.nan_to_num(v1.tensor(), nan=-7.0), torch.nan_to_num(v2.tensor(), nan=-7.0))
print(json.dumps({"scene": scene, "chunks": nchunks, "sequential_ms": round(t_seq, 4), "overlapped_ms": round(t_ovl, 4),
                  "images_identical": same, "volume_identical": same_vol}))
