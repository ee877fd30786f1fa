#!/usr/bin/env python3
"""Is RaycastSdf limited by the dependent-gather chain (latency) or by a throughput limit?  Runs k identical raycasts of the
same volume concurrently on k streams (separate outputs) and compares the wall time with one raycast: a latency-bound
kernel overlaps almost for free, a throughput-bound one takes k times as long."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from kangaroo_amd import roo, scenes  # noqa: E402

N, w, h = 512, 640, 480
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
    res = {}
    for k in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(k)]
        outs = [(roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)) for _ in range(k)]
        ms = []
        for rep in range(20):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for s in streams:
                s.wait_event(a)
            for s, (rd, rn, ri) in zip(streams, outs):
                roo.RaycastSdf(rd, rn, ri, vol, scenes.orbit_pose(rep % 30, 30), K, near, far, tr, True, stream=s.cuda_stream)
            for s in streams:
                torch.cuda.current_stream().wait_stream(s)
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        ms.sort()
        res["x%d_ms" % k] = round(ms[len(ms) // 2], 4)
    out[scene] = res
    del vol
    torch.cuda.empty_cache()
print(json.dumps(out))
