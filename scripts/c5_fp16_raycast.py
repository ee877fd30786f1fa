#!/usr/bin/env python3
"""BASELINE config C5 on one GPU: a 2048^3 fp16 TSDF (32 GiB, resident in HBM) -- raycast-only
throughput.  The volume is the analytic sphere SDF (SdfSphere, the reference's own synthetic
volume, examples/Raycast.cpp:58) so that no 2048^3 fuse history is needed.
Usage: python scripts/c5_fp16_raycast.py [N] > profiles/r01_c5_fp16_raycast.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from kangaroo_amd import roo, scenes  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
w, h = 640, 480
K = scenes.intrinsics(w, h)
for kind in ("f16", "f32"):
    if kind == "f32" and N > 1536:
        continue  # 2048^3 fp32 = 64 GiB also fits, but config C5 is about the half volume
    vol = roo.BoundedVolume(N, N, N, (-1, -1, -1), (1, 1, 1), kind=kind)
    roo.SdfSphere(vol, (0.0, 0.0, 0.0), 0.9)
    torch.cuda.synchronize()
    T_wc = np.array([[1, 0, 0, 0.05], [0, 1, 0, -0.02], [0, 0, 1, -2.6]], np.float32)
    tr = float(2.0 * np.linalg.norm(vol.VoxelSizeUnits()))
    rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
    for _ in range(3):
        roo.RaycastSdf(rd, rn, ri, vol, T_wc, K, 0.1, 10.0, tr, True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record()
        roo.RaycastSdf(rd, rn, ri, vol, T_wc, K, 0.1, 10.0, tr, True)
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    hits = int(torch.isfinite(rd.tensor()).sum())
    d = rd.MemcpyToHost()
    # analytic check: depth along the central ray = 2.6 - 0.9 (+- a voxel)
    c = d[h // 2, w // 2]
    print("C5 %s %d^3 (%.1f GiB) raycast 640x480: median %.4f ms, min %.4f ms -> %.1f Mrays/s, %.0f fps; hits %d, centre depth %.4f (analytic ~%.4f)" % (
        kind, N, vol.ELEM * N ** 3 / 2 ** 30, ms[len(ms) // 2], ms[0], w * h / ms[len(ms) // 2] / 1e3, 1e3 / ms[len(ms) // 2], hits, c, 2.6 - 0.9), flush=True)
    del vol
    torch.cuda.empty_cache()
