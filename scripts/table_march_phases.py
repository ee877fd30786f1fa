#!/usr/bin/env python3
"""table_march_phases.py -- where the class-table march's time goes in S_full (512^3, 640x480): the tracked RaycastSdf launched
back to back (tables up to date: no rebuild) with the far plane pulled in, so that the rays end (as misses) before the wall's
band / half way / right behind the box's front face; plus the plain march for the same cut-offs."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gc  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402
from kangaroo_amd import roo, scenes  # noqa: E402
from kangaroo_amd.pipeline import FramePipeline  # noqa: E402

roo.set_math_mode("fast")
N, w, h = 512, 640, 480
scene = sys.argv[1] if len(sys.argv) > 1 else "full"
bmin, bmax, near, far = scenes.SCENES[scene]
pipe = FramePipeline(roo, (N, N, N), bmin, bmax, w, h, near=near, far=far, track=True)
poses = [scenes.orbit_pose(i, 30) for i in range(30)]
frames = [roo.Image(w, h).MemcpyFromHost(scenes.render_depth(scene, w, h, p, pipe.K)) for p in poses]
for i in range(90):
    pipe.step(poses[i % 30], frames[i % 30])
torch.cuda.synchronize()
gc.disable()   # (a generation-2 collection in the middle of a timed loop reads as a 1 ms kernel)
out = {}
T_wc = poses[7]
for label, f in (("far 8.0 (the frame's call)", 8.0), ("far 5.85 (rays end in front of the wall's band)", 5.85), ("far 5.0", 5.0), ("far 4.05", 4.05), ("far 3.9 (no ray enters the box)", 3.9)):
    res = {}
    for name, kw in (("tables", {"summary": pipe.summary}), ("plain", {})):
        for _ in range(5):
            roo.RaycastSdf(pipe.ray_d, pipe.ray_n, pipe.ray_i, pipe.vol, T_wc, pipe.K, near, f, pipe.trunc, True, **kw)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40):
            roo.RaycastSdf(pipe.ray_d, pipe.ray_n, pipe.ray_i, pipe.vol, T_wc, pipe.K, near, f, pipe.trunc, True, **kw)
        b.record()
        torch.cuda.synchronize()
        res[name + "_us"] = round(1e3 * a.elapsed_time(b) / 40, 2)
    res["hits"] = int(torch.isfinite(pipe.ray_d.tensor()).sum())
    out[label] = res
print(json.dumps(out, indent=1))
