#!/usr/bin/env python3
"""Host-side cost of one frame of the Python frame loop (bench.py's step): how long the CPU needs to ISSUE a frame, against
how long the GPU needs to run it.  If the two are close the loop is launch-bound on a slower host."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from kangaroo_amd import roo, scenes  # noqa: E402
from kangaroo_amd.pipeline import FramePipeline  # noqa: E402

N, w, h, scene = 512, 640, 480, "full"
roo.set_math_mode("fast")
bmin, bmax, near, far = scenes.SCENES[scene]
K = scenes.intrinsics(w, h)
for track in (False, True):
    pipe = FramePipeline(roo, (N, N, N), bmin, bmax, w, h, K=K, near=near, far=far, track=track)
    poses = [scenes.orbit_pose(i, 30) for i in range(30)]
    frames = []
    for T_wc in poses:
        im = roo.Image(w, h, "f32", pitch=pipe.raw.pitch)
        im.MemcpyFromHost(scenes.render_depth(scene, w, h, T_wc, K))
        frames.append(im)
    for i in range(30):
        pipe.step(poses[i], frames[i])
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for s in range(n):
        pipe.step(poses[s % 30], frames[s % 30])
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("track=%s: host issues a frame in %.3f ms, frames complete every %.3f ms" % (track, 1e3 * t_issue / n, 1e3 * t_all / n), flush=True)
    if not track:
        pr = cProfile.Profile()
        pr.enable()
        for s in range(200):
            pipe.step(poses[s % 30], frames[s % 30])
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    del pipe
    torch.cuda.empty_cache()
