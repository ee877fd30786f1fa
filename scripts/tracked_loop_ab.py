#!/usr/bin/env python3
"""The tracked KinectFusion loop (raycast of three pyramid levels in one launch -> device-side ICP refinement -> SdfFuse at
the estimated pose; fast numerics, 512^3, 640x480) with and without the brick summary and with track="auto" (the pipeline picks), scenes S_room and S_full.
Usage: python scripts/tracked_loop_ab.py [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from kangaroo_amd import roo, scenes  # noqa: E402
from kangaroo_amd.pipeline import TrackingPipeline  # noqa: E402

N, w, h = 512, 640, 480
frames_n = int(sys.argv[1]) if len(sys.argv) > 1 else 90
roo.set_math_mode("fast")
for scene in ("room", "full"):
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h)
    dev = [roo.Image(w, h).MemcpyFromHost(scenes.render_depth(scene, w, h, scenes.orbit_pose(i, 30), K)) for i in range(30)]
    for rep in range(2):
        for track in (False, True, "auto"):
            pipe = TrackingPipeline(roo, (N, N, N), bmin, bmax, w, h, near=near, far=far, device_icp=True, track=track)
            worst, lost = 0.0, 0
            ev = []
            for i in range(frames_n):
                if i == (40 if track == "auto" else 10):   # auto: the policy decides on frames 8-19 (plus the frames until its events are read)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                T_est = pipe.step(T_wl_init=scenes.orbit_pose(0, 30) if i == 0 else None, raw_image=dev[i % 30])
                worst = max(worst, float(np.linalg.norm(T_est[:3, 3] - scenes.orbit_pose(i % 30, 30)[:3, 3])))
                lost += 0 if pipe.tracking_good else 1
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / (frames_n - (40 if track == "auto" else 10)) * 1e3
            print("S_%-4s rep %d summary %-5s  %.4f ms/frame  %7.1f frames/s  worst position error %.2f mm, lost %d%s" % (
                scene, rep, track, ms, 1e3 / ms, worst * 1e3, lost,
                "  (%s)" % (pipe.track_decision or {}).get("chosen") if track == "auto" else ""), flush=True)
            del pipe
            torch.cuda.empty_cache()
