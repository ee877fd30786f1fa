#!/usr/bin/env python3
"""Per-frame overhead of the composite path (3 kernels + 2 all-reduces) with a single-rank RCCL group on one GPU."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
from kangaroo_amd import roo, scenes
from kangaroo_amd.pipeline import SlabPipeline, FramePipeline
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29641")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
roo.set_math_mode("fast")
N, w, h = 512, 640, 480
bmin, bmax, near, far = scenes.SCENES["full"]
K = scenes.intrinsics(w, h)
pipe = SlabPipeline(roo, dist, (N, N, N), bmin, bmax, w, h, halo="recompute", K=K, near=near, far=far)
frames = [roo.Image(w, h, "f32", pitch=pipe.raw.pitch).MemcpyFromHost(scenes.render_depth("full", w, h, scenes.orbit_pose(i, 30), K)) for i in range(30)]
def run(with_comp, steps=60):
    for i in range(10):
        pipe.step(scenes.orbit_pose(i, 30), frames[i]);
        if with_comp: pipe.composite()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(steps):
        i = s % 30
        pipe.preprocess(frames[i]); pipe.fuse(scenes.orbit_pose(i, 30)); pipe.raycast(scenes.orbit_pose(i, 30))
        if with_comp: pipe.composite()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
print("slab pipeline world=1: %.4f ms/frame; with the composite path forced (3 kernels + 2 single-rank RCCL all-reduces): %.4f ms/frame" % (run(False), run(True)))
dist.destroy_process_group()
