"""How long after the first launch do kernel times settle?  Per-bucket mean SdfFuse / RaycastSdf time of the bench
workload (512^3, 640x480, S_full, fast numerics) from a cold start, and again after a 2 s idle gap."""
import sys, time
import numpy as np
import torch
from kangaroo_amd import roo, scenes
from kangaroo_amd.pipeline import FramePipeline

N, w, h = 512, 640, 480
scene = "full"
bmin, bmax, near, far = scenes.SCENES[scene]
roo.set_math_mode("fast")
K = scenes.intrinsics(w, h)
pipe = FramePipeline(roo, (N, N, N), bmin, bmax, w, h, K=K, near=near, far=far)
poses = [scenes.orbit_pose(i, 30) for i in range(30)]
frames = []
for T in poses:
    im = roo.Image(w, h, "f32", pitch=pipe.raw.pitch)
    im.MemcpyFromHost(scenes.render_depth(scene, w, h, T, K))
    frames.append(im)
torch.cuda.synchronize()
if "--count" in sys.argv:   # what bench.py does before its warm-up: 30 diagnostic launches
    for i in range(30):
        pipe.preprocess(frames[i])
        roo.SdfFuseCount(pipe.vol, pipe.filtered, pipe.normals, scenes.se3_inverse(poses[i]), K, pipe.trunc, pipe.mincostheta)
    torch.cuda.synchronize()

def run(n, label):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
    t0 = time.perf_counter()
    for s in range(n):
        i = s % 30
        pipe.preprocess(frames[i])
        ev[s][0].record()
        pipe.fuse(poses[i])
        ev[s][1].record()
        pipe.raycast(poses[i])
        ev[s][2].record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    f = np.array([a.elapsed_time(b) for a, b, _ in ev])
    r = np.array([b.elapsed_time(c) for _, b, c in ev])
    B = 150
    print(label, "total %.1f ms, %.0f fps" % (el * 1e3, n / el))
    for k in range(0, n, B):
        print("  steps %4d-%4d  (t = %4.0f ms)  fuse %.4f  raycast %.4f" % (k, k + B - 1, 1e3 * el * k / n, f[k:k + B].mean(), r[k:k + B].mean()))

run(900, "cold start")
run(450, "back to back")
