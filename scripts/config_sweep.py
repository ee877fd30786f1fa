#!/usr/bin/env python3
"""Per-kernel timings of the BASELINE.json configurations that fit one GPU:
  C2  512^3,  640x480   (bench.py's workload)
  C3  512^3, 1280x960   full preprocess chain + BilateralFilter LDS tile-shape sweep
  C4' 1024^3, 640x480   the 8-GPU volume of config C4 on one GPU (8 GiB)
Each bilateral tile shape runs in its own process (KFX_BILATERAL_TILE is read once).
Usage: python scripts/config_sweep.py > profiles/r01_config_sweep.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {0: "32x8 (1 px/thread)", 1: "64x4 (1)", 2: "16x16 (1)", 3: "32x16 (2 px/thread, default)", 4: "32x32 (4)", 5: "64x16 (4)"}


def timeit(fn, reps=20):
    import torch
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


def child(N, w, h, math):
    import torch
    from kangaroo_amd import roo, scenes
    roo.set_math_mode(math)
    out = {}
    scene = "room"
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    vol = roo.BoundedVolume(N, N, N, bmin, bmax)
    roo.SdfReset(vol, float("nan"))
    raw = roo.Image(w, h).MemcpyFromHost(scenes.render_depth(scene, w, h, None, K))
    f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
    rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
    Tid = scenes.identity_pose()
    out["bilateral_ms"] = timeit(lambda: roo.BilateralFilter(f, raw, **scenes.BILATERAL))
    out["depth_to_vbo_ms"] = timeit(lambda: roo.DepthToVbo(vbo, f, K))
    out["normals_ms"] = timeit(lambda: roo.NormalsFromVbo(nrm, vbo))
    for i in range(3):
        roo.SdfFuse(vol, f, nrm, Tid, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
    out["sdf_fuse_ms"] = timeit(lambda: roo.SdfFuse(vol, f, nrm, Tid, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA))
    out["n_updated"] = roo.SdfFuseCount(vol, f, nrm, Tid, K, tr, scenes.MIN_COS_THETA)
    out["raycast_ms"] = timeit(lambda: roo.RaycastSdf(rd, rn, ri, vol, Tid, K, near, far, tr, True))
    out["reset_ms"] = timeit(lambda: roo.SdfReset(vol, float("nan")), reps=5)
    print("RESULT " + json.dumps(out), flush=True)


def run(N, w, h, math="fast", env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, __file__, "--child", str(N), str(w), str(h), math], env=e, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        return {"error": p.stderr[-800:]}
    return json.loads(line[0][7:])


def show(tag, N, w, h, r):
    if "error" in r:
        print("%-40s FAILED %s" % (tag, r["error"]))
        return
    px = w * h
    fuse_gbs = (16.0 * r["n_updated"] + 20.0 * px) / (r["sdf_fuse_ms"] * 1e-3) / 1e9
    print("%-40s bilateral %.4f ms (%.1f GB/s, %.1f Gexp/s) | vbo %.4f ms (%.0f GB/s) | normals %.4f ms (%.0f GB/s) | "
          "fuse %.4f ms (%.0f GB/s, %.1f%% updated) | raycast %.4f ms | reset %.4f ms (%.0f GB/s)" % (
              tag, r["bilateral_ms"], 8.0 * px / r["bilateral_ms"] / 1e6, 98.0 * px / r["bilateral_ms"] / 1e6,
              r["depth_to_vbo_ms"], 20.0 * px / r["depth_to_vbo_ms"] / 1e6, r["normals_ms"], 32.0 * px / r["normals_ms"] / 1e6,
              r["sdf_fuse_ms"], fuse_gbs, 100.0 * r["n_updated"] / N ** 3, r["raycast_ms"], r["reset_ms"],
              8.0 * N ** 3 / r["reset_ms"] / 1e6), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    else:
        print("# per-kernel medians (HIP events, 20 launches), scene S_room identity pose, MI355X")
        show("C2  512^3  640x480 fast", 512, 640, 480, run(512, 640, 480))
        show("C2  512^3  640x480 exact", 512, 640, 480, run(512, 640, 480, "exact"))
        print("# C3: 1280x960, bilateral LDS tile-shape sweep (tile = workgroup footprint, 256 threads)")
        for k, name in SHAPES.items():
            show("C3 512^3 1280x960 tile %s" % name, 512, 1280, 960, run(512, 1280, 960, env={"KFX_BILATERAL_TILE": str(k)}))
        print("# C4 volume size on one GPU (1024^3 = 8 GiB)")
        show("C4' 1024^3 640x480 fast", 1024, 640, 480, run(1024, 640, 480))
        show("C4' 1024^3 640x480 exact", 1024, 640, 480, run(1024, 640, 480, "exact"))
