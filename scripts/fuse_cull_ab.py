#!/usr/bin/env python3
"""A/B of the fast SdfFuse's brick cull by the tile's own costheta bound (fuse.hip, KFX_FUSE_CULL): SdfFuse time in the frame loop
(kfx_frame_step's device events, plain pair of kernels) and a checksum of the volume's bits after five frames from a reset, per
scene and image size, each variant in a process of its own (the knob is read once).  The checksums must agree: the cull changes
which bricks are evaluated, never a value.

Usage: python scripts/fuse_cull_ab.py [out.json]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = [("room", 640, 480), ("room", 1280, 960), ("full", 640, 480), ("full", 1280, 960)]


def child(N=512, frames=360):
    import numpy as np
    import torch
    from kangaroo_amd import roo, scenes
    from kangaroo_amd.pipeline import FramePipeline
    roo.set_math_mode("fast")
    out = {}
    for scene, w, h in CASES:
        bmin, bmax, near, far = scenes.SCENES[scene]
        K = scenes.intrinsics(w, h)
        pipe = FramePipeline(roo, (N, N, N), bmin, bmax, w, h, K=K, near=near, far=far, track=False, timing_slots=frames + 64)
        kf = pipe.kframe
        pipe.set_timing(kf.EVENTS_FUSE)
        poses = [scenes.orbit_pose(i, 30) for i in range(30)]
        imgs = []
        for T in poses:
            im = roo.Image(w, h, "f32", pitch=pipe.raw.pitch)
            im.MemcpyFromHost(scenes.render_depth(scene, w, h, T, K))
            imgs.append(im)
        for i in range(5):
            pipe.step(poses[i], imgs[i])
        torch.cuda.synchronize()
        t = pipe.vol.tensor().contiguous().view(torch.int32).to(torch.int64)
        t = torch.where((t & 0x7fffffff) > 0x7f800000, torch.full_like(t, 0x7fc00000), t)   # any NaN
        chk = int((t * (torch.arange(t.numel(), device=t.device).reshape(t.shape) % 8191 + 1)).sum().item())
        del t
        for i in range(1500):   # the clocks of a GPU that was idle
            pipe.step(poses[i % 30], imgs[i % 30])
        first = kf.count
        for i in range(frames):
            pipe.step(poses[i % 30], imgs[i % 30])
        tm = kf.timings(first, frames)
        out["%s_%dx%d" % (scene, w, h)] = {"sdf_fuse_ms": round(float(np.mean(tm[:, 1])), 5), "frame_period_ms": round(float(np.nanmean(tm[:, 4])), 5), "checksum": chk}
        del pipe, imgs
        torch.cuda.empty_cache()
    print("RESULT " + json.dumps(out), flush=True)


def main():
    if os.environ.get("KFX_AB_CHILD"):
        child()
        return
    res = {}
    for rnd in range(2):   # interleaved: off, on, off, on
        for cull in ("0", "1"):
            env = dict(os.environ, KFX_AB_CHILD="1", KFX_FUSE_CULL=cull)
            out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=1800)
            line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(out.stdout[-2000:], out.stderr[-2000:])
                sys.exit(1)
            res.setdefault("cull_" + cull, []).append(json.loads(line[0][7:]))
    summary = {}
    for case in res["cull_0"][0]:
        off = [r[case]["sdf_fuse_ms"] for r in res["cull_0"]]
        on = [r[case]["sdf_fuse_ms"] for r in res["cull_1"]]
        same = len({r[case]["checksum"] for v in res.values() for r in v}) == 1
        summary[case] = {"sdf_fuse_ms_cull_off": off, "sdf_fuse_ms_cull_on": on, "gain": round(1.0 - min(on) / min(off), 4), "same_volume_bits": same}
        print(case, summary[case], flush=True)
    if len(sys.argv) > 1:
        json.dump({"summary": summary, "runs": res}, open(sys.argv[1], "w"), indent=1)
    sys.exit(0 if all(v["same_volume_bits"] for v in summary.values()) else 2)


if __name__ == "__main__":
    main()
