#!/usr/bin/env python3
"""Distribution of march lengths (samples per ray) of RaycastSdf at 512^3: needs a debug build of the library
(-DKFX_RAY_DEBUG_STEPS writes the sample count into the shade image), passed via KFX_LIB_PATH."""
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
    rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
    roo.RaycastSdf(rd, rn, ri, vol, scenes.orbit_pose(1, 30), K, near, far, tr, True)
    steps = ri.tensor().float().cpu()
    hit = torch.isfinite(rd.tensor().cpu())
    tiles = steps.reshape(h // 2, 2, w // 32, 32).permute(0, 2, 1, 3).reshape(-1, 64)   # 32 x 2 wave tiles
    wave_max = tiles.max(dim=1).values
    q = lambda t, p: float(torch.quantile(t, p))
    out[scene] = {"rays_marching": int((steps > 0).sum()), "mean": round(float(steps[steps > 0].mean()), 1),
                  "p50": q(steps[steps > 0], 0.5), "p90": q(steps[steps > 0], 0.9), "p99": q(steps[steps > 0], 0.99), "max": float(steps.max()),
                  "mean_steps_hit": round(float(steps[hit].mean()), 1), "mean_steps_miss": round(float(steps[(~hit) & (steps > 0)].mean()), 1) if ((~hit) & (steps > 0)).any() else 0,
                  "wave_max_mean": round(float(wave_max[wave_max > 0].mean()), 1), "wave_max_p90": q(wave_max[wave_max > 0], 0.9), "wave_max_max": float(wave_max.max()),
                  "sum_wave_max": float(wave_max.sum()), "sum_steps_div64": float(steps.sum() / 64)}
    del vol
    torch.cuda.empty_cache()
print(json.dumps(out))
