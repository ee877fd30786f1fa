#!/usr/bin/env python3
"""march_speculation_model.py -- CPU model (numpy + the oracle's volume; no GPU): how predictable are the steps of the longest
rays, and how many dependent round trips would a march need whose idle lanes sample AHEAD for the rays still under way?

RaycastSdf's kernel lasts as long as its longest chains of dependent samples.  The chain can only be cut by
sampling positions before the previous sample is known -- i.e. by predicting the step.  The reference's step is
max(sdf, min_delta) for a positive sample and trunc for a NaN one (cu_raycast.cu:77-80), so it is predictable exactly when the
sample is NaN, when it is +trunc (free space) or when 0 < sdf <= min_delta (the step is min_delta whatever the value).  This
script fuses the S_room orbit into a volume with the oracle, marches every ray of one pose in numpy (float32, same expressions),
records each ray's step sequence and replays it under a team model: a wave of 64 lanes serves its `a` active rays with
T = min(Tmax, 64 // a) lanes each; a team samples T positions ahead assuming every step repeats the previous one, and accepts
the prefix of them whose assumed positions are the real chain's (bit-equal steps).  Output: round trips on the critical path of
each 32 x 2 wave tile, with and without teams."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (analysis script: not product code)
from kangaroo_amd import scenes  # noqa: E402

f32 = np.float32


def fuse_stream(scene, N, w, h, frames):
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    vol = oracle.Volume(N, N, N, bmin, bmax)
    oracle.sdf_reset(vol, float("nan"))
    for i in range(frames):
        T_wc = scenes.orbit_pose(i % 30, 30)
        raw = oracle.Image.from_numpy(scenes.render_depth(scene, w, h, T_wc, K))
        f, vbo, nrm = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h, channels=4)
        oracle.bilateral(f, raw, nthreads=0, **scenes.BILATERAL)
        oracle.depth_to_vbo(vbo, f, K)
        oracle.normals_from_vbo(nrm, vbo)
        oracle.sdf_fuse(vol, f, nrm, scenes.se3_inverse(T_wc), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, nthreads=0)
    return vol, K, tr, near, far


def march_all(vol, K, tr, near, far, T_wc, w, h, max_steps=400):
    """Per ray: list of steps taken (float32), as the reference march takes them."""
    val = vol.data[..., 0].copy()
    if os.environ.get("SNAP", "1") != "0":
        # fast numerics (what the benchmark runs) keep observed free space at +trunc bit for bit (the incremental running average,
        # DESIGN.md 5.1); the exact oracle's average drifts by a few ulp per frame: snap those cells to model the fast-mode volume
        with np.errstate(invalid="ignore"):
            val[np.abs(val - f32(tr)) <= f32(1e-5) * f32(tr)] = f32(tr)
    D, H, W = val.shape
    bmin, bmax = np.asarray(vol.boxmin, f32), np.asarray(vol.boxmax, f32)
    size = (bmax - bmin).astype(f32)
    dims1 = np.array([W - 1, H - 1, D - 1], f32)
    hi2 = np.array([W - 2, H - 2, D - 2], f32)
    voxel = (size / dims1).astype(f32)
    T = np.asarray(T_wc, f32).reshape(3, 4)
    u, v = np.meshgrid(np.arange(w, dtype=f32), np.arange(h, dtype=f32))
    rc = np.stack([(u - K[2]) / K[0], (v - K[3]) / K[1], np.ones_like(u)], -1).astype(f32)
    rw = (rc @ T[:, :3].T).astype(f32)
    c = T[:, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        ta, tb = (bmin - c) / rw, (bmax - c) / rw
    tmin, tmax = np.minimum(ta, tb).max(-1), np.maximum(ta, tb).min(-1)
    lam = np.maximum(tmin, f32(near)).astype(f32)
    lam_max = np.minimum(tmax, f32(far)).astype(f32)
    active = lam < lam_max
    last = np.full((h, w), np.nan, f32)
    steps = np.zeros((max_steps, h, w), f32)     # 0: no step (ray over)
    kinds = np.zeros((max_steps, h, w), np.int8)  # 1 min_delta, 2 trunc (NaN or +trunc sample), 3 other
    n = np.zeros((h, w), np.int32)
    min_delta = voxel[0]
    for k in range(max_steps):
        if not active.any():
            break
        pos = c + rw * lam[..., None]
        pf = ((pos - bmin) / size * dims1).astype(f32)
        base = np.clip(np.floor(pf), 0, hi2)
        fr = (pf - base).astype(f32)
        ix, iy, iz = (base[..., i].astype(np.int64) for i in range(3))
        ix, iy, iz = np.where(active, ix, 0), np.where(active, iy, 0), np.where(active, iz, 0)

        def at(dz, dy, dx):
            return val[iz + dz, iy + dy, ix + dx]
        lerp = lambda a, b, t: (a + t * (b - a)).astype(f32)
        fx, fy, fz = fr[..., 0], fr[..., 1], fr[..., 2]
        with np.errstate(invalid="ignore"):
            sdf = lerp(lerp(lerp(at(0, 0, 0), at(0, 0, 1), fx), lerp(at(0, 1, 0), at(0, 1, 1), fx), fy),
                       lerp(lerp(at(1, 0, 0), at(1, 0, 1), fx), lerp(at(1, 1, 0), at(1, 1, 1), fx), fy), fz)
            stop = active & (sdf <= 0)
            delta = np.where(sdf > 0, np.maximum(sdf, min_delta), f32(tr)).astype(f32)
        go = active & ~stop
        n += active
        steps[k] = np.where(go, delta, 0)
        with np.errstate(invalid="ignore"):
            kinds[k] = np.where(go, np.where(np.isnan(sdf) | (sdf == f32(tr)), 2, np.where(sdf <= min_delta, 1, 3)), 0)
        lam = np.where(go, lam + delta, lam).astype(f32)
        last = np.where(go, sdf, last)
        active = go & (lam < lam_max)
    return steps[:n.max() + 1], kinds[:n.max() + 1], n


def team_round_trips(steps_ray, T):
    """Round trips a team of T lanes needs for one ray's step sequence: each round samples up to T positions assuming every
    step repeats the one before the round's first sample; accepted = the prefix whose assumed steps equal the real ones."""
    n = len(steps_ray) + 1          # samples = steps + the stopping sample
    if T <= 1 or n <= 1:
        return n
    rounds, i, prev = 0, 0, None
    while i < n:
        rounds += 1
        # sample i is always valid; sample i + j (j >= 1) is valid if steps i .. i + j - 1 all equal the predicted step
        pred = steps_ray[i - 1] if i > 0 else None
        j = 1
        while j < T and i + j < n and pred is not None and i + j - 1 < len(steps_ray) and steps_ray[i + j - 1] == pred:
            j += 1
        i += j
    return rounds


def main():
    N, w, h = int(os.environ.get("N", 256)), 640, 480
    out = {}
    for scene in ("room",):
        vol, K, tr, near, far = fuse_stream(scene, N, w, h, 34)
        T_wc = scenes.orbit_pose(4, 30)
        steps, kinds, n = march_all(vol, K, tr, near, far, T_wc, w, h)
        marching = n > 0
        res = {"volume": N, "rays_marching": int(marching.sum()), "samples_mean": round(float(n[marching].mean()), 1), "samples_max": int(n.max())}
        long_rays = n >= np.percentile(n[marching], 99)
        for name, sel in (("all rays", marching), ("longest 1 % of the rays", long_rays)):
            k = kinds[:, sel]
            tot = (k > 0).sum()
            res[name] = {"steps": int(tot), "min_delta": round(float((k == 1).sum() / tot), 3), "trunc_or_nan": round(float((k == 2).sum() / tot), 3),
                         "data_dependent": round(float((k == 3).sum() / tot), 3)}
            s = steps[:, sel]
            rep = ((s[1:] == s[:-1]) & (s[1:] > 0)).sum() / max(1, (s[1:] > 0).sum())
            res[name]["repeats_previous_step"] = round(float(rep), 3)
        # per 32 x 2 wave tile: critical path in round trips, plain and with teams
        crit_plain, crit_team = [], []
        for ty in range(0, h, 2):
            for tx in range(0, w, 32):
                nn = n[ty:ty + 2, tx:tx + 32].reshape(-1)
                if nn.max() == 0:
                    continue
                st = steps[:, ty:ty + 2, tx:tx + 32].reshape(steps.shape[0], -1)
                crit_plain.append(int(nn.max()))
                # rays sorted by length; while a rays are active the team size is min(8, 64 // a): evaluate each ray's round trips with the
                # team size it has when it is among the last `a` rays (optimistic: applies that T to its whole remaining chain)
                order = np.argsort(-nn)
                worst = 0
                for rank, r in enumerate(order[:8]):
                    a = int((nn >= nn[r]).sum())          # rays still active when this one ends
                    # first part of the chain runs while more rays are active: assume no teams until only 16 rays are left
                    n16 = int(np.sort(nn)[-17]) if (nn > 0).sum() > 16 else 0   # samples after which at most 16 rays remain
                    seq = st[:nn[r] - 1, r] if nn[r] > 1 else np.zeros(0, f32)
                    head = min(n16, nn[r])
                    tail = seq[head:] if head < len(seq) else np.zeros(0, f32)
                    worst = max(worst, head + team_round_trips(list(tail), 4 if a > 8 else 8))
                crit_team.append(worst)
        cp, ct = np.array(crit_plain), np.array(crit_team)
        res["wave_tiles"] = len(cp)
        res["critical_path_round_trips"] = {"plain_max": int(cp.max()), "plain_p99": float(np.percentile(cp, 99)), "plain_mean": round(float(cp.mean()), 1),
                                            "teams_max": int(ct.max()), "teams_p99": float(np.percentile(ct, 99)), "teams_mean": round(float(ct.mean()), 1)}
        out[scene] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
