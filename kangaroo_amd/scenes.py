"""Synthetic cameras, scenes and trajectories for the KinectFusion hot path.

Deterministic inputs shared by the parity tests and bench.py.
Camera model follows the reference application's default
(applications/kinectfusion/main.cpp:63-64): fu = fv = 570.342 * w / 640,
u0 = w/2 - 0.5, v0 = h/2 - 0.5.  Depth is metres, float32, NaN = invalid.
Pure numpy: no GPU and no oracle dependency.
"""
import math

import numpy as np

# application defaults, applications/kinectfusion/main.cpp:149-158
BILATERAL = dict(gs=1.5, gr=0.1, size=3, minval=0.2)
TRUNC_DIST_FACTOR = 2.0
MAX_W = 1000.0
MIN_COS_THETA = 0.1

SCENES = {
    # name: (volume bbox min, max, raycast near, far)
    "room": ((-1.0, -1.0, 2.0), (1.0, 1.0, 4.0), 0.4, 8.0),
    "full": ((-1.0, -1.0, 4.0), (1.0, 1.0, 6.0), 0.4, 8.0),
}


def intrinsics(w, h):
    f = np.float32(w * 570.342 / 640.0)
    return np.array([f, f, w / 2.0 - 0.5, h / 2.0 - 0.5], dtype=np.float32)


def intrinsics_level(K, level):
    """ImageIntrinsics::operator[] (ImageIntrinsics.h:137-142)."""
    s = np.float32(1.0 / (1 << level))
    K = np.asarray(K, np.float32)
    return np.array([s * K[0], s * K[1], s * (K[2] + np.float32(0.5)) - np.float32(0.5),
                     s * (K[3] + np.float32(0.5)) - np.float32(0.5)], dtype=np.float32)


def trunc_dist(boxmin, boxmax, dims, factor=TRUNC_DIST_FACTOR):
    """trunc_dist_factor * length(vol.VoxelSizeUnits()), main.cpp:221."""
    sz = (np.asarray(boxmax, np.float32) - np.asarray(boxmin, np.float32))
    vs = sz / np.array([dims[0] - 1, dims[1] - 1, dims[2] - 1], np.float32)
    n = np.float32(np.sqrt(np.float32(vs[0] * vs[0] + vs[1] * vs[1]) + vs[2] * vs[2]))
    return float(np.float32(factor) * n)


def identity_pose():
    return np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], dtype=np.float32)


def orbit_pose(i, n=30, yaw_deg=5.0, trans=0.05):
    """Small orbit: +-yaw_deg yaw about y and +-trans m sideways; T_wc (camera->world)."""
    ph = 2.0 * math.pi * i / max(n, 1)
    yaw = math.radians(yaw_deg) * math.sin(ph)
    c, s = math.cos(yaw), math.sin(yaw)
    tx = trans * math.sin(ph)
    ty = 0.5 * trans * (1.0 - math.cos(ph)) - 0.5 * trans
    return np.array([[c, 0, s, tx], [0, 1, 0, ty], [-s, 0, c, 0.0]], dtype=np.float32)


def se3_inverse(T):
    """SE3inv (MatUtils.h:202-214) in float32."""
    T = np.asarray(T, np.float32).reshape(3, 4)
    R = T[:, :3].T.copy()
    t = -(R @ T[:, 3]).astype(np.float32)
    return np.concatenate([R, t[:, None]], axis=1).astype(np.float32)


def render_depth(scene, w, h, T_wc=None, K=None, noise_sigma=0.0, seed=1234):
    """Analytic z-depth (ray parameter with ray_c.z = 1) of the synthetic scenes."""
    K = intrinsics(w, h) if K is None else np.asarray(K, np.float32)
    T = identity_pose() if T_wc is None else np.asarray(T_wc, np.float32).reshape(3, 4)
    u = np.arange(w, dtype=np.float32)[None, :]
    v = np.arange(h, dtype=np.float32)[:, None]
    rc = np.stack(np.broadcast_arrays((u - K[2]) / K[0], (v - K[3]) / K[1], np.float32(1.0)), -1).astype(np.float32)
    rw = (rc @ T[:, :3].T).astype(np.float32)
    c = T[:, 3]
    best = np.full((h, w), np.inf, np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        if scene == "room":
            lo = np.array([-0.9, -0.9, -10.0], np.float32)
            hi = np.array([0.9, 0.9, 3.8], np.float32)
            a = (lo - c) / rw
            b = (hi - c) / rw
            texit = np.maximum(a, b).min(-1)
            best = np.where(texit > 0, texit, best)
            oc = (np.array([0, 0, 3.0], np.float32) - c)
            ldotc = (rw * oc).sum(-1)
            lsq = (rw * rw).sum(-1)
            csq = np.float32((oc * oc).sum())
            disc = ldotc * ldotc - lsq * (csq - np.float32(0.25))
            ts = (ldotc - np.sqrt(np.maximum(disc, 0))) / lsq
            hit = (disc >= 0) & (ts > 0) & (ts < best)
            best = np.where(hit, ts, best)
        elif scene == "full":
            t = (np.float32(5.95) - c[2]) / rw[..., 2]
            best = np.where(t > 0, t, best)
        else:
            raise ValueError(scene)
    d = np.where(np.isfinite(best), best, np.nan).astype(np.float32)
    if noise_sigma > 0:
        rng = np.random.default_rng(seed)
        d = (d + rng.normal(0.0, noise_sigma, d.shape).astype(np.float32)).astype(np.float32)
    return np.ascontiguousarray(d)
