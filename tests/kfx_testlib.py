"""Shared helpers for the parity tests: build a scenario with the CPU oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import oracle  # noqa: E402  (tests may use the oracle; the product never does)
from kangaroo_amd import scenes  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def nan_equal(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.array_equal(a, b, equal_nan=True))


def mismatch_report(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    both_nan = np.isnan(a) & np.isnan(b)
    bad = ~((a == b) | both_nan)
    n = int(bad.sum())
    if n == 0:
        return "identical"
    with np.errstate(invalid="ignore"):
        d = np.abs(a - b)[bad]
    idx = np.argwhere(bad)[:5].tolist()
    return "%d/%d differ, max|d|=%s, first at %s, a=%s b=%s" % (
        n, a.size, np.nanmax(d) if np.isfinite(d).any() else "nan-mismatch", idx,
        a[bad][:5].tolist(), b[bad][:5].tolist())


def preprocess_oracle(depth_np, K, bil=scenes.BILATERAL):
    """Bilateral -> DepthToVbo -> NormalsFromVbo with the oracle; returns oracle Images."""
    h, w = depth_np.shape
    d = oracle.Image.from_numpy(depth_np)
    f = oracle.Image(w, h)
    vbo = oracle.Image(w, h, channels=4)
    nrm = oracle.Image(w, h, channels=4)
    oracle.bilateral(f, d, bil["gs"], bil["gr"], bil["size"], bil["minval"])
    oracle.depth_to_vbo(vbo, f, K)
    oracle.normals_from_vbo(nrm, vbo)
    return f, vbo, nrm


def make_volume(N, scene, dims=None, pitch_bytes=None):
    bmin, bmax, near, far = scenes.SCENES[scene]
    dims = dims or (N, N, N)
    vol = oracle.Volume(dims[0], dims[1], dims[2], bmin, bmax, pitch_bytes=pitch_bytes)
    oracle.sdf_reset(vol, float("nan"))
    return vol


def fuse_frames_oracle(vol, scene, w, h, n_frames, n_orbit=8, full_extent=False, nthreads=0, noise_sigma=0.0):
    """Fuse n_frames of the orbit trajectory; returns list of per-frame dicts.  noise_sigma > 0: SURVEY 8(d)'s noisy input -- Gaussian
    depth noise of that many metres per pixel, seed 1234 + frame (a fresh draw per frame, as a sensor's)."""
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(vol.boxmin, vol.boxmax, (vol.w, vol.h, vol.d))
    frames = []
    for i in range(n_frames):
        T_wc = scenes.orbit_pose(i, n_orbit)
        raw = scenes.render_depth(scene, w, h, T_wc, K, noise_sigma=noise_sigma, seed=1234 + i)
        f, vbo, nrm = preprocess_oracle(raw, K)
        T_cw = scenes.se3_inverse(T_wc)
        n = oracle.sdf_fuse(vol, f, nrm, T_cw, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA,
                            full_extent=full_extent, nthreads=nthreads)
        frames.append(dict(T_wc=T_wc, T_cw=T_cw, raw=raw, filtered=f.data.copy(), vbo=vbo.data.copy(),
                           normals=nrm.data.copy(), n_updated=n))
    return K, tr, frames


def upload_image(roo, arr):
    arr = np.asarray(arr)
    kind = {(np.dtype(np.float32), 2): "f32", (np.dtype(np.float32), 3): "f32x4",
            (np.dtype(np.uint16), 2): "u16", (np.dtype(np.uint8), 2): "u8"}[(arr.dtype, arr.ndim)]
    im = roo.Image(arr.shape[1], arr.shape[0], kind)
    im.MemcpyFromHost(arr)
    return im


def upload_volume(roo, ovol, pitch=None):
    v = roo.BoundedVolume(ovol.w, ovol.h, ovol.d, ovol.boxmin, ovol.boxmax, pitch=pitch)
    v.MemcpyFromHost(ovol.data)
    return v


def march_in_slabs(roo, vol, w, h, T_wc, K, near, far, tr, mode, tiles=4, world=8, ghost=2):
    """Render `vol` (a whole volume on the GPU) the way `world` ranks would: every rank THREAD holds a copy of its planes + `ghost` ghost
    planes and calls the C entry point of the multi-GPU raycast over the in-process transport (kfx_comm_create_threads): mode
    "exact" = kfx_slab_raycast_exact_tiled, "composite" = kfx_raycast_sdf + kfx_slab_composite_direct.  Returns rank 0's
    (depth, normals, shade) images after checking that every rank holds the same ones."""
    import ctypes as C
    import threading
    import torch
    from kangaroo_amd import _lib, slab
    L = slab._L()
    comms = slab.Comm.threads(world)
    ranks = []
    for r in range(world):
        lay = slab.layout(vol.d, float(vol.boxmin[2]), float(vol.boxmax[2]), r, world, ghost)
        lo = (vol.boxmin[0], vol.boxmin[1], lay.local_zmin)
        hi = (vol.boxmax[0], vol.boxmax[1], lay.local_zmax)
        local = roo.BoundedVolume(vol.w, vol.h, lay.s1 - lay.s0, lo, hi, pitch=vol.pitch)
        local.planes(0, local.d).copy_(vol.planes(lay.s0, lay.s1))
        imgs = (roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h))
        nbytes = L.kfx_slab_exact_tiled_scratch_bytes(w, h, tiles, world) if mode == "exact" else L.kfx_slab_composite_direct_scratch_bytes(w, h, world)
        scratch = torch.empty(int(nbytes), dtype=torch.uint8, device="cuda")
        ranks.append((lay, local, imgs, scratch))
    torch.cuda.synchronize()
    twc = np.ascontiguousarray(np.asarray(T_wc, np.float32)[:3].reshape(-1))
    kk = np.ascontiguousarray(np.asarray(K, np.float32))
    PF = _lib.PF
    status = [None] * world

    def work(r):
        lay, local, (d, n, i), scratch = ranks[r]
        if mode == "exact":
            steps = C.c_int(0)
            status[r] = L.kfx_slab_raycast_exact_tiled(d.ref(), n.ref(), i.ref(), C.c_void_p(scratch.data_ptr()), local.ref(), C.byref(lay), twc.ctypes.data_as(PF),
                                                       kk.ctypes.data_as(PF), near, far, tr, 1, tiles, comms[r].ref(), None, None, C.byref(steps))
        else:
            st = _lib.load().kfx_raycast_sdf(d.ref(), n.ref(), i.ref(), local.ref(), twc.ctypes.data_as(PF), kk.ctypes.data_as(PF), near, far, tr, 1, None)
            st2 = L.kfx_slab_composite_direct(d.ref(), n.ref(), i.ref(), C.c_void_p(scratch.data_ptr()), comms[r].ref(), None)
            status[r] = st or st2
        _lib.load().kfx_stream_synchronize(None)
    threads = [threading.Thread(target=work, args=(r,)) for r in range(1, world)]
    for t_ in threads:
        t_.start()
    work(0)
    for t_ in threads:
        t_.join()
    comms[0].destroy()
    assert status == [0] * world, status
    first = [x.MemcpyToHost() for x in ranks[0][2]]
    for r in range(1, world):
        for a, b in zip(first, ranks[r][2]):
            assert nan_equal(a, b.MemcpyToHost()), "rank %d holds other images than rank 0" % r
    return ranks[0][2]
