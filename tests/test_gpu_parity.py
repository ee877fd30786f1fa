"""GPU parity tests: the HIP kernels (through the C ABI, via kangaroo_amd.roo) against the CPU
oracle on identical seeded inputs and against the committed golden fixtures.

Bar: bit-exact (NaN-aware) for SdfFuse, RaycastSdf, DepthToVbo, NormalsFromVbo, SdfReset,
SdfSphere -- the exact build uses IEEE fp32 without FMA contraction in the reference's operation
order.  BilateralFilter uses the hardware exp approximation (`__expf`, as the reference does,
cu_bilateral.cu:31-32), so it is compared with a stated tolerance of 2e-6 relative.
"""
import json
import os
import subprocess

import numpy as np
import pytest

import kfx_testlib as T
from kfx_testlib import oracle, scenes

pytestmark = pytest.mark.gpu

BILATERAL_RTOL = 2e-6


def load_golden(name):
    z = np.load(os.path.join(T.GOLDEN, name + ".npz"))
    return z, json.loads(str(z["meta"]))


# ---------------------------------------------------------------------------------
# golden fixtures (reference-header outputs)
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["room32_3frames", "full32_holes", "room_ragged_roi"])
def test_gpu_reproduces_golden_chain(roo, name):
    z, m = load_golden(name)
    dims = m["dims"]
    K = np.array(m["K"], np.float32)
    vol = roo.BoundedVolume(dims[0], dims[1], dims[2], m["boxmin"], m["boxmax"])
    roo.SdfReset(vol, float("nan"))
    for i in range(m["n_frames"]):
        f = T.upload_image(roo, z["filtered_%d" % i])
        nrm = T.upload_image(roo, z["normals_%d" % i])
        work = vol
        if "roi_frustum_%d" % i in z:
            fr = z["roi_frustum_%d" % i]
            lo, hi = roo.FitToFrustum(z["poses"][i], m["w"], m["h"], K, 2.2, 3.3)
            assert np.array_equal(np.concatenate([lo, hi]), fr)
            work = vol.SubBoundingVolume(lo, hi)
            assert [work.w, work.h, work.d] == z["roi_dims_%d" % i].tolist()
            assert np.array_equal(work.boxmin, z["roi_boxmin_%d" % i]) and np.array_equal(work.boxmax, z["roi_boxmax_%d" % i])
        roo.SdfFuse(work, f, nrm, scenes.se3_inverse(z["poses"][i]), K, m["trunc"], m["max_w"], m["mincostheta"])
    got = vol.MemcpyToHost()
    assert T.nan_equal(got, z["volume"]), T.mismatch_report(got, z["volume"])
    w, h = m["w"], m["h"]
    rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
    roo.RaycastSdf(rd, rn, ri, vol, z["poses"][-1], K, m["near"], m["far"], m["trunc"], m["subpix"])
    assert T.nan_equal(rd.MemcpyToHost(), z["ray_depth"]), T.mismatch_report(rd.MemcpyToHost(), z["ray_depth"])
    assert T.nan_equal(rn.MemcpyToHost(), z["ray_norm"]), T.mismatch_report(rn.MemcpyToHost(), z["ray_norm"])
    assert T.nan_equal(ri.MemcpyToHost(), z["ray_img"]), T.mismatch_report(ri.MemcpyToHost(), z["ray_img"])


def test_gpu_sphere_golden(roo):
    z, m = load_golden("sphere32_trunc0")
    N = m["dims"][0]
    vol = roo.BoundedVolume(N, N, N, m["boxmin"], m["boxmax"])
    roo.SdfReset(vol, float("nan"))
    roo.SdfSphere(vol, m["center"], m["r"])
    assert T.nan_equal(vol.MemcpyToHost(), z["volume"])
    w, h = m["w"], m["h"]
    rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
    roo.RaycastSdf(rd, rn, ri, vol, z["T_wc"], np.array(m["K"], np.float32), m["near"], m["far"], m["trunc"], True)
    assert T.nan_equal(rd.MemcpyToHost(), z["ray_depth"])
    assert T.nan_equal(rn.MemcpyToHost(), z["ray_norm"])
    assert T.nan_equal(ri.MemcpyToHost(), z["ray_img"])


def test_gpu_preprocess_golden(roo):
    z, m = load_golden("room32_3frames")
    K = np.array(m["K"], np.float32)
    b = m["bilateral"]
    for i in range(m["n_frames"]):
        raw = T.upload_image(roo, z["raw_%d" % i])
        f = roo.Image(m["w"], m["h"])
        roo.BilateralFilter(f, raw, b["gs"], b["gr"], b["size"], b["minval"])
        got = f.MemcpyToHost()
        exp = z["filtered_%d" % i]
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        ok = np.isfinite(exp)
        assert np.allclose(got[ok], exp[ok], rtol=BILATERAL_RTOL, atol=0), np.abs(got[ok] / exp[ok] - 1).max()
        # downstream ops are exact given identical inputs
        fexp = T.upload_image(roo, exp)
        vbo, nrm = roo.Image(m["w"], m["h"], "f32x4"), roo.Image(m["w"], m["h"], "f32x4")
        roo.DepthToVbo(vbo, fexp, K)
        roo.NormalsFromVbo(nrm, vbo)
        assert T.nan_equal(vbo.MemcpyToHost(), z["vbo_%d" % i])
        assert T.nan_equal(nrm.MemcpyToHost(), z["normals_%d" % i])


# ---------------------------------------------------------------------------------
# seeded oracle comparisons at sizes the oracle finishes in seconds
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("scene,N,w,h,frames", [("room", 64, 160, 120, 4), ("full", 64, 160, 120, 2),
                                                ("room", 128, 640, 480, 2),
                                                # pixels per voxel cross 1.3 inside the volume: SdfFuse splits into two
                                                # launches with different LDS tile capacities (fuse.hip, cap_for)
                                                ("room", 192, 400, 300, 2)])
def test_gpu_fuse_raycast_vs_oracle(roo, scene, N, w, h, frames):
    ovol = T.make_volume(N, scene)
    K, tr, fr = T.fuse_frames_oracle(ovol, scene, w, h, frames)
    bmin, bmax, near, far = scenes.SCENES[scene]
    vol = roo.BoundedVolume(N, N, N, bmin, bmax)
    roo.SdfReset(vol, float("nan"))
    for f in fr:
        gf, gn = T.upload_image(roo, f["filtered"]), T.upload_image(roo, f["normals"])
        # the diagnostics counter evaluates the same predicate as the oracle
        assert roo.SdfFuseCount(vol, gf, gn, f["T_cw"], K, tr, scenes.MIN_COS_THETA) == f["n_updated"]
        roo.SdfFuse(vol, gf, gn, f["T_cw"], K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
    got = vol.MemcpyToHost()
    assert T.nan_equal(got, ovol.data), T.mismatch_report(got, ovol.data)
    for subpix in (True, False):
        od, on, oi = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
        oracle.raycast_sdf(od, on, oi, ovol, fr[-1]["T_wc"], K, near, far, tr, subpix)
        rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
        roo.RaycastSdf(rd, rn, ri, vol, fr[-1]["T_wc"], K, near, far, tr, subpix)
        assert T.nan_equal(rd.MemcpyToHost(), od.data), T.mismatch_report(rd.MemcpyToHost(), od.data)
        assert T.nan_equal(rn.MemcpyToHost(), on.data), T.mismatch_report(rn.MemcpyToHost(), on.data)
        assert T.nan_equal(ri.MemcpyToHost(), oi.data), T.mismatch_report(ri.MemcpyToHost(), oi.data)
        # the march's diagnostics counters (bench.py's RaycastSdf roofline): samples, rays in the box, hits and distinct
        # voxels touched equal the oracle's bitmap count
        st, U = oracle.raycast_sdf_touch(od, on, oi, ovol, fr[-1]["T_wc"], K, near, far, tr, subpix)
        got = roo.RaycastSdfCount(vol, w, h, fr[-1]["T_wc"], K, near, far, tr, subpix)
        assert got == dict(samples=st["steps"], rays=st["rays"], hits=st["hits"], U=U), (got, st, U)


def test_gpu_fuse_ragged_dims_and_padded_pitch(roo):
    """Quirk Q1 (no tail beyond (dim/8)*8) and the opt-in full extent; odd pitches exercise the
    8-byte (one voxel per lane) kernel variant."""
    dims = (44, 37, 29)
    w, h = 96, 72
    for full in (False, True):
        for pitch in (None, dims[0] * 8 + 8, dims[0] * 8 + 64):
            ovol = oracle.Volume(dims[0], dims[1], dims[2], *scenes.SCENES["room"][:2], pitch_bytes=pitch)
            oracle.sdf_reset(ovol, float("nan"))
            K, tr, fr = T.fuse_frames_oracle(ovol, "room", w, h, 2, full_extent=full)
            vol = roo.BoundedVolume(dims[0], dims[1], dims[2], ovol.boxmin, ovol.boxmax, pitch=ovol.pitch)
            roo.SdfReset(vol, float("nan"))
            for f in fr:
                roo.SdfFuse(vol, T.upload_image(roo, f["filtered"]), T.upload_image(roo, f["normals"]), f["T_cw"], K,
                            tr, scenes.MAX_W, scenes.MIN_COS_THETA, full_extent=full)
            got = vol.MemcpyToHost()
            assert T.nan_equal(got, ovol.data), (full, pitch, T.mismatch_report(got, ovol.data))


def test_gpu_sub_volume_views(roo):
    """Fuse and raycast through offset, non-multiple-of-8 sub-volume views (application ROI path)."""
    N, w, h = 48, 96, 72
    ovol = T.make_volume(N, "room")
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(ovol.boxmin, ovol.boxmax, (N, N, N))
    vol = roo.BoundedVolume(N, N, N, ovol.boxmin, ovol.boxmax)
    roo.SdfReset(vol, float("nan"))
    for i in range(3):
        T_wc = scenes.orbit_pose(i, 8)
        f, vbo, nrm = T.preprocess_oracle(scenes.render_depth("room", w, h, T_wc, K), K)
        lo, hi = oracle.fit_to_frustum(T_wc, w, h, K, 2.3 + 0.1 * i, 3.4)
        osub = oracle.sub_bounding_volume(ovol, lo, hi)
        glo, ghi = roo.FitToFrustum(T_wc, w, h, K, 2.3 + 0.1 * i, 3.4)
        assert np.array_equal(lo, glo) and np.array_equal(hi, ghi)
        gsub = vol.SubBoundingVolume(glo, ghi)
        assert (gsub.w, gsub.h, gsub.d) == (osub.w, osub.h, osub.d)
        assert np.array_equal(gsub.boxmin, osub.boxmin) and np.array_equal(gsub.boxmax, osub.boxmax)
        T_cw = scenes.se3_inverse(T_wc)
        oracle.sdf_fuse(osub, f, nrm, T_cw, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
        roo.SdfFuse(gsub, T.upload_image(roo, f.data), T.upload_image(roo, nrm.data), T_cw, K, tr, scenes.MAX_W,
                    scenes.MIN_COS_THETA)
        od, on, oi = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
        oracle.raycast_sdf(od, on, oi, osub, T_wc, K, 0.4, 8.0, tr, True)
        rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
        roo.RaycastSdf(rd, rn, ri, gsub, T_wc, K, 0.4, 8.0, tr, True)
        assert T.nan_equal(rd.MemcpyToHost(), od.data) and T.nan_equal(rn.MemcpyToHost(), on.data)
        assert T.nan_equal(ri.MemcpyToHost(), oi.data)
    assert T.nan_equal(vol.MemcpyToHost(), ovol.data)


def test_gpu_behind_camera_and_inside_volume(roo):
    """Quirk Q2 (no Z>0 guard) and a camera inside the volume: must match the oracle exactly."""
    N, w, h = 40, 80, 60
    K = scenes.intrinsics(w, h)
    for bmin, bmax in (((-1, -1, -1.0), (1, 1, 4.0)), ((-1, -1, 0.5), (1, 1, 3.5))):
        ovol = oracle.Volume(N, N, N, bmin, bmax)
        oracle.sdf_reset(ovol, float("nan"))
        tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
        vol = roo.BoundedVolume(N, N, N, bmin, bmax)
        roo.SdfReset(vol, float("nan"))
        rng = np.random.default_rng(5)
        for i in range(2):
            T_wc = scenes.orbit_pose(i, 6, yaw_deg=20.0, trans=0.2)
            raw = scenes.render_depth("room", w, h, T_wc, K)
            f, vbo, nrm = T.preprocess_oracle(raw, K)
            nn = nrm.data.copy()
            nn[::5, ::3, :3] *= -1.0  # back-facing / noisy normals pass the costheta test behind the camera
            nn[1::7, 2::5, :3] += rng.normal(0, 0.3, nn[1::7, 2::5, :3].shape).astype(np.float32)
            nrm2 = oracle.Image.from_numpy(nn)
            T_cw = scenes.se3_inverse(T_wc)
            oracle.sdf_fuse(ovol, f, nrm2, T_cw, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
            roo.SdfFuse(vol, T.upload_image(roo, f.data), T.upload_image(roo, nn), T_cw, K, tr, scenes.MAX_W,
                        scenes.MIN_COS_THETA)
        got = vol.MemcpyToHost()
        assert T.nan_equal(got, ovol.data), T.mismatch_report(got, ovol.data)
        od, on, oi = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
        oracle.raycast_sdf(od, on, oi, ovol, T_wc, K, 0.1, 8.0, tr, True)
        rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
        roo.RaycastSdf(rd, rn, ri, vol, T_wc, K, 0.1, 8.0, tr, True)
        assert T.nan_equal(rd.MemcpyToHost(), od.data) and T.nan_equal(rn.MemcpyToHost(), on.data)
        assert T.nan_equal(ri.MemcpyToHost(), oi.data)


def test_gpu_weight_saturation(roo):
    """LimitWeight: many frames with a small max_w (Sdf.h:22-24)."""
    N, w, h = 32, 80, 60
    ovol = T.make_volume(N, "full")
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(ovol.boxmin, ovol.boxmax, (N, N, N))
    vol = roo.BoundedVolume(N, N, N, ovol.boxmin, ovol.boxmax)
    roo.SdfReset(vol, float("nan"))
    f, vbo, nrm = T.preprocess_oracle(scenes.render_depth("full", w, h, None, K), K)
    gf, gn = T.upload_image(roo, f.data), T.upload_image(roo, nrm.data)
    Tid = scenes.identity_pose()
    for _ in range(12):
        oracle.sdf_fuse(ovol, f, nrm, Tid, K, tr, 0.9, scenes.MIN_COS_THETA)
        roo.SdfFuse(vol, gf, gn, Tid, K, tr, 0.9, scenes.MIN_COS_THETA)
    got = vol.MemcpyToHost()
    assert T.nan_equal(got, ovol.data) and np.nanmax(got[..., 1]) == np.float32(0.9)


def test_gpu_reset_fills_padding(roo):
    vol = roo.BoundedVolume(10, 9, 8, pitch=10 * 8 + 48)
    vol.storage.zero_()
    roo.SdfReset(vol, 0.25)
    raw = vol.storage.cpu().numpy().view(np.float32)
    span = ((vol.d - 1) * vol.img_pitch + (vol.h - 1) * vol.pitch + vol.w * 8) // 4
    assert (raw[0:span:2] == 0.25).all() and (raw[1:span:2] == 0).all() and (raw[span:] == 0).all()
    sub = vol.SubVolume((1, 2, 3), (5, 4, 3))  # 8-byte aligned only: unaligned fill path
    roo.SdfReset(sub, -1.0)
    host = vol.MemcpyToHost()
    assert (host[3:5, 2:, 1:, 0] == -1.0).any() and host[0, 0, 0, 0] == 0.25


@pytest.mark.parametrize("kind,minval", [("f32", 0.2), ("f32", None), ("u16", 200), ("u8", None)])
@pytest.mark.parametrize("size", [1, 3, 5])
def test_gpu_bilateral_variants(roo, kind, minval, size):
    w, h = 70, 45  # not multiples of the tile
    rng = np.random.default_rng(11)
    base = scenes.render_depth("room", w, h)
    if kind == "f32":
        a = base.copy()
        a[5:9, 10:20] = np.nan
        a[20:22, :] = 0.05
        gr = 0.1
    elif kind == "u16":
        a = np.nan_to_num(base * 1000.0).astype(np.uint16)
        a[5:9, 10:20] = 0
        gr = 100.0
    else:
        a = rng.integers(0, 255, (h, w)).astype(np.uint8)
        gr = 30.0
    oin = oracle.Image.from_numpy(a)
    oout = oracle.Image(w, h)
    oracle.bilateral(oout, oin, 1.5, gr, size, minval)
    gout = roo.Image(w, h)
    roo.BilateralFilter(gout, T.upload_image(roo, a), 1.5, gr, size, minval)
    got, exp = gout.MemcpyToHost(), oout.data
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    ok = np.isfinite(exp)
    assert np.allclose(got[ok], exp[ok], rtol=BILATERAL_RTOL, atol=1e-30), np.abs(got[ok] / exp[ok] - 1).max()


def test_gpu_bilateral_large_window_fallback(roo):
    w, h = 40, 30
    a = scenes.render_depth("room", w, h)
    oout = oracle.Image(w, h)
    oracle.bilateral(oout, oracle.Image.from_numpy(a), 6.0, 0.1, 18, 0.2)
    gout = roo.Image(w, h)
    roo.BilateralFilter(gout, T.upload_image(roo, a), 6.0, 0.1, 18, 0.2)
    assert np.allclose(gout.MemcpyToHost(), oout.data, rtol=1e-5, equal_nan=True)


def test_gpu_depth_to_vbo_u16_and_pyramid_levels(roo):
    w, h = 64, 48
    K = scenes.intrinsics(640, 480)
    d16 = (np.nan_to_num(scenes.render_depth("room", w, h)) * 1000).astype(np.uint16)
    for level in (0, 3):
        Kl = scenes.intrinsics_level(K, level)
        ov = oracle.Image(w, h, channels=4)
        oracle.depth_to_vbo(ov, oracle.Image.from_numpy(d16), Kl, 0.001)
        gv = roo.Image(w, h, "f32x4")
        roo.DepthToVbo(gv, T.upload_image(roo, d16), Kl, 0.001)
        assert T.nan_equal(gv.MemcpyToHost(), ov.data)


def test_gpu_empty_and_tiny_inputs(roo):
    # zero-sized outputs are a no-op, like an empty CUDA grid
    e = roo.Image(0, 0)
    roo.NormalsFromVbo(roo.Image(0, 0, "f32x4"), roo.Image(0, 0, "f32x4"))
    roo.BilateralFilter(e, e, 1.5, 0.1, 3, 0.2)
    # volume smaller than one 8^3 block: reference launches an empty grid
    vol = roo.BoundedVolume(7, 7, 7)
    roo.SdfReset(vol, 1.0)
    d, n = roo.Image(16, 16), roo.Image(16, 16, "f32x4")
    roo.SdfFuse(vol, d, n, scenes.identity_pose(), scenes.intrinsics(16, 16), 0.1, 100.0, 0.1)
    assert (vol.MemcpyToHost()[..., 0] == 1.0).all()
    with pytest.raises(roo.KfxError):
        roo.SdfFuse(vol, roo.Image(2, 2), n, scenes.identity_pose(), scenes.intrinsics(16, 16), 0.1, 100.0, 0.1)


# ---------------------------------------------------------------------------------
# fast-math ("perf build") mode: stated tolerance instead of bit equality
# ---------------------------------------------------------------------------------
FAST_TSDF_TOL = 1e-4        # BASELINE.json north_star: TSDF L-inf < 1e-4 vs reference
FAST_FLIP_FRACTION = 2e-6   # voxels allowed to be classified differently (update / bilinear-cell flips)


def _fast_vs_oracle(roo, scene, N, w, h, frames):
    ovol = T.make_volume(N, scene)
    K, tr, fr = T.fuse_frames_oracle(ovol, scene, w, h, frames)
    bmin, bmax, near, far = scenes.SCENES[scene]
    vol = roo.BoundedVolume(N, N, N, bmin, bmax)
    roo.SdfReset(vol, float("nan"))
    prev = roo.set_math_mode("fast")
    try:
        assert roo.get_math_mode() == "fast"
        for f in fr:
            roo.SdfFuse(vol, T.upload_image(roo, f["filtered"]), T.upload_image(roo, f["normals"]), f["T_cw"], K, tr,
                        scenes.MAX_W, scenes.MIN_COS_THETA)
        got = vol.MemcpyToHost()
    finally:
        roo.set_math_mode(prev)
    exp = ovol.data
    nan_g, nan_e = np.isnan(got[..., 0]), np.isnan(exp[..., 0])
    flips = int((nan_g != nan_e).sum())                       # observed on one side only
    both = ~nan_g & ~nan_e
    dw = np.abs(got[..., 1][both] - exp[..., 1][both]) / np.maximum(np.abs(exp[..., 1][both]), 1e-12)
    same = dw <= 1e-3                                         # same set of frames updated the voxel on both sides
    dv = np.abs(got[..., 0][both] - exp[..., 0][both])
    return dict(n=N ** 3, flips=flips, history_flips=int((~same).sum()), linf=float(dv[same].max()),   # TRUE max, nothing filtered
                n_above_tol=int((dv[same] > FAST_TSDF_TOL).sum()), top5=np.sort(dv[same])[-5:].tolist(),
                linf_all_common=float(dv.max()), trunc=float(tr), w_rel=float(np.median(dw[same])))


@pytest.mark.parametrize("scene,N,w,h,frames", [("room", 64, 160, 120, 4), ("full", 64, 160, 120, 2),
                                                ("room", 128, 640, 480, 3)])
def test_gpu_fast_mode_within_tolerance(roo, scene, N, w, h, frames):
    """KFX_MATH_FAST SdfFuse (rcp/rsq/FMA) on the oracle's filtered images vs the exact oracle: TSDF values within 1e-4 on
    EVERY identically classified voxel (the true maximum; no exception list); the voxels classified differently (observed
    on one side only, or updated by a different set of frames: predicate / bilinear-cell boundaries) are counted, must stay
    negligible, and can differ by at most the clamp range.  The whole chain from raw depth at the benchmarked sizes is
    tests/test_gpu_chain.py."""
    r = _fast_vs_oracle(roo, scene, N, w, h, frames)
    budget = max(3, int(FAST_FLIP_FRACTION * r["n"] * frames))
    assert r["flips"] + r["history_flips"] <= budget, r
    assert r["linf"] < FAST_TSDF_TOL and r["n_above_tol"] == 0, r
    assert r["linf_all_common"] <= 2 * r["trunc"] * (1 + 1e-6), r


def test_gpu_fast_mode_unaligned_and_default_restored(roo):
    """The fast generic kernel (8-byte cells, odd pitch) agrees with the fast tiled kernel's
    tolerance, and the default mode is exact again afterwards."""
    dims, w, h = (44, 40, 32), 96, 72
    ovol = oracle.Volume(dims[0], dims[1], dims[2], *scenes.SCENES["room"][:2], pitch_bytes=dims[0] * 8 + 8)
    oracle.sdf_reset(ovol, float("nan"))
    K, tr, fr = T.fuse_frames_oracle(ovol, "room", w, h, 2)
    vol = roo.BoundedVolume(dims[0], dims[1], dims[2], ovol.boxmin, ovol.boxmax, pitch=ovol.pitch)
    roo.SdfReset(vol, float("nan"))
    prev = roo.set_math_mode("fast")
    try:
        for f in fr:
            roo.SdfFuse(vol, T.upload_image(roo, f["filtered"]), T.upload_image(roo, f["normals"]), f["T_cw"], K, tr,
                        scenes.MAX_W, scenes.MIN_COS_THETA)
    finally:
        roo.set_math_mode(prev)
    assert roo.get_math_mode() == "exact"
    got, exp = vol.MemcpyToHost(), ovol.data
    same = np.isnan(got[..., 0]) == np.isnan(exp[..., 0])
    assert (~same).sum() <= 3
    both = ~np.isnan(got[..., 0]) & ~np.isnan(exp[..., 0])
    d = np.abs(got[..., 0][both] - exp[..., 0][both])
    assert (d > FAST_TSDF_TOL).sum() <= 3 and np.median(d) < 1e-7


# ---------------------------------------------------------------------------------
# full-size (BASELINE config C2) size-independent properties
# ---------------------------------------------------------------------------------
def test_gpu_full_size_properties(roo):
    """512^3 / 640x480: (1) Z-slab decomposition is bit-identical to the monolithic fuse (each
    voxel is independent); (2) fusing the same frame twice doubles the weight and keeps
    the value; (3) fuse -> raycast reproduces the input depth to sub-voxel accuracy; (4) updated
    count on S_full is every voxel."""
    import torch
    N, w, h = 512, 640, 480
    K = scenes.intrinsics(w, h)
    for scene in ("full", "room"):
        bmin, bmax, near, far = scenes.SCENES[scene]
        tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
        raw = scenes.render_depth(scene, w, h, None, K)
        graw = T.upload_image(roo, raw)
        f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
        roo.BilateralFilter(f, graw, **scenes.BILATERAL)
        roo.DepthToVbo(vbo, f, K)
        roo.NormalsFromVbo(nrm, vbo)
        Tid = scenes.identity_pose()
        vol = roo.BoundedVolume(N, N, N, bmin, bmax)
        roo.SdfReset(vol, float("nan"))
        roo.SdfFuse(vol, f, nrm, Tid, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
        a = vol.tensor().clone()
        updated = int((~torch.isnan(a[..., 0])).sum())
        if scene == "full":
            # everything in front of the wall (z = 5.95) plus one truncation band behind it
            assert 0.97 * N ** 3 < updated < 0.995 * N ** 3
        else:
            assert 0.3 * N ** 3 < updated < 0.9 * N ** 3
        # (1) slabs: through the slab entry point (voxel positions by the full volume's expression) the
        #     decomposition is BIT-IDENTICAL to the monolithic fuse
        vol2 = roo.BoundedVolume(N, N, N, bmin, bmax)
        roo.SdfReset(vol2, float("nan"))
        for z0 in range(0, N, 64):
            roo.SdfFuse(vol2.ZSlab(z0, z0 + 64), f, nrm, Tid, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA,
                        slab=(N, z0, bmin[2], bmax[2]))
        b = vol2.tensor()
        assert bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())
        # ... while plain sub-volume views (bbox recomputed from voxel positions, as SubBoundingVolume does)
        # differ by a 1-ulp position change: same classification almost everywhere, values within 1e-4
        roo.SdfReset(vol2, float("nan"))
        for z0 in range(0, N, 64):
            roo.SdfFuse(vol2.ZSlab(z0, z0 + 64), f, nrm, Tid, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
        same_class = (torch.isnan(a[..., 0]) == torch.isnan(b[..., 0])).float().mean().item()
        assert same_class > 0.9999
        both = ~torch.isnan(a[..., 0]) & ~torch.isnan(b[..., 0])
        assert (a[..., 0][both] - b[..., 0][both]).abs().max().item() < 1e-4  # north-star TSDF tolerance
        del vol2, b
        # (2) idempotent value, doubled weight
        roo.SdfFuse(vol, f, nrm, Tid, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
        c = vol.tensor()
        m = ~torch.isnan(a[..., 0])
        assert torch.allclose(c[..., 1][m], 2 * a[..., 1][m], rtol=1e-6)
        assert torch.allclose(c[..., 0][m], a[..., 0][m], atol=1e-6)
        # (3) raycast reproduces the measured depth
        rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
        roo.RaycastSdf(rd, rn, ri, vol, Tid, K, near, far, tr, True)
        d = rd.MemcpyToHost()
        filt = f.MemcpyToHost()
        err = np.abs(d - filt)
        ok = np.isfinite(err)
        voxel = (bmax[0] - bmin[0]) / (N - 1)
        assert ok.sum() > (0.05 if scene == "full" else 0.5) * w * h
        assert np.median(err[ok]) < 0.5 * voxel
        del vol, a, c
        torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------
# multi-GPU plumbing on one GPU: the slab pipeline's device-side tensor code and RCCL calls
# ---------------------------------------------------------------------------------
def test_gpu_slab_pipeline_single_rank_rccl(roo):
    """World-size-1 RCCL group on the single GPU: exercises SlabPipeline's device tensor code
    (key packing, MIN / SUM all-reduces, halo plane views) with the real HIP operators; with one
    rank the composite must reproduce the local raycast exactly."""
    import torch
    import torch.distributed as dist
    from kangaroo_amd.pipeline import FramePipeline, SlabPipeline
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        N, w, h = 64, 160, 120
        bmin, bmax, near, far = scenes.SCENES["room"]
        slab = SlabPipeline(roo, dist, (N, N, N), bmin, bmax, w, h, halo="exchange", raycast="composite", near=near, far=far)
        mono = FramePipeline(roo, (N, N, N), bmin, bmax, w, h, near=near, far=far)
        for i in range(2):
            T_wc = scenes.orbit_pose(i, 8)
            raw = scenes.render_depth("room", w, h, T_wc, slab.K)
            slab.raw.MemcpyFromHost(raw)
            mono.raw.MemcpyFromHost(raw)
            slab.step(T_wc)
            mono.step(T_wc)
        assert T.nan_equal(slab.vol.MemcpyToHost(), mono.vol.MemcpyToHost())
        before = (slab.ray_d.MemcpyToHost(), slab.ray_n.MemcpyToHost(), slab.ray_i.MemcpyToHost())
        assert T.nan_equal(before[0], mono.ray_d.MemcpyToHost())
        slab.composite()  # one rank: the winner is always rank 0
        torch.cuda.synchronize()
        assert T.nan_equal(slab.ray_d.MemcpyToHost(), before[0])
        assert T.nan_equal(slab.ray_n.MemcpyToHost(), before[1]) and T.nan_equal(slab.ray_i.MemcpyToHost(), before[2])
        assert slab.vol.planes(1, 3).numel() == 2 * slab.vol.img_pitch
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------
# fp16 TSDF cells (BASELINE config C5)
# ---------------------------------------------------------------------------------
def _fuse_frames_half(roo, N, scene, w, h, frames, mode="exact"):
    bmin, bmax, near, far = scenes.SCENES[scene]
    ovol = oracle.VolumeH(N, N, N, bmin, bmax)
    oracle.sdf_reset(ovol, float("nan"))
    vol = roo.BoundedVolume(N, N, N, bmin, bmax, kind="f16")
    roo.SdfReset(vol, float("nan"))
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    prev = roo.set_math_mode(mode)
    try:
        for i in range(frames):
            T_wc = scenes.orbit_pose(i, 8)
            f, vbo, nrm = T.preprocess_oracle(scenes.render_depth(scene, w, h, T_wc, K), K)
            T_cw = scenes.se3_inverse(T_wc)
            oracle.sdf_fuse(ovol, f, nrm, T_cw, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
            roo.SdfFuse(vol, T.upload_image(roo, f.data), T.upload_image(roo, nrm.data), T_cw, K, tr, scenes.MAX_W,
                        scenes.MIN_COS_THETA)
    finally:
        roo.set_math_mode(prev)
    return ovol, vol, K, tr, T_wc, near, far


@pytest.mark.parametrize("scene,N,w,h,frames", [("room", 64, 160, 120, 4), ("full", 64, 160, 120, 3)])
def test_gpu_half_cells_fuse_raycast_exact(roo, scene, N, w, h, frames):
    """SDF_h volumes: fuse (rounding every intermediate to half) and raycast are bit-identical to the oracle."""
    ovol, vol, K, tr, T_wc, near, far = _fuse_frames_half(roo, N, scene, w, h, frames)
    got = vol.MemcpyToHost()
    assert got.dtype == np.float16 and vol.pitch >= N * 4
    assert T.nan_equal(got.view(np.uint16)[~np.isnan(got)], ovol.data.view(np.uint16)[~np.isnan(ovol.data)])
    assert np.array_equal(np.isnan(got), np.isnan(ovol.data))
    od, on, oi = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
    st = oracle.raycast_sdf(od, on, oi, ovol, T_wc, K, near, far, tr, True)
    rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
    roo.RaycastSdf(rd, rn, ri, vol, T_wc, K, near, far, tr, True)
    assert T.nan_equal(rd.MemcpyToHost(), od.data) and T.nan_equal(rn.MemcpyToHost(), on.data)
    assert T.nan_equal(ri.MemcpyToHost(), oi.data)
    assert st["hits"] > (0.3 if scene == "room" else 0.02) * w * h


def test_gpu_half_cells_sphere_and_accuracy_vs_fp32(roo):
    """SdfSphere on half cells matches the oracle; a half-cell fuse stays within half precision of
    the fp32 volume (accuracy report of config C5: |val| <= trunc, half ulp <= trunc * 2^-11)."""
    N = 32
    ov = oracle.VolumeH(N, N, N)
    oracle.sdf_reset(ov, float("nan"))
    oracle.sdf_sphere(ov, (0.05, -0.1, 0.0), 0.7)
    gv = roo.BoundedVolume(N, N, N, kind="f16")
    roo.SdfReset(gv, float("nan"))
    roo.SdfSphere(gv, (0.05, -0.1, 0.0), 0.7)
    assert T.nan_equal(gv.MemcpyToHost(), ov.data)
    # fp16 vs fp32 fuse of the same frames
    oh, vh, K, tr, T_wc, near, far = _fuse_frames_half(roo, 64, "room", 160, 120, 3)
    o32 = T.make_volume(64, "room")
    T.fuse_frames_oracle(o32, "room", 160, 120, 3)
    a, b = vh.MemcpyToHost().astype(np.float32), o32.data
    both = ~np.isnan(a[..., 0]) & ~np.isnan(b[..., 0])
    assert (np.isnan(a[..., 0]) == np.isnan(b[..., 0])).mean() > 0.999
    assert np.abs(a[..., 0][both] - b[..., 0][both]).max() < 4 * tr * 2.0 ** -11 + 1e-6


def test_gpu_half_cells_fast_mode(roo):
    ovol, vol, K, tr, T_wc, near, far = _fuse_frames_half(roo, 64, "room", 160, 120, 3, mode="fast")
    a, b = vol.MemcpyToHost().astype(np.float32), ovol.data.astype(np.float32)
    assert (np.isnan(a[..., 0]) != np.isnan(b[..., 0])).sum() <= 3
    both = ~np.isnan(a[..., 0]) & ~np.isnan(b[..., 0])
    # results differ by one half ulp where the fp32 intermediate sits on a half rounding boundary
    # (ulp of a value near trunc is trunc * 2^-10): rare, and never more than two ulps apart
    d = np.abs(a[..., 0][both] - b[..., 0][both])
    assert (d > 0).mean() < 0.01 and (d > tr * 2.0 ** -9).sum() <= 3


# ---------------------------------------------------------------------------------
# frame pre-amble: mm -> m conversion and the NaN-aware depth pyramid
# ---------------------------------------------------------------------------------
def test_gpu_scale_bias_and_pyramid(roo):
    w, h = 160, 120
    mm = scenes.render_depth("room", w, h) * np.float32(1000.0)
    mm[10:14, 20:40] = np.nan
    mm[50, :] = np.nan
    om, omm = oracle.Image(w, h), oracle.Image.from_numpy(mm)
    oracle.elementwise_scale_bias(om, omm, 1.0 / 1000.0, 0.0)
    pyr = roo.Pyramid(w, h, 4)
    gmm = T.upload_image(roo, mm)
    roo.ElementwiseScaleBias(pyr[0], gmm, 1.0 / 1000.0)
    assert T.nan_equal(pyr[0].MemcpyToHost(), om.data)
    roo.ElementwiseScaleBias(gmm, gmm, 0.5, 3.0)  # in place, with a bias
    exp = oracle.Image(w, h)
    oracle.elementwise_scale_bias(exp, omm, 0.5, 3.0)
    assert T.nan_equal(gmm.MemcpyToHost(), exp.data)
    roo.BoxReduceIgnoreInvalid(pyr)
    prev = om
    for l in range(1, 4):
        nxt = oracle.Image(w >> l, h >> l)
        oracle.box_half_ignore_invalid(nxt, prev)
        got = pyr[l].MemcpyToHost()
        assert got.shape == (h >> l, w >> l) and T.nan_equal(got, nxt.data), l
        prev = nxt
    assert np.isnan(pyr[1].MemcpyToHost()[5:7, 10:20]).all()  # fully invalid 2x2 blocks stay invalid


@pytest.mark.parametrize("w,h,levels", [(640, 480, 4), (160, 120, 4), (100, 76, 3), (67, 45, 4), (33, 31, 2), (40, 40, 1)])
def test_gpu_depth_pyramid_with_maps_in_one_launch(roo, w, h, levels):
    """kfx_depth_pyramid_vbo_normals_f32: BoxReduceIgnoreInvalid + DepthToVbo + NormalsFromVbo on every level as ONE launch against
    the 2 * levels - 1 separate launches: depth levels bit for bit, vertex and normal maps bit for bit wherever they are numbers and
    NaN in the same places (which operand's NaN an instruction hands on is the compiler's choice per kernel) -- sizes that are no
    multiple of the 32-pixel tile, odd sizes (a level's last row / column of 2 x 2 blocks is cut), NaN holes and stripes."""
    rng = np.random.default_rng(w * 1000 + h)
    d0 = scenes.render_depth("room", w, h).astype(np.float32)
    d0[rng.random((h, w)) < 0.05] = np.nan
    d0[h // 3, :] = np.nan
    d0[:, w // 2: w // 2 + 3] = np.nan
    d0[-1, -1] = np.inf
    K = scenes.intrinsics(w, h)
    K_levels = [np.array([K[0] / (1 << l), K[1] / (1 << l), (K[2] + 0.5) / (1 << l) - 0.5, (K[3] + 0.5) / (1 << l) - 0.5], np.float32) for l in range(levels)]
    pyr = [roo.Pyramid(w, h, levels) for _ in range(2)]
    maps = [[[roo.Image(w >> l, h >> l, "f32x4") for l in range(levels)] for _ in range(2)] for _ in range(2)]   # [variant][vbo|nrm][level]
    for v in range(2):
        pyr[v][0].MemcpyFromHost(d0)
        for l in range(1, levels):   # what a launch leaves alone stays recognisable
            pyr[v][l].MemcpyFromHost(np.full((h >> l, w >> l), -7.0, np.float32))
    roo.BoxReduceIgnoreInvalid(pyr[0])
    for l in range(levels):
        roo.DepthToVbo(maps[0][0][l], pyr[0][l], K_levels[l])
        roo.NormalsFromVbo(maps[0][1][l], maps[0][0][l])
    roo.DepthPyramidVboNormals(pyr[1], maps[1][0], maps[1][1], K_levels)
    for l in range(levels):
        a, b = pyr[0][l].MemcpyToHost(), pyr[1][l].MemcpyToHost()
        assert a.shape == (h >> l, w >> l) and a.tobytes() == b.tobytes(), ("depth", l, T.mismatch_report(b, a))
        for k, name in ((0, "vbo"), (1, "normals")):
            a, b = maps[0][k][l].MemcpyToHost(), maps[1][k][l].MemcpyToHost()
            assert T.nan_equal(b, a), (name, l, T.mismatch_report(b, a))
    assert np.isfinite(maps[1][1][0].MemcpyToHost()[..., :3]).any()


def test_gpu_composite_kernels_match_tensor_expressions(roo):
    """The fused pack / select / unpack kernels of the multi-GPU raycast composite, with the two
    all-reduces emulated on one GPU (elementwise min / sum over two 'ranks'), against the plain tensor
    expressions SlabPipeline.composite falls back to -- and against a single-volume raycast."""
    import torch
    N, w, h = 64, 160, 120
    ovol = T.make_volume(N, "room")
    K, tr, fr = T.fuse_frames_oracle(ovol, "room", w, h, 2)
    vol = T.upload_volume(roo, ovol)
    bmin, bmax, near, far = scenes.SCENES["room"]
    T_wc = fr[-1]["T_wc"]
    ranks = []
    for (z0, z1) in ((0, 34), (30, 64)):  # two overlapping slabs
        rd, rn, ri = roo.Image(w, h, pitch=w * 4), roo.Image(w, h, "f32x4", pitch=w * 16), roo.Image(w, h, pitch=w * 4)
        roo.RaycastSdf(rd, rn, ri, vol.ZSlab(z0, z1), T_wc, K, near, far, tr, True)
        ranks.append((rd, rn, ri))
    # tensor-expression reference (what pipeline.composite does without the fused kernels)
    keys, bits_l = [], []
    for r, (rd, rn, ri) in enumerate(ranks):
        d = rd.tensor()
        hit = torch.isfinite(d)
        bits = torch.where(hit, d, torch.full_like(d, float("inf"))).contiguous().view(torch.int32).to(torch.int64)
        keys.append((bits << 8) | r)
        bits_l.append((hit, bits))
    key_ref = torch.minimum(keys[0], keys[1])
    pay_ref = torch.zeros((h, w, 4), device=key_ref.device)   # {n.x, n.y, n.z, shade}: KFX_COMPOSITE_PAYLOAD floats per pixel
    for r, (rd, rn, ri) in enumerate(ranks):
        hit, bits = bits_l[r]
        mine = hit & ((key_ref & 0xFF) == r) & ((key_ref >> 8) == bits)
        pay_ref[..., 0:3] += torch.where(mine.unsqueeze(-1), rn.tensor()[..., 0:3], torch.zeros_like(rn.tensor()[..., 0:3]))
        pay_ref[..., 3] += torch.where(mine, ri.tensor(), torch.zeros_like(ri.tensor()))
    # fused kernels
    kbuf = [torch.empty(w * h, dtype=torch.int64, device="cuda") for _ in ranks]
    for r, (rd, rn, ri) in enumerate(ranks):
        roo.CompositePack(rd, rn, ri, kbuf[r], r)
    key = torch.minimum(kbuf[0], kbuf[1])
    assert torch.equal(key.view(h, w), key_ref)
    pbuf = [torch.empty(w * h * 4, dtype=torch.float32, device="cuda") for _ in ranks]
    for r, (rd, rn, ri) in enumerate(ranks):
        roo.CompositeSelect(rd, rn, ri, key, pbuf[r], r)
    pay = pbuf[0] + pbuf[1]
    assert torch.equal(pay.view(h, w, 4), pay_ref)
    rd, rn, ri = ranks[0]
    roo.CompositeUnpack(rd, rn, ri, key, pay)
    d = rd.MemcpyToHost()
    # merged image vs the single-volume raycast: same hits up to boundary pixels, depth within a fraction of a voxel
    od, on, oi = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
    oracle.raycast_sdf(od, on, oi, ovol, T_wc, K, near, far, tr, True)
    both = np.isfinite(d) & np.isfinite(od.data)
    assert (np.isfinite(d) != np.isfinite(od.data)).mean() < 0.01
    assert np.median(np.abs(d[both] - od.data[both])) < 0.05 * (bmax[0] - bmin[0]) / (N - 1)
    n = rn.MemcpyToHost()
    assert (n[~np.isfinite(d)] == 0).all() and (n[np.isfinite(d)][:, 3] == 1).all()


@pytest.mark.parametrize("world,w,h", [(2, 160, 120), (3, 161, 97), (8, 160, 120)])
def test_gpu_direct_send_composite_kernels(roo, world, w, h):
    """kfx_composite_strips_pack / merge / unpack (the direct-send merge), with the all-to-all and the all-gather emulated on one
    GPU over `world` overlapping slabs of one volume, against the key / payload composite of the kernels above: the same winner
    per pixel, the same images (image sizes that do not divide by the rank count: the last strip is padded)."""
    import torch
    from kangaroo_amd.pipeline import slab_range
    N = 64
    ovol = T.make_volume(N, "room")
    K, tr, fr = T.fuse_frames_oracle(ovol, "room", w, h, 2)
    vol = T.upload_volume(roo, ovol)
    bmin, bmax, near, far = scenes.SCENES["room"]
    T_wc = fr[-1]["T_wc"]
    dense = w % 64 == 0   # the odd-sized case runs on pitched images (rows padded by the allocator)
    mk = lambda: (roo.Image(w, h, pitch=w * 4 if dense else None), roo.Image(w, h, "f32x4", pitch=w * 16 if dense else None), roo.Image(w, h, pitch=w * 4 if dense else None))
    ranks = []
    for r in range(world):
        z0, z1 = slab_range(N, r, world)
        rd, rn, ri = mk()
        roo.RaycastSdf(rd, rn, ri, vol.ZSlab(max(z0 - 2, 0), min(z1 + 2, N)), T_wc, K, near, far, tr, True)
        ranks.append((rd, rn, ri))
    # the key / payload composite (MIN and SUM over the ranks emulated elementwise)
    kbuf = [torch.empty(w * h, dtype=torch.int64, device="cuda") for _ in ranks]
    for r, (rd, rn, ri) in enumerate(ranks):
        roo.CompositePack(rd, rn, ri, kbuf[r], r)
    key = torch.stack(kbuf).min(dim=0).values
    pbuf = [torch.empty(w * h * 4, dtype=torch.float32, device="cuda") for _ in ranks]
    for r, (rd, rn, ri) in enumerate(ranks):
        roo.CompositeSelect(rd, rn, ri, key, pbuf[r], r)
    pay = torch.stack(pbuf).sum(dim=0)
    wd, wn, wi = mk()
    roo.CompositeUnpack(wd, wn, wi, key, pay)
    want = [x.MemcpyToHost() for x in (wd, wn, wi)]
    # the direct-send merge
    S = roo.CompositeStripPixels(w, h, world)
    assert S % 64 == 0 and S * world >= w * h > (S - 64) * world
    send = [torch.full((world, roo.STRIP_PLANES, S), float("nan"), device="cuda") for _ in ranks]
    for r, (rd, rn, ri) in enumerate(ranks):
        roo.CompositeStripsPack(rd, rn, ri, send[r], world)
    full = torch.empty((world, roo.STRIP_PLANES, S), device="cuda")
    for j in range(world):   # rank j: receives strip j of every rank, merges, its result is slot j of the all-gather
        recv = torch.stack([send[r][j] for r in range(world)]).contiguous()
        merged = torch.empty((roo.STRIP_PLANES, S), device="cuda")
        roo.CompositeStripsMerge(recv, merged, S, world)
        full[j] = merged
    gd, gn, gi = mk()
    roo.CompositeStripsUnpack(gd, gn, gi, full, world)
    got = [x.MemcpyToHost() for x in (gd, gn, gi)]
    assert np.isfinite(got[0]).mean() > 0.3
    for a, b in zip(got, want):   # (values: a -0 of a winner's normal stays -0 here and becomes +0 in the sum)
        assert np.array_equal(a, b, equal_nan=True)
    # the winner is the nearest of the per-rank hits
    stack = np.stack([np.where(np.isfinite(rd.MemcpyToHost()), rd.MemcpyToHost(), np.inf) for rd, _, _ in ranks])
    assert np.array_equal(np.where(np.isfinite(got[0]), got[0], np.inf), stack.min(axis=0))


def test_gpu_fuzz_direct_send_composite_kernels(roo):
    """Random image sizes (1 x 1 upwards), rank counts 1-9, pitched and dense images, depths with ties between ranks, NaN / inf
    misses, -0 and denormal payloads: pack / merge / unpack against a numpy model of "nearest finite depth wins, the lowest
    rank on ties" -- bit for bit (the direct merge moves values, it does not add them)."""
    import torch
    rng = np.random.default_rng(20261002)
    for case in range(48):
        w, h = int(rng.integers(1, 200)), int(rng.integers(1, 120))
        world = int(rng.integers(1, 10))
        dense = bool(rng.integers(0, 2))
        ranks, host = [], []
        pool = rng.uniform(0.4, 8.0, size=8).astype(np.float32)   # few distinct depths: many ties
        for r in range(world):
            d = rng.choice(pool, size=(h, w)).astype(np.float32) if case % 3 == 0 else rng.uniform(0.4, 8.0, size=(h, w)).astype(np.float32)
            miss = rng.random((h, w)) < 0.4
            d[miss] = rng.choice(np.array([np.nan, np.inf], np.float32), size=int(miss.sum()))
            n = rng.standard_normal((h, w, 4)).astype(np.float32)
            n[..., 3] = 1.0
            n[rng.random((h, w)) < 0.05, 0] = -0.0
            n[rng.random((h, w)) < 0.05, 1] = np.float32(1e-41)   # denormal
            i = rng.random((h, w)).astype(np.float32)
            rd = roo.Image(w, h, pitch=w * 4 if dense else None)
            rn = roo.Image(w, h, "f32x4", pitch=w * 16 if dense else None)
            ri = roo.Image(w, h, pitch=w * 4 if dense else None)
            rd.MemcpyFromHost(d); rn.MemcpyFromHost(n); ri.MemcpyFromHost(i)
            ranks.append((rd, rn, ri))
            host.append((d, n, i))
        S = roo.CompositeStripPixels(w, h, world)
        send = [torch.full((world, roo.STRIP_PLANES, S), float("nan"), device="cuda") for _ in ranks]
        for r, (rd, rn, ri) in enumerate(ranks):
            roo.CompositeStripsPack(rd, rn, ri, send[r], world)
        full = torch.empty((world, roo.STRIP_PLANES, S), device="cuda")
        for j in range(world):
            recv = torch.stack([send[r][j] for r in range(world)]).contiguous()
            merged = torch.empty((roo.STRIP_PLANES, S), device="cuda")
            roo.CompositeStripsMerge(recv, merged, S, world)
            full[j] = merged
        gd, gn, gi = ranks[0]   # unpack over rank 0's own images, as the pipeline does
        roo.CompositeStripsUnpack(gd, gn, gi, full, world)
        got_d, got_n, got_i = gd.MemcpyToHost(), gn.MemcpyToHost(), gi.MemcpyToHost()
        # numpy model
        depth = np.stack([np.where(np.isfinite(d), d, np.inf) for d, _, _ in host])
        win = depth.argmin(axis=0)                     # first minimum = lowest rank on ties
        best = depth.min(axis=0)
        hit = np.isfinite(best)
        want_d = np.where(hit, best, np.nan).astype(np.float32)
        nn = np.stack([n for _, n, _ in host])
        ii = np.stack([i for _, _, i in host])
        yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        want_n = np.where(hit[..., None], nn[win, yy, xx], 0).astype(np.float32)
        want_n[..., 3] = hit.astype(np.float32)
        want_i = np.where(hit, ii[win, yy, xx], 0).astype(np.float32)
        tag = (case, w, h, world, dense)
        assert np.array_equal(got_d.view(np.uint32)[hit], want_d.view(np.uint32)[hit]) and np.isnan(got_d[~hit]).all(), tag
        assert np.array_equal(got_n.view(np.uint32), want_n.view(np.uint32)), tag
        assert np.array_equal(got_i.view(np.uint32), want_i.view(np.uint32)), tag


@pytest.mark.parametrize("world,ghost", [(2, 2), (4, 1), (5, 3)])
def test_gpu_exact_slab_raycast_rounds(roo, world, ghost):
    """SURVEY 8(e) exact variant: `world` slabs of one volume marched in rounds with the state merge of
    SlabPipeline.raycast_exact (emulated in-process: per-slab states, integer-sum merge of the touched
    pixels).  Every round of every slab is compared with the oracle's slab march on the same input state,
    and the final images must equal RaycastSdf on the whole volume bit for bit."""
    import torch
    from kangaroo_amd.pipeline import slab_range
    N, w, h, scene = 64, 160, 120, "room"
    ovol = T.make_volume(N, scene)
    K, tr, fr = T.fuse_frames_oracle(ovol, scene, w, h, 3)
    bmin, bmax, near, far = scenes.SCENES[scene]
    vol = T.upload_volume(roo, ovol)
    T_wc = fr[-1]["T_wc"]
    rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
    roo.RaycastSdf(rd, rn, ri, vol, T_wc, K, near, far, tr, True)
    want = (rd.MemcpyToHost(), rn.MemcpyToHost(), ri.MemcpyToHost())
    assert np.isfinite(want[0]).mean() > 0.3

    spans = [slab_range(N, r, world) for r in range(world)]
    stored = [(max(z0 - ghost, 0), min(z1 + ghost, N)) for z0, z1 in spans]
    views = [vol.ZSlab(s0, s1) for s0, s1 in stored]

    def oslab(s0, s1):  # planes [s0, s1) of the oracle volume as a non-owning view (x/y box of the parent)
        st = oracle.KfoVolume(ovol.pitch, ovol.raw.ctypes.data + s0 * ovol.img_pitch, ovol.w, ovol.h, ovol.img_pitch, s1 - s0)
        for i in range(3):
            st.boxmin[i], st.boxmax[i] = float(ovol.boxmin[i]), float(ovol.boxmax[i])
        return oracle.SubVolume(ovol, st)
    oviews = [oslab(s0, s1) for s0, s1 in stored]
    states = [torch.empty((9, h, w), dtype=torch.float32, device="cuda") for _ in range(world)]
    rounds = 0
    while True:
        for r in range(world):
            slab = (N, stored[r][0], float(bmin[2]), float(bmax[2]))
            ost = states[r].cpu().numpy().copy()
            roo.RaycastSdfSlab(states[r], rounds == 0, views[r], slab, spans[r][0], spans[r][1], w, h, T_wc, K, near, far, tr, True)
            oracle.raycast_sdf_slab(ost, rounds == 0, oviews[r], slab, spans[r][0], spans[r][1], w, h, T_wc, K, near, far, tr, True)
            got = states[r].cpu().numpy()
            assert T.nan_equal(got, ost), (rounds, r, T.mismatch_report(got, ost))
        rounds += 1
        march = [s[0:5].view(torch.int32) for s in states]
        total = torch.zeros_like(march[0])
        for m in march:
            total += torch.where((m[4] != 0).unsqueeze(0), m, torch.zeros_like(m))
        # one toucher per pixel and round: the touched plane sums to exactly one 1.0f or to zero
        assert bool(((total[4] == 0) | (total[4] == 0x3F800000)).all())
        for m in march:
            m.copy_(torch.where((total[4] != 0).unsqueeze(0), total, m))
        status = states[0][3]
        if not bool(((status == 0) | (status == 3)).any()):
            break
        assert rounds <= world + 3
    assert rounds > 1
    out = torch.zeros((4, h, w), dtype=torch.int32, device="cuda")
    for s in states:
        out += s[5:9].view(torch.int32)
    states[0][5:9].view(torch.int32).copy_(out)
    gd, gn, gi = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
    roo.RaycastStateToImages(gd, gn, gi, states[0])
    assert T.nan_equal(gd.MemcpyToHost(), want[0]), T.mismatch_report(gd.MemcpyToHost(), want[0])
    assert T.nan_equal(gn.MemcpyToHost(), want[1]), T.mismatch_report(gn.MemcpyToHost(), want[1])
    assert T.nan_equal(gi.MemcpyToHost(), want[2]), T.mismatch_report(gi.MemcpyToHost(), want[2])


# ---------------------------------------------------------------------------------
# projective point-plane ICP
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h,holes", [(640, 480, True), (160, 120, False), (80, 60, True), (20, 15, True), (96, 72, True),
                                       (16, 16, False), (48, 3, True)])
def test_gpu_icp_point_plane_vs_oracle(roo, w, h, holes):
    """kfx_icp_point_plane against the oracle: the summed system, every per-block system left in the
    workspace and the debug image are bit-identical (block tree in the reference's order, fixed final order)."""
    import test_tracking_cpu as TT
    from kangaroo_amd import tracking
    K, Pl, Pr, Nr, _, _ = TT.icp_inputs("room", w, h, holes)
    T_lp = tracking.se3_exp([0.004, -0.003, 0.002, 0.003, -0.002, 0.001])
    KT = (tracking.k_matrix(K) @ T_lp[:3]).astype(np.float32)
    T_pl = tracking.se3_inv(T_lp)[:3].astype(np.float32)
    odbg = oracle.Image(w, h, channels=4)
    want, blocks = oracle.icp_point_plane(Pl, Pr, Nr, KT, T_pl, 0.1, odbg, want_blocks=True)
    gPl, gPr, gNr = T.upload_image(roo, Pl.data), T.upload_image(roo, Pr.data), T.upload_image(roo, Nr.data)
    ws = roo.Image(116 * max(len(blocks), 1), 1, "u8")
    dbg = roo.Image(w, h, "f32x4")
    got = roo.PoseRefinementProjectiveIcpPointPlane(gPl, gPr, gNr, KT, T_pl, 0.1, ws, dbg)
    assert got.obs == int(want["obs"]) and (w < 64 or got.obs > 0)
    assert got.JTy.tobytes() == want["JTy"].tobytes() and got.raw.tobytes() == want["JTJ"].tobytes()
    assert got.sqErr.tobytes() == want["sqErr"].tobytes()
    assert T.nan_equal(dbg.MemcpyToHost(), odbg.data)
    left = np.frombuffer(ws.MemcpyToHost().tobytes()[:116 * len(blocks)], oracle.LSS_DTYPE)
    assert left[1:].tobytes() == blocks[1:].tobytes()       # slot 0 holds the final sum
    assert left[0].tobytes() == want.tobytes()
    # the debug image is optional
    again = roo.PoseRefinementProjectiveIcpPointPlane(gPl, gPr, gNr, KT, T_pl, 0.1, ws, None)
    assert again.raw.tobytes() == got.raw.tobytes() and again.obs == got.obs


def test_gpu_icp_argument_errors_and_tracking_loop(roo):
    import oracle_ops
    import test_tracking_cpu as TT
    from kangaroo_amd import tracking
    from kangaroo_amd._lib import KfxError
    w, h = 160, 120
    K = scenes.intrinsics(w, h)
    T_wp, T_wl = scenes.orbit_pose(0, 60), scenes.orbit_pose(1, 60)
    _, ray_v, ray_n, Ks = TT.pyramid_maps("room", w, h, T_wp, K)
    _, kin_v, _, _ = TT.pyramid_maps("room", w, h, T_wl, K)
    up = lambda imgs: [T.upload_image(roo, im.data) for im in imgs]
    g_kin, g_rv, g_rn = up(kin_v), up(ray_v), up(ray_n)
    small = roo.Image(115, 1, "u8")
    with pytest.raises(KfxError):
        roo.PoseRefinementProjectiveIcpPointPlane(g_kin[3], g_rv[3], g_rn[3], np.eye(4)[:3], np.eye(4)[:3], 0.1, small)
    with pytest.raises(KfxError):   # dPr smaller than dPl
        roo.PoseRefinementProjectiveIcpPointPlane(g_kin[0], g_rv[1], g_rn[0], np.eye(4)[:3], np.eye(4)[:3], 0.1, roo.Image(1 << 16, 1, "u8"))
    ws = roo.Image(w * 232, h, "u8")          # main.cpp:111: w * sizeof(LeastSquaresSystem<float,12>) x h
    dbg = roo.Image(w, h, "f32x4")
    got = tracking.refine_pose(roo, g_kin, g_rv, g_rn, Ks, ws, dbg)
    want = tracking.refine_pose(oracle_ops, kin_v, ray_v, ray_n, Ks, None)
    assert np.array_equal(got[0], want[0]) and got[1] == want[1] and got[2] == want[2]
    truth = tracking.se3_inv(np.vstack([T_wl, [0, 0, 0, 1]])) @ np.vstack([T_wp, [0, 0, 0, 1]])
    assert np.linalg.norm((tracking.se3_inv(truth) @ got[0])[:3, 3]) < 0.2 * np.linalg.norm(truth[:3, 3])


def test_gpu_tracking_pipeline_follows_the_orbit(roo):
    """End-to-end tracked KinectFusion on the GPU (no pose given after frame 0) at the application's image size."""
    from kangaroo_amd.pipeline import TrackingPipeline
    N, w, h, frames = 128, 640, 480, 8
    bmin, bmax, near, far = scenes.SCENES["room"]
    pipe = TrackingPipeline(roo, (N, N, N), bmin, bmax, w, h, near=near, far=far)
    worst = 0.0
    for i in range(frames):
        T_true = scenes.orbit_pose(i, 30)
        pipe.raw.MemcpyFromHost(scenes.render_depth("room", w, h, T_true, pipe.K))
        T_est = pipe.step(T_wl_init=T_true if i == 0 else None)
        assert pipe.tracking_good and np.isfinite(pipe.rmse)
        worst = max(worst, float(np.linalg.norm(T_est[:3, 3] - T_true[:3, 3])))
    drift_if_static = float(np.linalg.norm(scenes.orbit_pose(frames - 1, 30)[:3, 3] - scenes.orbit_pose(0, 30)[:3, 3]))
    assert worst < 0.2 * drift_if_static, (worst, drift_if_static)


@pytest.mark.parametrize("device_icp", [False, True])
def test_gpu_tracking_pipeline_follows_the_orbit_on_noisy_depth(roo, device_icp):
    """Round-5 verdict, item 4b: the tracked loop on SURVEY 8(d)'s noisy input -- 2 mm of Gaussian depth noise per pixel, a fresh draw
    per frame (seed 1234 + frame) -- over a whole orbit at the application's image size: no frame lost, no reset, and the worst position
    error stays a small fraction of the distance the camera travels (the figures are written to gpurun_out/noise_tracking.json)."""
    import json
    import os
    from kangaroo_amd.pipeline import TrackingPipeline
    N, w, h, frames = 256, 640, 480, 30
    bmin, bmax, near, far = scenes.SCENES["room"]
    prev = roo.set_math_mode("fast")
    try:
        pipe = TrackingPipeline(roo, (N, N, N), bmin, bmax, w, h, near=near, far=far, device_icp=device_icp)
        worst, lost, rmses = 0.0, 0, []
        for i in range(frames):
            T_true = scenes.orbit_pose(i, 30)
            pipe.raw.MemcpyFromHost(scenes.render_depth("room", w, h, T_true, pipe.K, noise_sigma=0.002, seed=1234 + i))
            T_est = pipe.step(T_wl_init=T_true if i == 0 else None)
            lost += 0 if pipe.tracking_good else 1
            rmses.append(float(pipe.rmse))
            worst = max(worst, float(np.linalg.norm(T_est[:3, 3] - T_true[:3, 3])))
    finally:
        roo.set_math_mode(prev)
    travel = max(float(np.linalg.norm(scenes.orbit_pose(i, 30)[:3, 3] - scenes.orbit_pose(0, 30)[:3, 3])) for i in range(frames))
    rep = {"device_icp": device_icp, "volume": N, "image": [w, h], "frames": frames, "depth_noise_sigma_m": 0.002, "frames_lost": lost, "resets": pipe.resets,
           "worst_position_error_mm": 1e3 * worst, "largest_distance_from_start_mm": 1e3 * travel, "rmse_max": max(rmses[1:]), "rmse_last": rmses[-1]}
    print("noise_tracking", json.dumps(rep))
    try:
        d = os.path.join(T.ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "noise_tracking_%s.json" % ("device" if device_icp else "host")), "w") as fh:
            json.dump(rep, fh, indent=1)
    except OSError:
        pass
    assert lost == 0 and pipe.resets == 0 and all(np.isfinite(rmses)), rep
    assert worst < 0.1 * travel, rep     # (noise-free: 0.26 mm over the orbit at 512^3; the camera gets 50 mm away from its start)


def test_gpu_tracking_pipeline_next_frame_preamble_under_the_pose_wait(roo):
    """TrackingPipeline.step(..., next_image=...): the next frame's pre-amble enqueued between the device-resident refinement and
    the wait for its pose (kfx_icp_refine_then) into a second set of maps -- poses, rmse and the model are those of the loop that
    runs every pre-amble at the start of its own frame, bit for bit (a frame without depth in between: the recovery path too)."""
    from kangaroo_amd.pipeline import TrackingPipeline
    N, w, h, n = 96, 320, 240, 9
    bmin, bmax, near, far = scenes.SCENES["room"]
    K = scenes.intrinsics(w, h)
    imgs = []
    for i in range(n):
        d = scenes.render_depth("room", w, h, scenes.orbit_pose(i, 30), K)
        if i == 5:
            d = np.full_like(d, np.nan)   # tracking lost altogether: reset + re-fuse at the next frame
        imgs.append(T.upload_image(roo, d))
    res = []
    for prefetch in (False, True):
        pipe = TrackingPipeline(roo, (N, N, N), bmin, bmax, w, h, K=K, near=near, far=far, device_icp=True, track=False)
        out = []
        for i in range(n):
            T_est = pipe.step(scenes.orbit_pose(0, 30) if i == 0 else None, imgs[i], next_image=imgs[i + 1] if prefetch and i + 1 < n else None)
            out.append((np.asarray(T_est, np.float64).tobytes(), float(pipe.rmse) if np.isfinite(pipe.rmse) else None, bool(pipe.tracking_good)))
        res.append((out, pipe.vol.MemcpyToHost().tobytes(), pipe.resets))
    assert res[0][2] == res[1][2] == 1
    assert res[0][0] == res[1][0]
    assert res[0][1] == res[1][1]


@pytest.mark.parametrize("device_icp", [False, True])
def test_gpu_tracking_pipeline_recovers_after_a_frame_without_depth(roo, device_icp):
    """The application's recovery path (main.cpp:223-242) on the GPU, host solve loop and device-resident ICP loop: one all-NaN
    depth frame mid-orbit -> no correspondence, rmse NaN, nothing fused, pose kept; next frame -> T_wl = identity, SdfReset(NaN),
    the frame fused, tracking goes on in the new world frame (= that frame's camera)."""
    from kangaroo_amd.pipeline import TrackingPipeline
    N, w, h, drop = 128, 320, 240, 4
    bmin, bmax, near, far = scenes.SCENES["room"]
    pipe = TrackingPipeline(roo, (N, N, N), bmin, bmax, w, h, near=near, far=far, device_icp=device_icp)
    anchor = None
    for i in range(10):
        T_true = scenes.orbit_pose(i, 30)
        depth = scenes.render_depth("room", w, h, T_true, pipe.K)
        if i == drop:
            depth = np.full_like(depth, np.nan)
        pipe.raw.MemcpyFromHost(depth)
        before = pipe.T_wl.copy()
        T_est = pipe.step(T_wl_init=T_true if i == 0 else None)
        if i == drop:
            assert not pipe.tracking_good and not np.isfinite(pipe.rmse) and np.array_equal(T_est, before) and pipe.resets == 0
            continue
        if i == drop + 1:
            assert pipe.resets == 1
            anchor = np.vstack([T_true, [0, 0, 0, 1]])
        assert pipe.tracking_good and np.isfinite(pipe.rmse), i
        T_abs = T_est if anchor is None else anchor @ T_est
        assert np.linalg.norm(T_abs[:3, 3] - T_true[:3, 3]) < 5e-3, (i, T_abs[:3, 3], T_true[:3, 3])
    assert pipe.resets == 1
    v = pipe.vol.tensor()[..., 0]
    assert bool(v.isnan().any()) and int(v.isfinite().sum()) > 0.2 * N ** 3


# ---------------------------------------------------------------------------------
# colour fusion / colour raycast
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("dims,w,h,cw,ch,full", [((64, 64, 64), 160, 120, 160, 120, False), ((40, 24, 19), 80, 60, 96, 72, False),
                                                  ((21, 18, 9), 80, 60, 96, 72, True), ((128, 128, 128), 640, 480, 640, 480, False)])
def test_gpu_colour_fusion_and_raycast_vs_oracle(roo, dims, w, h, cw, ch, full):
    """kfx_sdf_fuse_color / kfx_raycast_sdf_color / kfx_color_reset against the oracle: SDF cells, colour
    cells and the three raycast images bit-identical (extents of the reference's 16x16 launch with its z loop)."""
    import test_color_cpu as TC
    ovol, ocvol, K, Kimg, tr, near, far, inputs = TC.color_setup(0, w, h, cw, ch, dims=dims)
    bmin, bmax = scenes.SCENES["room"][0], scenes.SCENES["room"][1]
    vol = roo.BoundedVolume(dims[0], dims[1], dims[2], bmin, bmax)
    cvol = roo.BoundedVolume(dims[0], dims[1], dims[2], bmin, bmax, kind="c32")
    roo.SdfReset(vol, float("nan"))
    roo.ColorReset(cvol)
    # Fill(0.5) covers the pitch padding of the span too
    span = (dims[2] - 1) * cvol.img_pitch + (dims[1] - 1) * cvol.pitch + dims[0] * 4
    raw = cvol.storage[:span].view(dtype=__import__("torch").float32).cpu().numpy()
    assert (raw == 0.5).all()
    for fr in inputs:
        n = oracle.sdf_fuse_color(ovol, ocvol, fr["f"], fr["nrm"], fr["T_cw"], K, fr["rgb"], fr["T_iw"], Kimg, tr, scenes.MAX_W,
                                  scenes.MIN_COS_THETA, full_extent=full, nthreads=0)
        assert n > 0
        rgb = roo.Image(cw, ch, "u8x3")
        rgb.MemcpyFromHost(fr["rgb"].data)
        roo.SdfFuseColor(vol, cvol, T.upload_image(roo, fr["f"].data), T.upload_image(roo, fr["nrm"].data), fr["T_cw"], K, rgb,
                         fr["T_iw"], Kimg, tr, scenes.MAX_W, scenes.MIN_COS_THETA, full_extent=full)
    got_v, got_c = vol.MemcpyToHost(), cvol.MemcpyToHost()
    assert T.nan_equal(got_v, ovol.data), T.mismatch_report(got_v, ovol.data)
    assert T.nan_equal(got_c, ocvol.data), T.mismatch_report(got_c, ocvol.data)
    assert ((got_c != 0.5) & (got_c >= 0) & (got_c <= 1)).any()
    od, on, oi = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
    oracle.raycast_sdf_color(od, on, oi, ovol, ocvol, inputs[-1]["T_wc"], K, near, far, tr, True, nthreads=0)
    rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
    roo.RaycastSdfColor(rd, rn, ri, vol, cvol, inputs[-1]["T_wc"], K, near, far, tr, True)
    assert T.nan_equal(rd.MemcpyToHost(), od.data) and T.nan_equal(rn.MemcpyToHost(), on.data)
    assert T.nan_equal(ri.MemcpyToHost(), oi.data), T.mismatch_report(ri.MemcpyToHost(), oi.data)
    assert np.isfinite(od.data).mean() > 0.2


# ---------------------------------------------------------------------------------
# BASELINE configs C4 / C5 at full size on one GPU (size-independent properties)
# ---------------------------------------------------------------------------------
def test_gpu_c4_1024_cubed_in_eight_slabs(roo):
    """Config C4 (1024^3 over 8 GPUs) with the eight ranks emulated on one MI355X: each 'rank' owns a separately
    allocated slab (128 planes + 2 ghost planes per side) and integrates it through kfx_sdf_fuse_slab; the planes
    must be bit-identical to the monolithic 8 GiB volume, and the exact slab march (state handed from slab to
    slab, merged as SlabPipeline.raycast_exact does) must reproduce the monolithic RaycastSdf bit for bit."""
    import torch
    from kangaroo_amd.pipeline import slab_range
    N, w, h, world, G, scene = 1024, 640, 480, 8, 2, "room"
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    full = roo.BoundedVolume(N, N, N, bmin, bmax)
    roo.SdfReset(full, float("nan"))
    spans = [slab_range(N, r, world) for r in range(world)]
    stored = [(max(z0 - G, 0), min(z1 + G, N)) for z0, z1 in spans]
    f32 = np.float32
    size_z = f32(bmax[2]) - f32(bmin[2])
    slabs = []
    for s0, s1 in stored:  # the bbox SlabPipeline._alloc_volume gives a rank's volume
        lo = (bmin[0], bmin[1], float(f32(bmin[2]) + size_z * f32(s0) / f32(N - 1)))
        hi = (bmax[0], bmax[1], float(f32(bmin[2]) + size_z * f32(s1 - 1) / f32(N - 1)))
        v = roo.BoundedVolume(N, N, s1 - s0, lo, hi)
        roo.SdfReset(v, float("nan"))
        slabs.append(v)
    f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
    for i in range(2):
        T_wc = scenes.orbit_pose(i, 30)
        graw = T.upload_image(roo, scenes.render_depth(scene, w, h, T_wc, K))
        roo.BilateralFilter(f, graw, **scenes.BILATERAL)
        roo.DepthToVbo(vbo, f, K)
        roo.NormalsFromVbo(nrm, vbo)
        T_cw = scenes.se3_inverse(T_wc)
        roo.SdfFuse(full, f, nrm, T_cw, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
        for v, (s0, s1) in zip(slabs, stored):
            roo.SdfFuse(v, f, nrm, T_cw, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, full_extent=True, slab=(N, s0, bmin[2], bmax[2]))
    ft = full.tensor().view(torch.int32)
    for v, (s0, s1) in zip(slabs, stored):
        assert torch.equal(v.tensor().view(torch.int32), ft[s0:s1]), (s0, s1)
    assert int((~torch.isnan(full.tensor()[..., 0])).sum()) > 0.3 * N ** 3
    # exact march over the eight slabs
    rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
    roo.RaycastSdf(rd, rn, ri, full, T_wc, K, near, far, tr, True)
    want = (rd.MemcpyToHost(), rn.MemcpyToHost(), ri.MemcpyToHost())
    assert np.isfinite(want[0]).mean() > 0.5
    states = [torch.empty((9, h, w), dtype=torch.float32, device="cuda") for _ in range(world)]
    rounds = 0
    while True:
        for r in range(world):
            roo.RaycastSdfSlab(states[r], rounds == 0, slabs[r], (N, stored[r][0], float(bmin[2]), float(bmax[2])), spans[r][0], spans[r][1],
                               w, h, T_wc, K, near, far, tr, True)
        rounds += 1
        march = [s[0:5].view(torch.int32) for s in states]
        total = torch.zeros_like(march[0])
        for m in march:
            total += torch.where((m[4] != 0).unsqueeze(0), m, torch.zeros_like(m))
        for m in march:
            m.copy_(torch.where((total[4] != 0).unsqueeze(0), total, m))
        if not bool(((states[0][3] == 0) | (states[0][3] == 3)).any()):
            break
        assert rounds <= world + 3
    out = torch.zeros((4, h, w), dtype=torch.int32, device="cuda")
    for s in states:
        out += s[5:9].view(torch.int32)
    states[0][5:9].view(torch.int32).copy_(out)
    gd, gn, gi = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
    roo.RaycastStateToImages(gd, gn, gi, states[0])
    assert T.nan_equal(gd.MemcpyToHost(), want[0]) and T.nan_equal(gn.MemcpyToHost(), want[1]) and T.nan_equal(gi.MemcpyToHost(), want[2])
    del full, slabs, states, ft
    torch.cuda.empty_cache()


def test_gpu_c5_2048_cubed_half_volume_raycast(roo):
    """Config C5: a 2048^3 fp16 TSDF (32 GiB, one MI355X holds it) ray-cast against the analytic sphere it encodes
    (SdfSphere, the reference's own synthetic volume): depth within a voxel + half precision of the exact
    ray-sphere intersection, camera-frame normals pointing back along the sphere's radius."""
    import torch
    N, w, h = 2048, 640, 480
    K = scenes.intrinsics(w, h)
    vol = roo.BoundedVolume(N, N, N, (-1, -1, -1), (1, 1, 1), kind="f16")
    assert vol.img_pitch * N == 32 * 2 ** 30
    R = 0.9
    roo.SdfSphere(vol, (0.0, 0.0, 0.0), R)
    T_wc = np.array([[1, 0, 0, 0.05], [0, 1, 0, -0.02], [0, 0, 1, -2.6]], np.float32)
    tr = float(2.0 * np.linalg.norm(vol.VoxelSizeUnits()))
    rd, rn, ri = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)
    roo.RaycastSdf(rd, rn, ri, vol, T_wc, K, 0.1, 10.0, tr, True)
    d, n = rd.MemcpyToHost(), rn.MemcpyToHost()
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    ray = np.stack([(u - K[2]) / K[0], (v - K[3]) / K[1], np.ones_like(u)], -1)
    c = T_wc[:, 3].astype(np.float64)
    b = (ray * c).sum(-1)
    a = (ray * ray).sum(-1)
    disc = b * b - a * (c @ c - R * R)
    lam = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / a, np.nan)
    inner = disc > a * (R * 0.05) ** 2 * 4            # stay off the silhouette, where the march grazes the surface
    hit = np.isfinite(d)
    assert (hit[inner]).all() and hit.sum() > 0.3 * w * h
    voxel = 2.0 / (N - 1)
    err = np.abs(d[inner] - lam[inner])
    assert err.max() < 2 * voxel + 2e-3, err.max()      # half cells: 11-bit significand on |sdf| <~ 1
    pos = c + ray * d[..., None]
    radial = pos / np.linalg.norm(pos, axis=-1, keepdims=True)   # world = camera axes (identity rotation)
    cosang = (n[..., :3] * radial).sum(-1)
    assert (cosang[inner] > 0.97).mean() > 0.99
    del vol
    torch.cuda.empty_cache()
    # the same 32 GiB cell array under SdfFuse: one frame of the room integrated into 2048^3 half cells (8.6 G voxels,
    # offsets beyond 2^32 in every index expression), then ray-cast back: the model reproduces the measured depth
    bmin, bmax, near, far = scenes.SCENES["room"]
    vol = roo.BoundedVolume(N, N, N, bmin, bmax, kind="f16")
    roo.SdfReset(vol, float("nan"))
    tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    graw = T.upload_image(roo, scenes.render_depth("room", w, h, None, K))
    f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
    roo.BilateralFilter(f, graw, **scenes.BILATERAL)
    roo.DepthToVbo(vbo, f, K)
    roo.NormalsFromVbo(nrm, vbo)
    Tid = scenes.identity_pose()
    prev = roo.set_math_mode("fast")
    try:
        roo.SdfFuse(vol, f, nrm, Tid, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
    finally:
        roo.set_math_mode(prev)
    zw = int((3.8 - bmin[2]) / (bmax[2] - bmin[2]) * (N - 1))     # slices around the back wall (z = 3.8 m): 29 GiB into the array
    last = vol.tensor()[zw - 6: zw + 2]
    assert bool((~torch.isnan(last[..., 0].float())).any())
    assert bool(torch.isnan(vol.tensor()[N - 4:][..., 0].float()).all())      # behind the wall: never observed
    roo.RaycastSdf(rd, rn, ri, vol, Tid, K, near, far, tr, True)
    d2, filt = rd.MemcpyToHost(), f.MemcpyToHost()
    err = np.abs(d2 - filt)
    ok = np.isfinite(err)
    assert ok.sum() > 0.5 * w * h
    assert np.median(err[ok]) < 2.0 * (bmax[0] - bmin[0]) / (N - 1) + 1e-3   # a voxel or two + half precision of the cells
    del vol, last
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------
# mesh extraction
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["sphere", "fused_room_colour", "ragged_unobserved"])
def test_gpu_marching_cubes_vs_oracle(roo, tmp_path, case):
    """kfx_mc_count / kfx_mc_emit against the oracle's host marching cubes with the same case tables: vertices,
    normals and colours bit-identical and in the reference's emission order (x outer, y, z inner)."""
    import test_color_cpu as TC
    import test_mesh_cpu as TM
    from kangaroo_amd import mesh
    ntri, mask, tri = TM.tables()
    ocvol = None
    if case == "sphere":
        ovol = TM.sphere_volume(48, 0.7)
    elif case == "fused_room_colour":
        ovol, ocvol, K, Kimg, tr, near, far, inputs = TC.color_setup(0, 160, 120, 160, 120, dims=(64, 64, 64))
        for fr in inputs:
            oracle.sdf_fuse_color(ovol, ocvol, fr["f"], fr["nrm"], fr["T_cw"], K, fr["rgb"], fr["T_iw"], Kimg, tr, 1000.0, 0.1, nthreads=0)
    else:
        ovol = oracle.Volume(21, 13, 34, (-1, -0.5, -1), (1, 0.7, 1.5), pitch_bytes=21 * 8 + 40)
        oracle.sdf_sphere(ovol, (0.1, 0.0, 0.2), 0.45)
        ovol.data[:, :, :, 0][(np.add.outer(np.add.outer(np.arange(34), np.arange(13)), np.arange(21)) % 7) == 0] = np.nan
    want_v, want_n, want_c = oracle.marching_cubes(ovol, ocvol, ntri, mask, tri)
    assert len(want_v) > 300
    vol = roo.BoundedVolume(ovol.w, ovol.h, ovol.d, ovol.boxmin, ovol.boxmax, pitch=ovol.pitch if case == "ragged_unobserved" else None)
    vol.MemcpyFromHost(ovol.data)
    cvol = None
    if ocvol is not None:
        cvol = roo.BoundedVolume(ocvol.w, ocvol.h, ocvol.d, ocvol.boxmin, ocvol.boxmax, kind="c32")
        cvol.MemcpyFromHost(ocvol.data)
    v, n, c = mesh.ExtractMesh(vol, cvol)
    assert v.shape == want_v.shape
    assert T.nan_equal(v.cpu().numpy(), want_v) and T.nan_equal(n.cpu().numpy(), want_n)
    if ocvol is not None:
        assert T.nan_equal(c.cpu().numpy(), want_c) and (want_c[:, :3] != 0.5).any()
    else:
        assert c is None
    nt = mesh.SaveMesh(str(tmp_path / "m"), vol, cvol)
    raw = open(str(tmp_path / "m.ply"), "rb").read()
    assert nt == len(want_v) // 3 and raw.startswith(b"ply\nformat binary_little_endian 1.0") and b"element face %d" % nt in raw[:600]


# ---------------------------------------------------------------------------------
# the rest of cu_raycast.h / cu_sdffusion.h: analytic renderers and SdfDistance
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h", [(160, 120), (67, 45)])
def test_gpu_analytic_renderers_and_sdf_distance(roo, w, h):
    K = scenes.intrinsics(w, h)
    T_wc = scenes.orbit_pose(2, 8)
    # box
    od, gd = oracle.Image(w, h), roo.Image(w, h)
    oracle.raycast_box(od, T_wc, K, (-0.5, -0.4, 2.0), (0.6, 0.5, 3.0))
    roo.RaycastBox(gd, T_wc, K, (-0.5, -0.4, 2.0), (0.6, 0.5, 3.0))
    assert T.nan_equal(gd.MemcpyToHost(), od.data) and np.isfinite(od.data).any() and np.isnan(od.data).any()
    # sphere then plane composited into the same depth / shade images (nearer hit wins, NaN = empty)
    oi, gi = oracle.Image(w, h), roo.Image(w, h)
    od.data[...] = np.nan
    gd.MemcpyFromHost(od.data)
    for c, r in (((0.1, 0.0, 3.0), 0.5), ((-0.4, 0.2, 2.5), 0.3)):
        oracle.raycast_sphere(od, oi, T_wc, K, c, r)
        roo.RaycastSphere(gd, gi, T_wc, K, c, r)
    oracle.raycast_plane(od, oi, T_wc, K, (0.0, 0.0, -1.0 / 3.8))
    roo.RaycastPlane(gd, gi, T_wc, K, (0.0, 0.0, -1.0 / 3.8))
    assert T.nan_equal(gd.MemcpyToHost(), od.data) and T.nan_equal(gi.MemcpyToHost(), oi.data)
    assert np.isfinite(od.data).mean() > 0.9
    # depth only (img = None) for the sphere
    od2, gd2 = oracle.Image(w, h), roo.Image(w, h)
    od2.data[...] = np.nan
    gd2.MemcpyFromHost(od2.data)
    oracle.raycast_sphere(od2, None, T_wc, K, (0.1, 0.0, 3.0), 0.5)
    roo.RaycastSphere(gd2, None, T_wc, K, (0.1, 0.0, 3.0), 0.5)
    assert T.nan_equal(gd2.MemcpyToHost(), od2.data)
    # SdfDistance of the rendered depth against a fused volume
    ovol = T.make_volume(48, "room")
    Kf, tr, fr = T.fuse_frames_oracle(ovol, "room", w, h, 2)
    vol = T.upload_volume(roo, ovol)
    odist, gdist = oracle.Image(w, h), roo.Image(w, h)
    oracle.sdf_distance(odist, od, ovol, T_wc, K)
    roo.SdfDistance(gdist, gd, vol, T_wc, K, tr)
    assert T.nan_equal(gdist.MemcpyToHost(), odist.data) and np.isfinite(odist.data).any()


def test_gpu_depth_tools(roo):
    """Disp2Depth, FilterBadKinectData (float / unsigned short) and ColourVbo against the oracle, bit / byte exact."""
    from test_oracle_cpu import depth_tool_inputs
    rng = np.random.default_rng(9)
    w, h = 83, 41
    disp = rng.uniform(0.0, 64.0, (h, w)).astype(np.float32)
    disp[::7, ::3] = 0.0
    od, gd = oracle.Image(w, h), roo.Image(w, h)
    di = oracle.Image(w, h)
    di.data[...] = disp
    oracle.disp2depth(di, od, 570.0, 0.075, 1.0)
    roo.Disp2Depth(T.upload_image(roo, disp), gd, 570.0, 0.075, 1.0)
    assert T.nan_equal(gd.MemcpyToHost(), od.data) and np.isnan(od.data).any() and np.isfinite(od.data).any()
    mm = rng.integers(0, 5000, (h, w)).astype(np.uint16)
    for arr in (mm, mm.astype(np.float32)):
        oi = oracle.Image(w, h, arr.dtype)
        oi.data[...] = arr
        oo, go = oracle.Image(w, h), roo.Image(w, h)
        oracle.filter_bad_kinect(oo, oi)
        roo.FilterBadKinectData(go, T.upload_image(roo, arr))
        assert T.nan_equal(go.MemcpyToHost(), oo.data) and np.isnan(oo.data).any()
    vbo, rgb, KT = depth_tool_inputs()
    want = oracle.Image(vbo.w, vbo.h, np.uint8, 4)
    oracle.colour_vbo(want, vbo, rgb, KT)
    grgb = roo.Image(rgb.w, rgb.h, "u8x3")
    grgb.MemcpyFromHost(rgb.data)
    gid = roo.Image(vbo.w, vbo.h, "u8x4")
    roo.ColourVbo(gid, T.upload_image(roo, vbo.data), grgb, KT)
    assert np.array_equal(gid.MemcpyToHost(), want.data)


@pytest.mark.parametrize("guide_kind,size", [("f32", 2), ("u8", 3)])
def test_gpu_joint_bilateral_filter(roo, guide_kind, size):
    """BilateralFilter(dOut, dIn, dImg, gs, gr, gc, size): hardware exp, so the same 2e-6 relative bar as the plain filter."""
    rng = np.random.default_rng(11)
    w, h = 97, 53
    depth = (2.0 + 0.5 * rng.random((h, w))).astype(np.float32)
    guide = rng.integers(0, 256, (h, w)).astype(np.uint8) if guide_kind == "u8" else rng.random((h, w)).astype(np.float32)
    oi, og, oo = oracle.Image(w, h), oracle.Image(w, h, guide.dtype), oracle.Image(w, h)
    oi.data[...] = depth
    og.data[...] = guide
    gc = 20.0 if guide_kind == "u8" else 0.1
    oracle.bilateral_guided(oo, oi, og, 1.5, 0.1, gc, size)
    go = roo.Image(w, h)
    roo.BilateralFilterGuided(go, T.upload_image(roo, depth), T.upload_image(roo, guide), 1.5, 0.1, gc, size)
    got = go.MemcpyToHost()
    assert np.allclose(got, oo.data, rtol=BILATERAL_RTOL, atol=0) and np.isfinite(got).all()


@pytest.mark.parametrize("kind,minval,size", [("f32", 0.2, 3), ("f32", None, 2), ("u16", 200, 3), ("u8", None, 1)])
def test_gpu_bilateral_fast_mode(roo, kind, minval, size):
    """BilateralFilter under KFX_MATH_FAST (reciprocal-multiply exponent argument, unrolled taps): same NaN pattern
    as the exact kernel and values within 1e-5 relative of the oracle (the weights are hardware exp either way)."""
    rng = np.random.default_rng(21)
    w, h = 161, 77
    if kind == "f32":
        img = scenes.render_depth("room", w, h, None, scenes.intrinsics(w, h)).astype(np.float32)
        img += rng.normal(0, 0.01, img.shape).astype(np.float32)
        img[rng.random(img.shape) < 0.05] = np.nan
        img[::9, ::4] = 0.1
    elif kind == "u16":
        img = rng.integers(0, 4000, (h, w)).astype(np.uint16)
    else:
        img = rng.integers(0, 256, (h, w)).astype(np.uint8)
    oi, oo = oracle.Image(w, h, img.dtype), oracle.Image(w, h)
    oi.data[...] = img
    gr = 0.1 if kind == "f32" else 40.0
    oracle.bilateral(oo, oi, 1.5, gr, size, minval)
    go = roo.Image(w, h)
    prev = roo.set_math_mode("fast")
    try:
        roo.BilateralFilter(go, T.upload_image(roo, img), 1.5, gr, size, minval)
    finally:
        roo.set_math_mode(prev)
    got = go.MemcpyToHost()
    assert np.array_equal(np.isnan(got), np.isnan(oo.data))
    ok = ~np.isnan(got)
    assert np.allclose(got[ok], oo.data[ok], rtol=1e-5, atol=0)


@pytest.mark.parametrize("w,h", [(160, 120), (67, 45), (1, 1), (64, 1)])
def test_gpu_fused_vbo_normals_equals_the_two_operators(roo, w, h):
    rng = np.random.default_rng(5)
    K = scenes.intrinsics(max(w, 8), max(h, 8))
    depth = (2.0 + rng.random((h, w))).astype(np.float32)
    depth[rng.random((h, w)) < 0.1] = np.nan
    gd = T.upload_image(roo, depth)
    v1, n1, v2, n2 = roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
    roo.DepthToVbo(v1, gd, K)
    roo.NormalsFromVbo(n1, v1)
    roo.DepthToVboNormals(v2, n2, gd, K)
    assert T.nan_equal(v2.MemcpyToHost(), v1.MemcpyToHost()) and T.nan_equal(n2.MemcpyToHost(), n1.MemcpyToHost())


def test_gpu_device_resident_icp_loop(roo):
    """kfx_icp_refine (solves on the device, one synchronisation) against the host loop on the same maps: the same pose to
    float64 round-off of sin / cos (the float32 transforms handed to the kernel may differ in a last bit, so the bar is
    1e-6 on the transform), and the tracked pipeline follows the orbit equally well."""
    import test_tracking_cpu as TT
    from kangaroo_amd import tracking
    from kangaroo_amd.pipeline import TrackingPipeline
    w, h = 160, 120
    K = scenes.intrinsics(w, h)
    T_wp, T_wl = scenes.orbit_pose(0, 60), scenes.orbit_pose(1, 60)
    _, ray_v, ray_n, Ks = TT.pyramid_maps("room", w, h, T_wp, K)
    _, kin_v, _, _ = TT.pyramid_maps("room", w, h, T_wl, K)
    up = lambda imgs: [T.upload_image(roo, im.data) for im in imgs]
    g_kin, g_rv, g_rn = up(kin_v), up(ray_v), up(ray_n)
    ws, dbg = roo.Image(w * 232, h, "u8"), roo.Image(w, h, "f32x4")
    for its in (tracking.DEFAULT_ITS, (4, 3, 3, 3), (2, 0, 0, 0)):
        want_T, want_rmse, want_good = tracking.refine_pose(roo, g_kin, g_rv, g_rn, Ks, ws, dbg, its=its)
        got_T, got_rmse, got_obs, got_good = roo.IcpRefine(g_kin, g_rv, g_rn, Ks, its, 0.1, 0.10, ws, dbg)
        assert np.abs(got_T - want_T).max() < 1e-6, (its, np.abs(got_T - want_T).max())
        assert abs(got_rmse - want_rmse) < 1e-6 and got_good == want_good and got_obs > 0
    N, frames = 64, 6
    bmin, bmax, near, far = scenes.SCENES["room"]
    pipe = TrackingPipeline(roo, (N, N, N), bmin, bmax, 320, 240, near=near, far=far, device_icp=True)
    assert pipe.device_icp
    worst = 0.0
    for i in range(frames):
        T_true = scenes.orbit_pose(i, 30)
        pipe.raw.MemcpyFromHost(scenes.render_depth("room", 320, 240, T_true, pipe.K))
        T_est = pipe.step(T_wl_init=T_true if i == 0 else None)
        assert pipe.tracking_good
        worst = max(worst, float(np.linalg.norm(T_est[:3, 3] - T_true[:3, 3])))
    assert worst < 0.2 * float(np.linalg.norm(scenes.orbit_pose(frames - 1, 30)[:3, 3] - scenes.orbit_pose(0, 30)[:3, 3]))


def test_gpu_persistent_icp_kernel_equals_the_chain_of_launches():
    """kfx_icp_refine as ONE persistent launch (k_icp_refine_persistent: all levels and iterations, a grid-wide barrier between
    the block systems and the step, every workgroup solving for itself) against the chain of twelve launches
    (KFX_ICP_PERSISTENT=0): the same per-pixel function, block tree, block order and float64 step, so the same pose, rmse and
    observation count BIT FOR BIT -- at 640x480 (1200 blocks over the resident grid: several per workgroup) and 160x120, for the
    application's schedule and two others, repeated calls (the barrier words are re-armed per call)."""
    import json
    import sys
    code = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from kangaroo_amd import roo, scenes
import kfx_testlib as T
import test_tracking_cpu as TT
out = []
for w, h in ((640, 480), (160, 120)):
    K = scenes.intrinsics(w, h)
    T_wp, T_wl = scenes.orbit_pose(0, 60), scenes.orbit_pose(1, 60)
    _, ray_v, ray_n, Ks = TT.pyramid_maps("room", w, h, T_wp, K)
    _, kin_v, _, _ = TT.pyramid_maps("room", w, h, T_wl, K)
    up = lambda imgs: [T.upload_image(roo, im.data) for im in imgs]
    g_kin, g_rv, g_rn = up(kin_v), up(ray_v), up(ray_n)
    ws, dbg = roo.Image(w * 232, h, "u8"), roo.Image(w, h, "f32x4")
    for its in ((1, 0, 2, 3), (4, 3, 3, 3), (2, 0, 0, 0)):
        for rep in range(3):
            Tm, rmse, obs, good = roo.IcpRefine(g_kin, g_rv, g_rn, Ks, its, 0.1, 0.10, ws, dbg)
            out.append([np.asarray(Tm, np.float64).tobytes().hex(), float(np.float32(rmse)).hex(), int(obs), bool(good), dbg.MemcpyToHost().tobytes().hex()[:4096]])
print("RESULT " + json.dumps(out))
""" % (T.ROOT, os.path.join(T.ROOT, "tests"))
    res = {}
    # "pair": k_icp_point_plane + k_lss_final_solve per iteration (the chain; its 6 x 6 solve runs on a whole wave);
    # "persistent": one launch for the whole loop (the solve on one lane, as written in round 4)
    for name, env in (("pair", dict(KFX_ICP_PERSISTENT="0")), ("persistent", dict(KFX_ICP_PERSISTENT="1"))):
        o = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900, cwd=T.ROOT)
        line = [l for l in o.stdout.splitlines() if l.startswith("RESULT ")]
        assert o.returncode == 0 and line, o.stdout[-2000:] + o.stderr[-3000:]
        res[name] = json.loads(line[0][7:])
    assert len(res["pair"]) == len(res["persistent"]) == 18
    for other in ("persistent",):
        for a, b in zip(res["pair"], res[other]):
            assert a == b, (other, a[:4], b[:4])
            assert a[2] > 0
        for k in range(0, 18, 3):   # repeated calls give the same answer (the ticket / barrier words are re-armed)
            assert res[other][k] == res[other][k + 1] == res[other][k + 2]


def test_gpu_texture_depth(roo):
    """TextureDepth, single-keyframe and blended forms, against the oracle (bit-identical)."""
    from test_oracle_cpu import texture_inputs
    K, T_wd, rd, rn, ri, kfs = texture_inputs()
    w, h = rd.w, rd.h
    gd, gn, gi = T.upload_image(roo, rd.data), T.upload_image(roo, rn.data), T.upload_image(roo, ri.data)
    gk = []
    for img, T_iw, Kk in kfs:
        g = roo.Image(img.w, img.h, "u8x3")
        g.MemcpyFromHost(img.data)
        gk.append((g, T_iw, Kk))
    for sel, phong in ((slice(0, 1), False), (slice(0, 3), True), (slice(1, 2), True)):
        want = oracle.Image(w, h, channels=4)
        oracle.texture_depth(want, kfs[sel], rd, rn, T_wd, K, ri if phong else None)
        out = roo.Image(w, h, "f32x4")
        roo.TextureDepth(out, gk[sel], gd, gn, T_wd, K, gi if phong else None)
        assert T.nan_equal(out.MemcpyToHost(), want.data), T.mismatch_report(out.MemcpyToHost(), want.data)


def test_gpu_colour_fusion_fast_mode(roo):
    """Colour fusion under KFX_MATH_FAST against the exact path: same classification up to a handful of voxels at
    predicate boundaries, TSDF within 1e-4 and grey levels within 2e-3 (half a grey step of 1/255)."""
    import torch
    import test_color_cpu as TC
    dims, w, h = (128, 128, 128), 320, 240
    ovol, ocvol, K, Kimg, tr, near, far, inputs = TC.color_setup(0, w, h, w, h, dims=dims)
    bmin, bmax = scenes.SCENES["room"][0], scenes.SCENES["room"][1]
    res = {}
    for mode in ("exact", "fast"):
        prev = roo.set_math_mode(mode)
        try:
            vol = roo.BoundedVolume(*dims, bmin, bmax)
            cvol = roo.BoundedVolume(*dims, bmin, bmax, kind="c32")
            roo.SdfReset(vol, float("nan"))
            roo.ColorReset(cvol)
            for fr in inputs:
                rgb = roo.Image(w, h, "u8x3")
                rgb.MemcpyFromHost(fr["rgb"].data)
                roo.SdfFuseColor(vol, cvol, T.upload_image(roo, fr["f"].data), T.upload_image(roo, fr["nrm"].data), fr["T_cw"], K, rgb,
                                 fr["T_iw"], Kimg, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
            res[mode] = (vol.tensor().clone(), cvol.tensor().clone())
        finally:
            roo.set_math_mode(prev)
    (a, ca), (b, cb) = res["exact"], res["fast"]
    na, nb = torch.isnan(a[..., 0]), torch.isnan(b[..., 0])
    assert int((na != nb).sum()) <= 20
    both = ~na & ~nb
    d = (a[..., 0][both] - b[..., 0][both]).abs()
    assert float((d > 1e-4).float().mean()) < 1e-4 and float(d.median()) < 1e-6
    dc = (ca[..., 0][both] - cb[..., 0][both]).abs()
    assert float((dc > 2e-3).float().mean()) < 1e-4, float(dc.max())
    assert float(dc.median()) < 1e-6


# ---------------------------------------------------------------------------------
# arithmetic shortcuts that must not change a bit
# ---------------------------------------------------------------------------------
def test_gpu_division_by_uniform_divisor_is_the_ieee_quotient(roo):
    """kfx_device.h div_uniform (q0 = a * (1/b); r = fma(-b, q0, a); q = fma(r, 1/b, q0)) against the hardware's IEEE
    division, exhaustively: all 2^32 numerator patterns (those inside the shortcut's operand range, ~31 % of them) for
    divisors with awkward significands (all ones, one ulp above a power of two), the benchmark's box sizes, negative
    and random values across the allowed exponent range."""
    import ctypes as C
    import torch
    from kangaroo_amd import _lib
    L = _lib.load()
    _lib.load_debug().kfx_debug_div_uniform_check.restype = C.c_int
    _lib.load_debug().kfx_debug_div_uniform_check.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(7)
    special = np.array([0x3fffffff, 0x3f800001, 0x3f7fffff, 0x40000000, 0x3faaaaab, 0x3f800000, 0x3fb504f3, 0x3fc00001], np.uint32).view(np.float32)
    divisors = [2.0, 1.8, 3.0, 0.1, 7.3, -2.5, 1e-9, 3e9, float(np.float32(2.0) / np.float32(511.0))] + [float(x) for x in special]
    mant = rng.integers(0, 1 << 23, 24, dtype=np.uint32)
    expo = rng.integers(127 - 39, 127 + 39, 24, dtype=np.uint32)
    divisors += [float(x) for x in ((expo << 23) | mant).view(np.float32)]
    divisors += [511.0, 1023.0, 255.0, 2047.0, 199.0]   # dims - 1: the tiled SdfFuse kernels' voxel positions (fuse.hip voxel_pos)
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    for b in divisors:
        out.zero_()
        assert _lib.load_debug().kfx_debug_div_uniform_check(C.c_float(b), C.c_void_p(out.data_ptr()), None) == 0
        bad, tested = (int(v) for v in out.cpu())
        assert tested > 1_300_000_000 and bad == 0, (b, bad, tested)


def test_gpu_shared_reciprocal_division_and_sqrt_are_the_ieee_results(roo):
    """The exact SdfFuse kernel divides three numerators by one Z through a shared Newton-refined reciprocal and takes
    its square root by the rsq iteration (kfx_device.h: rcp_nr / div_core / sqrt_core) instead of hipcc's scaled expansions.
    Inside the operand ranges the kernel establishes per brick these must be the hardware's IEEE results bit for bit:
    every divisor significand in ten binades x 24 numerators each (two seeds: 4e9 quotients), and every float of
    [2^-80, 2^80] for the square root."""
    import ctypes as C
    import torch
    from kangaroo_amd import _lib
    L = _lib.load()
    _lib.load_debug().kfx_debug_div_core_check.restype = C.c_int
    _lib.load_debug().kfx_debug_div_core_check.argtypes = [C.c_uint, C.c_int, C.c_void_p, C.c_void_p]
    _lib.load_debug().kfx_debug_sqrt_core_check.restype = C.c_int
    _lib.load_debug().kfx_debug_sqrt_core_check.argtypes = [C.c_void_p, C.c_void_p]
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    for seed in (1, 20261002):
        out.zero_()
        assert _lib.load_debug().kfx_debug_div_core_check(seed, 24, C.c_void_p(out.data_ptr()), None) == 0
        bad, tested = (int(v) for v in out.cpu())
        assert tested == (1 << 23) * 10 * 24 and bad == 0, (seed, bad, tested)
    out.zero_()
    assert _lib.load_debug().kfx_debug_sqrt_core_check(C.c_void_p(out.data_ptr()), None) == 0
    bad, tested = (int(v) for v in out.cpu())
    assert tested == 160 * (1 << 23) + 1 and bad == 0, (bad, tested)


def test_gpu_wave_reductions_without_the_lds_crossbar_equal_shuffles():
    """wave_xor_combine (kfx_device.h: DPP quad permutes, v_permlane16_swap, v_permlane32_swap) -- the lane reductions of the
    tracked SdfFuse kernels -- gives min / max(x[lane], x[lane ^ d]) for d = 1, 2, 16, 32 exactly as __shfl_xor does."""
    import ctypes as C
    import torch
    from kangaroo_amd import _lib
    L = _lib.load()
    _lib.load_debug().kfx_debug_wave_xor_check.restype = C.c_int
    _lib.load_debug().kfx_debug_wave_xor_check.argtypes = [C.c_uint, C.c_void_p, C.c_void_p]
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    for seed in (3, 77, 20261002):
        assert _lib.load_debug().kfx_debug_wave_xor_check(seed, C.c_void_p(out.data_ptr()), None) == 0
    bad, tested = (int(v) for v in out.cpu())
    assert tested == 3 * 64 * 256 * 64 * 8 and bad == 0, (bad, tested)


def test_gpu_raycast_levels_equal_per_level_calls(roo):
    """kfx_raycast_sdf_levels (all pyramid levels of the tracking loop in one launch) writes, per level, exactly what
    kfx_raycast_sdf writes: fp32 and fp16 cells, odd level sizes, an empty level list, and the tracked pipeline
    follows the same trajectory with either."""
    import torch
    from kangaroo_amd.pipeline import TrackingPipeline
    N, w, h, scene = 96, 200, 148, "room"
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    for kind in ("f32", "f16"):
        vol = roo.BoundedVolume(N, N, N, bmin, bmax, kind=kind)
        roo.SdfReset(vol, float("nan"))
        f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
        for i in range(2):
            T_wc = scenes.orbit_pose(i, 30)
            roo.BilateralFilter(f, T.upload_image(roo, scenes.render_depth(scene, w, h, T_wc, K)), **scenes.BILATERAL)
            roo.DepthToVbo(vbo, f, K)
            roo.NormalsFromVbo(nrm, vbo)
            roo.SdfFuse(vol, f, nrm, scenes.se3_inverse(T_wc), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
        levels = (0, 1, 3)
        sizes = {l: (max(w >> l, 1), max(h >> l, 1)) for l in levels}
        Ks = [scenes.intrinsics_level(K, l) for l in levels]
        mk = lambda: [(roo.Image(*sizes[l]), roo.Image(*sizes[l], "f32x4"), roo.Image(*sizes[l])) for l in levels]
        a, b = mk(), mk()
        T_wc = scenes.orbit_pose(1, 30)
        for (d, n, i), Kl in zip(a, Ks):
            roo.RaycastSdf(d, n, i, vol, T_wc, Kl, near, far, tr, True)
        roo.RaycastSdfLevels(b, vol, T_wc, Ks, near, far, tr, True)
        for x, y in zip(a, b):
            for p_, q_ in zip(x, y):
                assert T.nan_equal(p_.MemcpyToHost(), q_.MemcpyToHost())
        # optional fourth output: the vertex map DepthToVbo(depth, K) of each rendering, from the same launch (one level without)
        c = [t + ((roo.Image(*sizes[l], "f32x4"),) if l != 1 else (None,)) for t, l in zip(mk(), levels)]
        roo.RaycastSdfLevels(c, vol, T_wc, Ks, near, far, tr, True)
        for x, y, Kl in zip(a, c, Ks):
            for p_, q_ in zip(x, y[:3]):
                assert T.nan_equal(p_.MemcpyToHost(), q_.MemcpyToHost())
            if y[3] is not None:
                want = roo.Image(x[0].w, x[0].h, "f32x4")
                roo.DepthToVbo(want, x[0], Kl)
                assert T.nan_equal(want.MemcpyToHost(), y[3].MemcpyToHost())
        assert np.isfinite(a[0][0].MemcpyToHost()).mean() > 0.2
        roo.RaycastSdfLevels([], vol, T_wc, [], near, far, tr, True)   # nothing to do, no error
    poses = []
    for one in (False, True):
        pipe = TrackingPipeline(roo, (N, N, N), bmin, bmax, 320, 240, near=near, far=far, one_raycast=one)
        assert pipe.one_raycast == one
        tr_ = []
        for i in range(4):
            T_true = scenes.orbit_pose(i, 30)
            pipe.raw.MemcpyFromHost(scenes.render_depth(scene, 320, 240, T_true, pipe.K))
            tr_.append(pipe.step(T_wl_init=T_true if i == 0 else None).copy())
        poses.append(np.stack(tr_))
        torch.cuda.synchronize()
    assert np.array_equal(poses[0], poses[1])


def test_gpu_sampler_general_paths_vs_oracle():
    """The raycast / mesh samplers without their two shortcuts (KFX_SAMPLER_SHORTCUTS=0: hardware division instead of
    div_uniform, 64-bit addresses instead of base + 32-bit offset) -- the paths that volumes above 4 GiB take -- rerun the
    oracle-parity tests of raycast, colour raycast, slab march and marching cubes in a child process."""
    import subprocess
    import sys
    env = dict(os.environ, KFX_SAMPLER_SHORTCUTS="0")
    sel = "fuse_raycast_vs_oracle or colour_fusion_and_raycast or exact_slab_raycast or marching_cubes or half_cells_fuse_raycast"
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", sel],
                         env=env, capture_output=True, text=True, timeout=1200, cwd=T.ROOT)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.parametrize("knob", ["KFX_FUSE_XCD_SWIZZLE=0", "KFX_FUSE_XCD_SWIZZLE=3", "KFX_FUSE_R_SMALL=0", "KFX_FUSE_R_SMALL=9",
                                  "KFX_FUSE_BRICK=0", "KFX_FUSE_BRICK=1", "KFX_FUSE_ZU=4", "KFX_FUSE_ZU=1", "KFX_FUSE_CAP=256",
                                  "KFX_FUSE_EXACT_SHARED=0"])
def test_gpu_fuse_scheduling_knobs_do_not_change_a_bit(knob):
    """The A/B knobs of SdfFuse (workgroup order, tile capacity, brick shape, slices per iteration, division
    sequences) select launch geometry and instruction sequences, never values: the oracle-parity tests of SdfFuse -- exact
    numerics bit for bit, fast numerics within its tolerance, tracked kernels included -- rerun in a child process per knob."""
    import subprocess
    import sys
    k, v = knob.split("=")
    here = os.path.dirname(os.path.abspath(__file__))
    sel = "fuse_raycast_vs_oracle or fast_mode_within_tolerance or tracked_fuse_and_raycast"
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), os.path.join(here, "test_gpu_summary.py"), "-m", "gpu", "-x", "-q",
                          "-k", sel], env=dict(os.environ, **{k: v}), capture_output=True, text=True, timeout=1200, cwd=T.ROOT)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


_COLOUR_FULL_SIZE = r"""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from kangaroo_amd import roo, scenes
N, w, h = 512, 640, 480
bmin, bmax, near, far = scenes.SCENES["room"]
K = scenes.intrinsics(w, h)
tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
roo.set_math_mode(sys.argv[2])
vol, cvol = roo.BoundedVolume(N, N, N, bmin, bmax), roo.BoundedVolume(N, N, N, bmin, bmax, kind="c32")
roo.SdfReset(vol, float("nan")); roo.ColorReset(cvol)
f, vbo, nrm, rgb = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4"), roo.Image(w, h, "u8x3")
rgb.MemcpyFromHost(np.random.default_rng(5).integers(0, 256, (h, w, 3), dtype=np.uint8))
for i in range(3):
    T_wc = scenes.orbit_pose(3 * i, 30)
    raw = roo.Image(w, h).MemcpyFromHost(scenes.render_depth("room", w, h, T_wc, K))
    roo.BilateralFilter(f, raw, **scenes.BILATERAL); roo.DepthToVbo(vbo, f, K); roo.NormalsFromVbo(nrm, vbo)
    T_cw = scenes.se3_inverse(T_wc)
    T_iw = T_cw.copy(); T_iw[0, 3] += 0.02   # colour camera 2 cm beside the depth camera
    roo.SdfFuseColor(vol, cvol, f, nrm, T_cw, K, rgb, T_iw, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
torch.cuda.synchronize()
hv = hashlib.sha256(vol.storage.cpu().numpy().tobytes()).hexdigest()
hc = hashlib.sha256(cvol.storage.cpu().numpy().tobytes()).hexdigest()
touched = int((cvol.tensor() != 0.5).sum())
print("DIGEST", hv, hc, touched)
"""


@pytest.mark.parametrize("math", ["exact", "fast"])
def test_gpu_colour_fusion_full_size_tiled_equals_untiled(math, tmp_path):
    """BASELINE size (512^3, 640x480), three frames with a displaced colour camera: the LDS-tiled colour fusion and the
    global-gather kernel (KFX_FUSE_TILED=0) must leave byte-identical SDF and colour volumes (both read the same texels:
    the tile is a copy), in either numerics mode."""
    import subprocess
    import sys
    script = tmp_path / "colour_full.py"
    script.write_text(_COLOUR_FULL_SIZE)
    digests = []
    for tiled in ("1", "0"):
        out = subprocess.run([sys.executable, str(script), T.ROOT, math], env=dict(os.environ, KFX_FUSE_TILED=tiled),
                             capture_output=True, text=True, timeout=900)
        line = [l for l in out.stdout.splitlines() if l.startswith("DIGEST")]
        assert out.returncode == 0 and line, out.stdout[-2000:] + out.stderr[-2000:]
        digests.append(line[0].split()[1:])
    assert digests[0] == digests[1], digests
    assert int(digests[0][2]) > 10_000_000
