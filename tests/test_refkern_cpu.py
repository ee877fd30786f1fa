"""Third witness for the kernels' loop scaffolding (round-3 verdict item 6) -- a CROSS-CHECK, not the pin.

oracle/_ref/libkfx_refkern.so (make -C oracle refkern) is the TEXT of the reference's five __global__ kernels
(cu_sdffusion.cu:16-53, cu_raycast.cu:14-28 + 34-104, cu_bilateral.cu:59-92, cu_normals.cu:12-38, cu_depth_tools.cu:59-70),
pulled by line range into a temporary translation unit and compiled on the host unchanged, driven over the reference's own
launch shapes.  It must agree bit for bit with (a) the committed goldens, which oracle/ref_harness.cpp produced from the
reference's headers, (b) that harness on fresh inputs, and (c) the plain-C restatement oracle/kfx_oracle.c -- three
independent readings of the same loops.  Build-container only: skipped where /root/reference (and so the library) is absent."""
import ctypes as C
import os

import numpy as np
import pytest

import kfx_testlib as T
from kfx_testlib import oracle, scenes
from test_oracle_cpu import load_golden

KERN_SO = os.path.join(T.ROOT, "oracle", "_ref", "libkfx_refkern.so")
HDR_SO = os.path.join(T.ROOT, "oracle", "_ref", "libkfx_refhdr.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(KERN_SO) and os.path.isdir("/root/reference")),
                                reason="oracle/_ref/libkfx_refkern.so not built (needs /root/reference: make -C oracle refkern)")
PF = C.POINTER(C.c_float)


def fp(a):
    a = np.ascontiguousarray(a, np.float32).reshape(-1)
    return a, a.ctypes.data_as(PF)


def kern():
    return C.CDLL(KERN_SO)


def kern_fuse(R, vol, f, nrm, T_cw, K, tr, max_w, mincos):
    _t, t = fp(T_cw)
    _k, k = fp(K)
    R.refkern_sdf_fuse(vol.ref(), f.ref(), nrm.ref(), t, k, C.c_float(tr), C.c_float(max_w), C.c_float(mincos))


def kern_raycast(R, vol, w, h, T_wc, K, near, far, tr, subpix):
    rd, rn, ri = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
    _t, t = fp(T_wc)
    _k, k = fp(K)
    R.refkern_raycast_sdf(rd.ref(), rn.ref(), ri.ref(), vol.ref(), t, k, C.c_float(near), C.c_float(far), C.c_float(tr), 1 if subpix else 0)
    return rd, rn, ri


@pytest.mark.parametrize("name", ["room32_3frames", "full32_holes", "room_ragged_roi"])
def test_reference_kernel_text_reproduces_the_chain_goldens(name):
    R = kern()
    z, m = load_golden(name)
    dims = m["dims"]
    K = np.array(m["K"], np.float32)
    vk = oracle.Volume(dims[0], dims[1], dims[2], m["boxmin"], m["boxmax"])
    vo = oracle.Volume(dims[0], dims[1], dims[2], m["boxmin"], m["boxmax"])
    oracle.sdf_reset(vk, float("nan"))
    oracle.sdf_reset(vo, float("nan"))
    for i in range(m["n_frames"]):
        f = oracle.Image.from_numpy(z["filtered_%d" % i])
        nrm = oracle.Image.from_numpy(z["normals_%d" % i])
        T_cw = scenes.se3_inverse(z["poses"][i])
        wk, wo = vk, vo
        if "roi_frustum_%d" % i in z:   # the application's SubBoundingVolume views (dims not multiples of 8: quirk Q1)
            fr = z["roi_frustum_%d" % i]
            wk = oracle.sub_bounding_volume(vk, fr[:3], fr[3:])
            wo = oracle.sub_bounding_volume(vo, fr[:3], fr[3:])
        kern_fuse(R, wk, f, nrm, T_cw, K, m["trunc"], m["max_w"], m["mincostheta"])
        oracle.sdf_fuse(wo, f, nrm, T_cw, K, m["trunc"], m["max_w"], m["mincostheta"])
        assert T.nan_equal(vk.data, vo.data), "frame %d: kernel text vs restatement: %s" % (i, T.mismatch_report(vk.data, vo.data))
    assert T.nan_equal(vk.data, z["volume"]), T.mismatch_report(vk.data, z["volume"])   # golden: producer = reference headers
    w, h = m["w"], m["h"]
    rd, rn, ri = kern_raycast(R, vk, w, h, z["poses"][-1], K, m["near"], m["far"], m["trunc"], m["subpix"])
    assert T.nan_equal(rd.data, z["ray_depth"]), T.mismatch_report(rd.data, z["ray_depth"])
    assert T.nan_equal(rn.data, z["ray_norm"]), T.mismatch_report(rn.data, z["ray_norm"])
    assert T.nan_equal(ri.data, z["ray_img"]), T.mismatch_report(ri.data, z["ray_img"])
    od, on, oi = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
    oracle.raycast_sdf(od, on, oi, vo, z["poses"][-1], K, m["near"], m["far"], m["trunc"], m["subpix"])
    assert T.nan_equal(rd.data, od.data) and T.nan_equal(rn.data, on.data) and T.nan_equal(ri.data, oi.data)


def test_reference_kernel_text_preprocess_matches_golden_and_restatement():
    R = kern()
    z, m = load_golden("room32_3frames")
    K = np.array(m["K"], np.float32)
    _k, k = fp(K)
    b = m["bilateral"]
    w, h = m["w"], m["h"]
    for i in range(m["n_frames"]):
        raw = oracle.Image.from_numpy(z["raw_%d" % i])
        f, vbo, nrm = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h, channels=4)
        R.refkern_bilateral_f32(f.ref(), raw.ref(), C.c_float(b["gs"]), C.c_float(b["gr"]), C.c_uint(b["size"]), C.c_float(b["minval"]))
        R.refkern_depth_to_vbo_f32(vbo.ref(), f.ref(), k, C.c_float(1.0))
        R.refkern_normals_from_vbo(nrm.ref(), vbo.ref())
        assert T.nan_equal(f.data, z["filtered_%d" % i]), T.mismatch_report(f.data, z["filtered_%d" % i])
        assert T.nan_equal(vbo.data, z["vbo_%d" % i])
        assert T.nan_equal(nrm.data, z["normals_%d" % i])
        of, ovbo, onrm = T.preprocess_oracle(z["raw_%d" % i], K, b)
        assert T.nan_equal(f.data, of.data) and T.nan_equal(vbo.data, ovbo.data) and T.nan_equal(nrm.data, onrm.data)


@pytest.mark.skipif(not os.path.exists(HDR_SO), reason="oracle/_ref/libkfx_refhdr.so not built")
def test_reference_kernel_text_vs_header_harness_fresh_seed():
    """Inputs that are in no fixture (other sizes, noisy depth, padded pitches): the kernel text against the header harness
    (whose loops are this repo's) and against the restatement -- SdfFuse over several frames, then all three raycast images."""
    R, H = kern(), C.CDLL(HDR_SO)
    H.ref_sdf_fuse.restype = C.c_uint64
    w, h, dims = 88, 66, (48, 40, 56)
    K = scenes.intrinsics(w, h)
    _k, k = fp(K)
    bmin, bmax, near, far = scenes.SCENES["room"]
    tr = scenes.trunc_dist(bmin, bmax, dims)
    vols = {n: oracle.Volume(dims[0], dims[1], dims[2], bmin, bmax, pitch_bytes=dims[0] * 8 + 64) for n in ("kern", "hdr", "c")}
    for v in vols.values():
        oracle.sdf_reset(v, float("nan"))
    T_wc = None
    for i in (2, 3, 7):
        T_wc = scenes.orbit_pose(i, 12, yaw_deg=8.0, trans=0.07)
        raw = scenes.render_depth("room", w, h, T_wc, K, noise_sigma=0.002, seed=311 + i)
        f, vbo, nrm = T.preprocess_oracle(raw, K)
        T_cw = scenes.se3_inverse(T_wc)
        _t, t = fp(T_cw)
        kern_fuse(R, vols["kern"], f, nrm, T_cw, K, tr, 1000.0, 0.1)
        H.ref_sdf_fuse(vols["hdr"].ref(), f.ref(), nrm.ref(), t, k, C.c_float(tr), C.c_float(1000.0), C.c_float(0.1), 0)
        oracle.sdf_fuse(vols["c"], f, nrm, T_cw, K, tr, 1000.0, 0.1)
    assert T.nan_equal(vols["kern"].data, vols["hdr"].data), T.mismatch_report(vols["kern"].data, vols["hdr"].data)
    assert T.nan_equal(vols["kern"].data, vols["c"].data), T.mismatch_report(vols["kern"].data, vols["c"].data)
    for subpix in (True, False):
        rd, rn, ri = kern_raycast(R, vols["kern"], w, h, T_wc, K, near, far, tr, subpix)
        hd, hn, hi = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
        _t, t = fp(T_wc)
        H.ref_raycast_geom(hd.ref(), hn.ref(), vols["hdr"].ref(), t, k, C.c_float(near), C.c_float(far), C.c_float(tr), 1 if subpix else 0)
        H.ref_raycast_shade(hi.ref(), hd.ref(), hn.ref(), k)
        od, on, oi = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
        oracle.raycast_sdf(od, on, oi, vols["c"], T_wc, K, near, far, tr, subpix)
        assert np.isfinite(rd.data).sum() > 0.2 * w * h
        for a, b_, c in ((rd, hd, od), (rn, hn, on), (ri, hi, oi)):
            assert T.nan_equal(a.data, b_.data), T.mismatch_report(a.data, b_.data)
            assert T.nan_equal(a.data, c.data), T.mismatch_report(a.data, c.data)
