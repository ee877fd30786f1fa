"""CPU tests of SURVEY 8(f) row f-3 (colour fusion / colour raycast): the oracle's restatement against the
reference's own headers (oracle/_ref) and the properties the colour volume must have."""
import ctypes as C
import os

import numpy as np
import pytest

import kfx_testlib as T
from kfx_testlib import oracle, scenes

REF_SO = os.path.join(T.ROOT, "oracle", "_ref", "libkfx_refhdr.so")
PF = C.POINTER(C.c_float)


def rgb_frame(w, h, seed=0):
    """Deterministic RGB test card: smooth gradients plus seeded noise, uint8 (h, w, 3)."""
    rng = np.random.default_rng(seed)
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    img = np.stack([(u * 255) // max(w - 1, 1), (v * 255) // max(h - 1, 1), ((u + v) * 3) % 256], -1).astype(np.int32)
    img += rng.integers(-20, 21, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def color_setup(N, w, h, cw, ch, dims=None, frames=2):
    """Fuse `frames` RGB-D frames with the oracle; the colour camera has its own size, intrinsics and a small
    offset from the depth camera (T_cd), as an RGB-D rig has."""
    dims = (N, N, N) if dims is None else dims
    bmin, bmax, near, far = scenes.SCENES["room"]
    K, Kimg = scenes.intrinsics(w, h), scenes.intrinsics(cw, ch)
    tr = scenes.trunc_dist(bmin, bmax, dims)
    vol = oracle.Volume(dims[0], dims[1], dims[2], bmin, bmax)
    cvol = oracle.ColorVolume(dims[0], dims[1], dims[2], bmin, bmax)
    oracle.sdf_reset(vol, float("nan"))
    oracle.color_reset(cvol)
    T_cd = np.array([[1, 0, 0, 0.025], [0, 1, 0, -0.003], [0, 0, 1, 0.002]], np.float32)   # colour <- depth camera
    inputs = []
    for i in range(frames):
        T_wc = scenes.orbit_pose(i, 8)
        f, vbo, nrm = T.preprocess_oracle(scenes.render_depth("room", w, h, T_wc, K), K)
        T_cw = scenes.se3_inverse(T_wc)
        T_iw = (np.vstack([T_cd, [0, 0, 0, 1]]) @ np.vstack([T_cw, [0, 0, 0, 1]]))[:3].astype(np.float32)
        rgb = oracle.Image(cw, ch, np.uint8, 3)
        rgb.data[...] = rgb_frame(cw, ch, seed=i)
        inputs.append(dict(f=f, nrm=nrm, T_cw=T_cw, T_iw=T_iw, rgb=rgb, T_wc=T_wc))
    return vol, cvol, K, Kimg, tr, near, far, inputs


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("dims,full", [((32, 32, 32), False), ((40, 24, 19), False), ((21, 18, 9), True)])
def test_oracle_colour_fusion_matches_reference_headers(dims, full):
    R = C.CDLL(REF_SO)
    R.ref_sdf_fuse_color.restype = C.c_uint64
    R.ref_sdf_fuse_color.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, PF, PF, C.c_void_p, PF, PF, C.c_float, C.c_float,
                                     C.c_float, C.c_int]
    R.ref_color_trilinear.restype = C.c_float
    R.ref_color_trilinear.argtypes = [C.c_void_p, PF]
    w, h, cw, ch = 80, 60, 96, 72
    out = []
    for which in ("oracle", "ref"):
        vol, cvol, K, Kimg, tr, near, far, inputs = color_setup(0, w, h, cw, ch, dims=dims)
        counts = []
        for fr in inputs:
            if which == "oracle":
                counts.append(oracle.sdf_fuse_color(vol, cvol, fr["f"], fr["nrm"], fr["T_cw"], K, fr["rgb"], fr["T_iw"], Kimg, tr, 1000.0,
                                                    0.1, full_extent=full))
            else:
                fp = lambda a: np.ascontiguousarray(a, np.float32).reshape(-1).ctypes.data_as(PF)
                keep = [np.ascontiguousarray(a, np.float32).reshape(-1) for a in (fr["T_cw"], K, fr["T_iw"], Kimg)]
                counts.append(R.ref_sdf_fuse_color(vol.ref(), cvol.ref(), fr["f"].ref(), fr["nrm"].ref(), keep[0].ctypes.data_as(PF),
                                                   keep[1].ctypes.data_as(PF), fr["rgb"].ref(), keep[2].ctypes.data_as(PF),
                                                   keep[3].ctypes.data_as(PF), tr, 1000.0, 0.1, 1 if full else 0))
        out.append((vol.data.copy(), cvol.data.copy(), counts, cvol))
    assert out[0][2] == out[1][2] and out[0][2][0] > 0
    assert T.nan_equal(out[0][0], out[1][0]) and T.nan_equal(out[0][1], out[1][1])
    # the colour sampler of the colour raycast
    rng = np.random.default_rng(1)
    bmin, bmax = np.array(scenes.SCENES["room"][0], np.float32), np.array(scenes.SCENES["room"][1], np.float32)
    rd, rn, ri = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
    vol, cvol = oracle.Volume(*dims, bmin, bmax), out[0][3]
    vol.data[...] = out[0][0]
    oracle.raycast_sdf_color(rd, rn, ri, vol, cvol, inputs[-1]["T_wc"], K, near, far, tr, True)
    hit = np.isfinite(rd.data)
    assert hit.any() and (ri.data[~hit] == 0).all()
    vs, us = np.nonzero(hit)
    for i in rng.choice(len(vs), size=min(200, len(vs)), replace=False):
        u, v, d = int(us[i]), int(vs[i]), rd.data[vs[i], us[i]]
        Twc = np.asarray(inputs[-1]["T_wc"], np.float32)
        ray_c = np.array([(np.float32(u) - K[2]) / K[0], (np.float32(v) - K[3]) / K[1], 1.0], np.float32)
        ray_w = np.array([np.float32(np.float32(np.float32(Twc[r, 0] * ray_c[0]) + np.float32(Twc[r, 1] * ray_c[1])) + np.float32(Twc[r, 2] * ray_c[2]))
                          for r in range(3)], np.float32)
        pos = (Twc[:, 3] + d * ray_w).astype(np.float32)
        assert R.ref_color_trilinear(cvol.ref(), pos.ctypes.data_as(PF)) == ri.data[v, u]


def test_colour_volume_properties():
    """Untouched cells keep the reset value 0.5; fused cells hold grey levels in [0,1]; a uniform RGB frame
    drives every fused cell to exactly that grey; the SDF part equals the grey-only fusion on the same extent."""
    N, w, h = 32, 80, 60
    vol, cvol, K, Kimg, tr, near, far, inputs = color_setup(N, w, h, w, h, frames=1)
    fr = inputs[0]
    fr["rgb"].data[...] = 51   # (51+51+51)/3/255 = 0.2
    n = oracle.sdf_fuse_color(vol, cvol, fr["f"], fr["nrm"], fr["T_cw"], K, fr["rgb"], fr["T_iw"], Kimg, tr, 1000.0, 0.1)
    touched = ~np.isnan(vol.data[..., 0])
    assert n == int(touched.sum()) > 0
    assert (cvol.data[..., 0][~touched] == 0.5).all()
    # (w*c + 0.5*0) / (w + 0): c up to the rounding of one multiply and one divide
    assert np.allclose(cvol.data[..., 0][touched], 0.2, rtol=3e-7, atol=0)
    # same SDF values as the grey fusion wherever both integrate (colour needs the voxel inside the RGB image too)
    ref = T.make_volume(N, "room")
    oracle.sdf_fuse(ref, fr["f"], fr["nrm"], fr["T_cw"], K, tr, 1000.0, 0.1, full_extent=True)
    assert T.nan_equal(vol.data[touched], ref.data[touched])
