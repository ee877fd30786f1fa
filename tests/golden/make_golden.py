#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ -- run in the BUILD container,
where /root/reference exists:

    make -C oracle all ref && python tests/golden/make_golden.py

Producers (recorded per array in each .npz under the key `producers`):
  * "ref"    -- the reference's own code compiled in place (oracle/_ref/libkfx_refhdr.so, see
                oracle/ref_harness.cpp): voxel positions, SE3/pinhole maths, bilinear/trilinear
                sampling, gradient stencil, SDF_t running average, ROI helpers; the bilateral and
                NormalsFromVbo loops on the reference's GetWithClampedRange / float4 operators;
                PhongShade compiled from the reference's own lines; SdfSphere on the reference's
                VoxelPositionInUnits / length / SDF_t(float).
  * "input"  -- synthetic inputs (kangaroo_amd/scenes.py).
Every producer="ref" array is also asserted equal to the plain-C restatement (oracle/kfx_oracle.c)
while the fixture is written, so a fixture can only be generated when oracle and reference agree.
The fixtures are data only: inputs and expected outputs.  No reference source text is stored.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from kangaroo_amd import scenes  # noqa: E402

REF_SO = os.path.join(ROOT, "oracle", "_ref", "libkfx_refhdr.so")
PF = C.POINTER(C.c_float)


def fp(a):
    a = np.ascontiguousarray(a, np.float32).reshape(-1)
    return a, a.ctypes.data_as(PF)


def load_ref():
    R = C.CDLL(REF_SO)
    R.ref_sdf_fuse.restype = C.c_uint64
    R.ref_trilinear.restype = C.c_float
    return R


def chain_fixture(R, name, scene, N, w, h, n_frames, dims=None, subpix=True, holes=False, roi=False):
    """Multi-frame fuse + raycast: volume and raycast geometry from the reference headers."""
    K = scenes.intrinsics(w, h)
    bmin, bmax, near, far = scenes.SCENES[scene]
    dims = dims or (N, N, N)
    vol = oracle.Volume(dims[0], dims[1], dims[2], bmin, bmax)
    oracle.sdf_reset(vol, float("nan"))
    tr = scenes.trunc_dist(bmin, bmax, dims)
    out = {}
    prod = {}
    poses, n_upd = [], []
    for i in range(n_frames):
        T_wc = scenes.orbit_pose(i, 8)
        raw = scenes.render_depth(scene, w, h, T_wc, K)
        if holes:  # invalid measurements: NaN blocks and a below-minval (0.2 m) stripe
            raw[10:20, 30:45] = np.nan
            raw[40:44, :] = 0.1
        d = oracle.Image.from_numpy(raw)
        f = oracle.Image(w, h)
        vbo = oracle.Image(w, h, channels=4)
        nrm = oracle.Image(w, h, channels=4)
        R.ref_bilateral_f32(f.ref(), d.ref(), C.c_float(1.5), C.c_float(0.1), 3, C.c_float(0.2), 1)
        _, k = fp(K)
        R.ref_depth_to_vbo(vbo.ref(), f.ref(), k, C.c_float(1.0))
        R.ref_normals_from_vbo(nrm.ref(), vbo.ref())
        of, ov, on_ = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h, channels=4)
        oracle.bilateral(of, d, 1.5, 0.1, 3, 0.2)
        oracle.depth_to_vbo(ov, of, K)
        oracle.normals_from_vbo(on_, ov)
        assert all(np.array_equal(a.data, b.data, equal_nan=True) for a, b in ((f, of), (vbo, ov), (nrm, on_))), \
            "oracle preprocess disagrees with the reference-header preprocess"
        T_cw = scenes.se3_inverse(T_wc)
        _, t = fp(T_cw)
        work = vol
        if roi:  # the application's ROI view: SubBoundingVolume(FitToFrustum), main.cpp:275-276
            lo, hi = (C.c_float * 3)(), (C.c_float * 3)()
            _, twc = fp(T_wc)
            R.ref_fit_to_frustum(lo, hi, twc, C.c_float(w), C.c_float(h), k, C.c_float(2.2), C.c_float(3.3))
            sub = oracle.KfoVolume()
            R.ref_sub_bounding_volume(C.byref(sub), vol.ref(), lo, hi)
            work = oracle.SubVolume(vol, sub)
            out["roi_origin_%d" % i] = np.array(work.origin, np.int32)
            out["roi_dims_%d" % i] = np.array([work.w, work.h, work.d], np.int32)
            out["roi_boxmin_%d" % i] = work.boxmin
            out["roi_boxmax_%d" % i] = work.boxmax
            out["roi_frustum_%d" % i] = np.array(list(lo) + list(hi), np.float32)
            for key in ("roi_origin_%d", "roi_dims_%d", "roi_boxmin_%d", "roi_boxmax_%d", "roi_frustum_%d"):
                prod[key % i] = "ref"
        n = R.ref_sdf_fuse(work.ref(), f.ref(), nrm.ref(), t, k, C.c_float(tr), C.c_float(scenes.MAX_W),
                           C.c_float(scenes.MIN_COS_THETA), 0)
        out["raw_%d" % i] = raw; prod["raw_%d" % i] = "input"
        out["filtered_%d" % i] = f.data.copy(); prod["filtered_%d" % i] = "ref"
        out["vbo_%d" % i] = vbo.data.copy(); prod["vbo_%d" % i] = "ref"
        out["normals_%d" % i] = nrm.data.copy(); prod["normals_%d" % i] = "ref"
        poses.append(T_wc)
        n_upd.append(int(n))
    out["volume"] = vol.data.copy(); prod["volume"] = "ref"
    # raycast from the last pose: geometry by the reference headers, shading by the reference's PhongShade
    T_wc = poses[-1]
    rd, rn = oracle.Image(w, h), oracle.Image(w, h, channels=4)
    _, t = fp(T_wc)
    _, k = fp(K)
    R.ref_raycast_geom(rd.ref(), rn.ref(), vol.ref(), t, k, C.c_float(near), C.c_float(far), C.c_float(tr),
                       1 if subpix else 0)
    od, on, oi = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
    st = oracle.raycast_sdf(od, on, oi, vol, T_wc, K, near, far, tr, subpix)
    ri = oracle.Image(w, h)
    R.ref_raycast_shade(ri.ref(), rd.ref(), rn.ref(), k)
    assert np.array_equal(od.data, rd.data, equal_nan=True) and np.array_equal(on.data, rn.data) and np.array_equal(oi.data, ri.data), \
        "oracle raycast disagrees with the reference-header raycast"
    out["ray_depth"] = rd.data.copy(); prod["ray_depth"] = "ref"
    out["ray_norm"] = rn.data.copy(); prod["ray_norm"] = "ref"
    out["ray_img"] = ri.data.copy(); prod["ray_img"] = "ref"
    out["poses"] = np.stack(poses); prod["poses"] = "input"
    meta = dict(scene=scene, dims=list(dims), w=w, h=h, n_frames=n_frames, K=K.tolist(), boxmin=list(bmin),
                boxmax=list(bmax), near=near, far=far, trunc=tr, max_w=scenes.MAX_W, mincostheta=scenes.MIN_COS_THETA,
                subpix=bool(subpix), n_updated=n_upd, raycast_stats={k_: int(v) for k_, v in st.items()},
                bilateral=scenes.BILATERAL)
    out["meta"] = np.array(json.dumps(meta))
    out["producers"] = np.array(json.dumps(prod))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %8.1f KiB  updated=%s hits=%d" % (name + ".npz", os.path.getsize(path) / 1024, n_upd, st["hits"]))


def sphere_fixture(R, name, N, w, h, trunc):
    """The reference's own synthetic test volume (examples/Raycast.cpp:58,66): SdfSphere + RaycastSdf."""
    K = scenes.intrinsics(w, h)
    vol = oracle.Volume(N, N, N, (-1, -1, -1), (1, 1, 1))
    oracle.sdf_reset(vol, float("nan"))
    # SdfSphere: the kernel body on the reference's VoxelPositionInUnits / length / SDF_t(float)
    _, cc = fp((0.05, -0.1, 0.0))
    R.ref_sdf_sphere(vol.ref(), cc, C.c_float(0.7))
    ovol = oracle.Volume(N, N, N, (-1, -1, -1), (1, 1, 1))
    oracle.sdf_reset(ovol, float("nan"))
    oracle.sdf_sphere(ovol, (0.05, -0.1, 0.0), 0.7)
    assert np.array_equal(ovol.data, vol.data, equal_nan=True), "oracle SdfSphere disagrees with the reference-header SdfSphere"
    pos = (C.c_float * 3)()
    for (x, y, z) in ((0, 0, 0), (N - 1, N - 1, N - 1), (5, 17, 9), (N // 2, 3, N - 2)):
        R.ref_voxel_position(vol.ref(), x, y, z, pos)
        p = np.array(list(pos), np.float32)
        c = np.array([0.05, -0.1, 0.0], np.float32)
        d = (p - c).astype(np.float32)
        dist = np.sqrt(np.float32(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]))
        assert vol.data[z, y, x, 0] == np.float32(dist - np.float32(0.7)), (x, y, z)
    T_wc = np.array([[1, 0, 0, 0.1], [0, 1, 0, -0.05], [0, 0, 1, -2.5]], np.float32)
    rd, rn = oracle.Image(w, h), oracle.Image(w, h, channels=4)
    _, t = fp(T_wc)
    _, k = fp(K)
    R.ref_raycast_geom(rd.ref(), rn.ref(), vol.ref(), t, k, C.c_float(0.1), C.c_float(10.0), C.c_float(trunc), 1)
    od, on, oi = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
    st = oracle.raycast_sdf(od, on, oi, vol, T_wc, K, 0.1, 10.0, trunc, True)
    ri = oracle.Image(w, h)
    R.ref_raycast_shade(ri.ref(), rd.ref(), rn.ref(), k)
    assert np.array_equal(od.data, rd.data, equal_nan=True) and np.array_equal(on.data, rn.data) and np.array_equal(oi.data, ri.data)
    meta = dict(dims=[N, N, N], w=w, h=h, K=K.tolist(), boxmin=[-1, -1, -1], boxmax=[1, 1, 1], near=0.1, far=10.0,
                trunc=trunc, center=[0.05, -0.1, 0.0], r=0.7, raycast_stats={k_: int(v) for k_, v in st.items()})
    prod = dict(volume="ref", ray_depth="ref", ray_norm="ref", ray_img="ref")
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, volume=vol.data.copy(), T_wc=T_wc, ray_depth=rd.data.copy(), ray_norm=rn.data.copy(),
                        ray_img=ri.data.copy(), meta=np.array(json.dumps(meta)), producers=np.array(json.dumps(prod)))
    print("%-28s %8.1f KiB  hits=%d" % (name + ".npz", os.path.getsize(path) / 1024, st["hits"]))


def helper_vectors(R, name):
    """Known-answer vectors of the reference's header helpers on seeded random inputs."""
    rng = np.random.default_rng(1234)
    out = {}
    # SDF_t::operator+= / LimitWeight
    n = 512
    a = rng.uniform(-0.05, 0.05, n).astype(np.float32)
    w = rng.uniform(0.01, 3.0, n).astype(np.float32)
    ov = rng.uniform(-0.05, 0.05, n).astype(np.float32)
    ow = rng.uniform(-1.0, 1200.0, n).astype(np.float32)
    ov[::7] = np.nan
    ow[::7] = 0.0
    res = np.zeros((n, 2), np.float32)
    o2 = (C.c_float * 2)()
    for i in range(n):
        R.ref_sdf_accumulate(C.c_float(a[i]), C.c_float(w[i]), C.c_float(ov[i]), C.c_float(ow[i]), C.c_float(1000.0), o2)
        res[i] = (o2[0], o2[1])
    out.update(acc_val=a, acc_w=w, acc_old_val=ov, acc_old_w=ow, acc_out=res)
    # trilinear / gradient at random positions (inside and outside the box) of a random volume
    N = 12
    vol = oracle.Volume(N, N + 1, N + 2, (-0.5, -0.25, 1.0), (0.75, 0.5, 2.5))
    vol.data[...] = rng.normal(0, 1, vol.data.shape).astype(np.float32)
    pos = rng.uniform([-0.7, -0.4, 0.8], [0.9, 0.7, 2.7], (400, 3)).astype(np.float32)
    tri = np.zeros(400, np.float32)
    grad = np.zeros((400, 3), np.float32)
    g3 = (C.c_float * 3)()
    for i in range(400):
        _, p = fp(pos[i])
        tri[i] = R.ref_trilinear(vol.ref(), p)
        R.ref_backward_diff(vol.ref(), p, g3)
        grad[i] = list(g3)
    out.update(samp_volume=vol.data.copy(), samp_boxmin=vol.boxmin, samp_boxmax=vol.boxmax, samp_pos=pos,
               samp_trilinear=tri, samp_gradient=grad)
    # SE3inv, pyramid intrinsics
    T = scenes.orbit_pose(3, 8)
    o12 = (C.c_float * 12)()
    _, t = fp(T)
    R.ref_se3_inverse(o12, t)
    out.update(se3_in=T, se3_out=np.array(list(o12), np.float32).reshape(3, 4))
    K = scenes.intrinsics(640, 480)
    _, k = fp(K)
    o4 = (C.c_float * 4)()
    lv = []
    for l in range(4):
        R.ref_intrinsics_level(o4, k, l)
        lv.append(list(o4))
    out.update(K=K, K_levels=np.array(lv, np.float32))
    sz = (C.c_size_t * 8)()
    R.ref_sizeof(sz)
    out["sizeof"] = np.array(list(sz), np.int64)  # Image, Volume, BoundedVolume, SDF_t, Mat3x4, Intrinsics, BoundingBox, float4
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %8.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024))


def next_rows_fixture(R, name):
    """SURVEY 8(f) rows, all from the reference's own headers (oracle/ref_harness.cpp): the ICP normal equations
    (per-block and summed), two frames of colour fusion, and the bytes of a saved volume file."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_color_cpu as TC
    import test_tracking_cpu as TT
    from kangaroo_amd import tracking
    out = {}
    # ---- ICP ----
    R.ref_icp_point_plane.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, PF, PF, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    R.ref_icp_point_plane.restype = None
    w, h = 80, 60
    K, Pl, Pr, Nr, _, _ = TT.icp_inputs("room", w, h, True)
    T_lp = tracking.se3_exp([0.004, -0.003, 0.002, 0.003, -0.002, 0.001])
    KT = (tracking.k_matrix(K) @ T_lp[:3]).astype(np.float32)
    T_pl = tracking.se3_inv(T_lp)[:3].astype(np.float32)
    dbg = oracle.Image(w, h, channels=4)
    lss, blocks = oracle.icp_point_plane(Pl, Pr, Nr, KT, T_pl, 0.1, dbg, want_blocks=True, fn=R.ref_icp_point_plane)
    out.update(icp_Pl=Pl.data.copy(), icp_Pr=Pr.data.copy(), icp_Nr=Nr.data.copy(), icp_KT=KT, icp_T_pl=T_pl, icp_c=np.float32(0.1),
               icp_lss=np.frombuffer(lss.tobytes(), np.uint8).copy(), icp_blocks=np.frombuffer(blocks.tobytes(), np.uint8).copy(),
               icp_debug=dbg.data.copy())
    # ---- colour fusion ----
    R.ref_sdf_fuse_color.restype = C.c_uint64
    R.ref_sdf_fuse_color.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, PF, PF, C.c_void_p, PF, PF, C.c_float, C.c_float,
                                     C.c_float, C.c_int]
    dims, cw, ch = (24, 20, 18), 96, 72
    vol, cvol, K, Kimg, tr, near, far, inputs = TC.color_setup(0, w, h, cw, ch, dims=dims)
    counts = []
    for i, fr in enumerate(inputs):
        keep = [np.ascontiguousarray(a, np.float32).reshape(-1) for a in (fr["T_cw"], K, fr["T_iw"], Kimg)]
        counts.append(R.ref_sdf_fuse_color(vol.ref(), cvol.ref(), fr["f"].ref(), fr["nrm"].ref(), keep[0].ctypes.data_as(PF),
                                           keep[1].ctypes.data_as(PF), fr["rgb"].ref(), keep[2].ctypes.data_as(PF),
                                           keep[3].ctypes.data_as(PF), tr, 1000.0, 0.1, 0))
        out["col_depth%d" % i] = fr["f"].data.copy()
        out["col_normals%d" % i] = fr["nrm"].data.copy()
        out["col_rgb%d" % i] = fr["rgb"].data.copy()
        out["col_T_cw%d" % i] = fr["T_cw"]
        out["col_T_iw%d" % i] = fr["T_iw"]
    out.update(col_K=K, col_Kimg=Kimg, col_trunc=np.float32(tr), col_dims=np.array(dims), col_counts=np.array(counts),
               col_volume=vol.data.copy(), col_colour=cvol.data.copy())
    # ---- SavePXM bytes ----
    import tempfile
    R.ref_save_pxm.argtypes = [C.c_char_p, C.c_void_p, C.c_int]
    small = oracle.Volume(6, 5, 4, (-1.25, -0.333333343, 2.0), (1.0, 0.1, 4.000001), pitch_bytes=6 * 8 + 16)
    small.data[...] = np.random.default_rng(5).standard_normal(small.data.shape).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "v.vol")
        assert R.ref_save_pxm(fn.encode(), small.ref(), 8) == 0
        out.update(pxm_volume=small.data.copy(), pxm_boxmin=small.boxmin, pxm_boxmax=small.boxmax,
                   pxm_bytes=np.frombuffer(open(fn, "rb").read(), np.uint8).copy())
    # ---- frame pre-amble (row f-1): mm -> m scale-bias and the NaN-aware 2x2 reduction, reference header arithmetic ----
    rng = np.random.default_rng(11)
    pw, ph = 70, 50
    mm = rng.uniform(300, 6000, (ph, pw)).astype(np.float32)
    mm[3:6, 9:20] = np.nan
    mm[20:22, 30:32] = np.nan          # one fully invalid 2x2 cell
    mm[41, 7] = np.inf
    src, metres, half = oracle.Image.from_numpy(mm), oracle.Image(pw, ph), oracle.Image(pw // 2, ph // 2)
    R.ref_elementwise_scale_bias_f32(metres.ref(), src.ref(), C.c_float(1.0 / 1000.0), C.c_float(0.0))
    R.ref_box_half_ignore_invalid_f32(half.ref(), metres.ref())
    out.update(pre_mm=mm, pre_metres=metres.data.copy(), pre_half=half.data.copy())
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %8.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024))


if __name__ == "__main__":
    if not os.path.exists(REF_SO):
        sys.exit("build oracle/_ref first: make -C oracle ref")
    R = load_ref()
    chain_fixture(R, "room32_3frames", "room", 32, 80, 60, 3)
    chain_fixture(R, "full32_holes", "full", 32, 80, 60, 2, holes=True)
    chain_fixture(R, "room_ragged_roi", "room", 0, 80, 60, 2, dims=(40, 36, 44), roi=True, subpix=False)
    sphere_fixture(R, "sphere32_trunc0", 32, 64, 48, 0.0)
    helper_vectors(R, "ref_helper_vectors")
    next_rows_fixture(R, "ref_next_rows")
