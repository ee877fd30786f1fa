"""CPU tests of the parity oracle: it must reproduce every committed golden vector (generated
from the reference's own header code, tests/golden/make_golden.py) bit for bit, and -- when
oracle/_ref is present (build container / GPU box snapshot) -- agree with the reference
headers on fresh seeded inputs."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import kfx_testlib as T
from kfx_testlib import oracle, scenes

REF_SO = os.path.join(T.ROOT, "oracle", "_ref", "libkfx_refhdr.so")


def load_golden(name):
    z = np.load(os.path.join(T.GOLDEN, name + ".npz"))
    return z, json.loads(str(z["meta"])) if "meta" in z else None


def replay_chain(z, m, fuse=None, raycast=None):
    """Re-run fuse+raycast from the fixture's inputs with the oracle (or given callables)."""
    dims = m["dims"]
    vol = oracle.Volume(dims[0], dims[1], dims[2], m["boxmin"], m["boxmax"])
    oracle.sdf_reset(vol, float("nan"))
    K = np.array(m["K"], np.float32)
    for i in range(m["n_frames"]):
        f = oracle.Image.from_numpy(z["filtered_%d" % i])
        nrm = oracle.Image.from_numpy(z["normals_%d" % i])
        T_cw = scenes.se3_inverse(z["poses"][i])
        work = vol
        if "roi_frustum_%d" % i in z:
            fr = z["roi_frustum_%d" % i]
            work = oracle.sub_bounding_volume(vol, fr[:3], fr[3:])
            assert list(work.origin) == z["roi_origin_%d" % i].tolist()
            assert [work.w, work.h, work.d] == z["roi_dims_%d" % i].tolist()
            assert np.array_equal(work.boxmin, z["roi_boxmin_%d" % i])
            assert np.array_equal(work.boxmax, z["roi_boxmax_%d" % i])
        n = oracle.sdf_fuse(work, f, nrm, T_cw, K, m["trunc"], m["max_w"], m["mincostheta"])
        assert n == m["n_updated"][i]
    return vol, K


@pytest.mark.parametrize("name", ["room32_3frames", "full32_holes", "room_ragged_roi"])
def test_oracle_reproduces_golden_chain(name):
    z, m = load_golden(name)
    vol, K = replay_chain(z, m)
    assert T.nan_equal(vol.data, z["volume"]), T.mismatch_report(vol.data, z["volume"])
    w, h = m["w"], m["h"]
    rd, rn, ri = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
    st = oracle.raycast_sdf(rd, rn, ri, vol, z["poses"][-1], K, m["near"], m["far"], m["trunc"], m["subpix"])
    assert T.nan_equal(rd.data, z["ray_depth"])
    assert T.nan_equal(rn.data, z["ray_norm"])
    assert T.nan_equal(ri.data, z["ray_img"])
    assert st == m["raycast_stats"]


def test_oracle_preprocess_matches_golden():
    z, m = load_golden("room32_3frames")
    K = np.array(m["K"], np.float32)
    b = m["bilateral"]
    for i in range(m["n_frames"]):
        f, vbo, nrm = T.preprocess_oracle(z["raw_%d" % i], K, b)
        assert T.nan_equal(f.data, z["filtered_%d" % i])
        assert T.nan_equal(vbo.data, z["vbo_%d" % i])      # producer: reference Unproject
        assert T.nan_equal(nrm.data, z["normals_%d" % i])


def test_oracle_sphere_golden():
    z, m = load_golden("sphere32_trunc0")
    N = m["dims"][0]
    vol = oracle.Volume(N, N, N, m["boxmin"], m["boxmax"])
    oracle.sdf_reset(vol, float("nan"))
    oracle.sdf_sphere(vol, m["center"], m["r"])
    assert T.nan_equal(vol.data, z["volume"])
    w, h = m["w"], m["h"]
    rd, rn, ri = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
    oracle.raycast_sdf(rd, rn, ri, vol, z["T_wc"], np.array(m["K"], np.float32), m["near"], m["far"], m["trunc"], True)
    assert T.nan_equal(rd.data, z["ray_depth"]) and T.nan_equal(rn.data, z["ray_norm"]) and T.nan_equal(ri.data, z["ray_img"])
    # analytic check: hit depth close to the true sphere (|c - o| - r along the central ray)
    hits = np.isfinite(rd.data)
    assert hits.sum() == m["raycast_stats"]["hits"] > 500


def test_oracle_helper_vectors():
    z, _ = load_golden("ref_helper_vectors")
    # layouts the C-ABI mirrors: Image 32, Volume 48, BoundedVolume 72, SDF_t 8, Mat3x4 48, K 16, BBox 24, float4 16
    assert z["sizeof"].tolist() == [32, 48, 72, 8, 48, 16, 24, 16]
    assert C.sizeof(oracle.KfoImage) == 32 and C.sizeof(oracle.KfoVolume) == 72
    for i in range(len(z["acc_val"])):
        got = oracle.sdf_accumulate(z["acc_val"][i], z["acc_w"][i], z["acc_old_val"][i], z["acc_old_w"][i], 1000.0)
        assert T.nan_equal(got, z["acc_out"][i]), i
    v = z["samp_volume"]
    vol = oracle.Volume(v.shape[2], v.shape[1], v.shape[0], z["samp_boxmin"], z["samp_boxmax"])
    vol.data[...] = v
    for i, p in enumerate(z["samp_pos"]):
        assert np.float32(oracle.trilinear(vol, p)) == z["samp_trilinear"][i], i
        assert np.array_equal(oracle.gradient(vol, p), z["samp_gradient"][i]), i
    assert np.array_equal(oracle.se3_inverse(z["se3_in"]), z["se3_out"])
    assert np.array_equal(scenes.se3_inverse(z["se3_in"]), z["se3_out"])
    for l in range(4):
        assert np.array_equal(oracle.intrinsics_level(z["K"], l), z["K_levels"][l])
        assert np.array_equal(scenes.intrinsics_level(z["K"], l), z["K_levels"][l])


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_vs_reference_headers_fresh_seed():
    """Bit-equality of the restatement with the reference's compiled headers on inputs that are
    not in the committed fixtures (different size, poses, padded pitches)."""
    R = C.CDLL(REF_SO)
    R.ref_sdf_fuse.restype = C.c_uint64
    PF = C.POINTER(C.c_float)

    def fp(a):
        a = np.ascontiguousarray(a, np.float32).reshape(-1)
        return a, a.ctypes.data_as(PF)

    w, h, dims = 96, 72, (48, 40, 56)
    K = scenes.intrinsics(w, h)
    bmin, bmax, near, far = scenes.SCENES["room"]
    tr = scenes.trunc_dist(bmin, bmax, dims)
    vols = []
    for which in ("oracle", "ref"):
        vol = oracle.Volume(dims[0], dims[1], dims[2], bmin, bmax, pitch_bytes=dims[0] * 8 + 64)
        oracle.sdf_reset(vol, float("nan"))
        for i in (1, 5, 6):
            T_wc = scenes.orbit_pose(i, 12, yaw_deg=9.0, trans=0.08)
            raw = scenes.render_depth("room", w, h, T_wc, K, noise_sigma=0.002, seed=77 + i)
            f, vbo, nrm = T.preprocess_oracle(raw, K)
            T_cw = scenes.se3_inverse(T_wc)
            if which == "oracle":
                oracle.sdf_fuse(vol, f, nrm, T_cw, K, tr, 1000.0, 0.1, full_extent=True)
            else:
                _, t = fp(T_cw)
                _, k = fp(K)
                R.ref_sdf_fuse(vol.ref(), f.ref(), nrm.ref(), t, k, C.c_float(tr), C.c_float(1000.0), C.c_float(0.1), 1)
        rd, rn, ri = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
        if which == "oracle":
            oracle.raycast_sdf(rd, rn, ri, vol, T_wc, K, near, far, tr, True)
        else:
            _, t = fp(T_wc)
            _, k = fp(K)
            R.ref_raycast_geom(rd.ref(), rn.ref(), vol.ref(), t, k, C.c_float(near), C.c_float(far), C.c_float(tr), 1)
        vols.append((vol.data.copy(), rd.data.copy(), rn.data.copy()))
    assert T.nan_equal(vols[0][0], vols[1][0]), T.mismatch_report(vols[0][0], vols[1][0])
    assert T.nan_equal(vols[0][1], vols[1][1])
    assert T.nan_equal(vols[0][2], vols[1][2])


def test_oracle_threads_do_not_change_results():
    vol1, vol2 = T.make_volume(32, "room"), T.make_volume(32, "room")
    T.fuse_frames_oracle(vol1, "room", 80, 60, 2, nthreads=1)
    T.fuse_frames_oracle(vol2, "room", 80, 60, 2, nthreads=4)
    assert T.nan_equal(vol1.data, vol2.data)


def test_oracle_quirks():
    # Q1: dims not multiple of 8 leave the tail untouched unless full_extent
    vol = T.make_volume(0, "full", dims=(20, 17, 13))
    K, tr, fr = T.fuse_frames_oracle(vol, "full", 80, 60, 1)
    upd = ~np.isnan(vol.data[..., 0])
    assert upd[:8, :16, :16].all() and not upd[8:].any() and not upd[:, 16:].any() and not upd[:, :, 16:].any()
    vol2 = T.make_volume(0, "full", dims=(20, 17, 13))
    T.fuse_frames_oracle(vol2, "full", 80, 60, 1, full_extent=True)
    assert (~np.isnan(vol2.data[..., 0])).all()
    # Q3: first observation overwrites the (NaN, 0) state; second averages and adds weights
    v = T.make_volume(16, "full")
    K, tr, f1 = T.fuse_frames_oracle(v, "full", 80, 60, 1, n_orbit=1)
    a = v.data.copy()
    T.fuse_frames_oracle(v, "full", 80, 60, 1, n_orbit=1)
    b = v.data
    assert np.allclose(b[..., 1], 2 * a[..., 1], rtol=1e-6) and np.allclose(b[..., 0], a[..., 0], atol=1e-6)
    # SdfReset fills the pitch padding too (thrust::fill over the contiguous span)
    pv = oracle.Volume(10, 9, 8, pitch_bytes=10 * 8 + 48)
    oracle.sdf_reset(pv, 0.25)
    raw = pv.raw.view(np.float32)
    span = ((pv.d - 1) * pv.img_pitch + (pv.h - 1) * pv.pitch + pv.w * 8) // 4
    assert (raw[0:span:2] == 0.25).all() and (raw[1:span:2] == 0).all() and (raw[span:] == 0).all()


def test_oracle_raycast_recovers_scene_depth():
    """Fuse then raycast reproduces the input depth (the reference's own visual check,
    applications/examples/SdfFusion.cpp:130-135): mean |d - gt| well below a voxel."""
    vol = T.make_volume(64, "room")
    K, tr, fr = T.fuse_frames_oracle(vol, "room", 160, 120, 1, n_orbit=1)
    rd, rn, ri = oracle.Image(160, 120), oracle.Image(160, 120, channels=4), oracle.Image(160, 120)
    st = oracle.raycast_sdf(rd, rn, ri, vol, fr[0]["T_wc"], K, 0.4, 8.0, tr, True)
    err = np.abs(rd.data - fr[0]["raw"])
    m = np.isfinite(err)
    assert st["hits"] > 0.5 * 160 * 120 and np.median(err[m]) < 0.5 * vol.voxel_size()[0]
    n = rn.data[np.isfinite(rd.data)]
    assert np.allclose(np.linalg.norm(n[:, :3], axis=1), 1.0, atol=1e-5) and (n[:, 3] == 1).all()
    miss = ~np.isfinite(rd.data)
    assert (rn.data[miss] == 0).all() and (ri.data[miss] == 0).all()


def test_analytic_renderers_and_roi():
    w, h = 64, 48
    K = scenes.intrinsics(w, h)
    Tid = scenes.identity_pose()
    d = oracle.Image(w, h)
    oracle.raycast_box(d, Tid, K, (-0.5, -0.5, 2.0), (0.5, 0.5, 3.0))
    c = d.data[h // 2, w // 2]
    assert c == np.float32(2.0) and np.isnan(d.data[0, 0])
    oracle.raycast_sphere(d, None, Tid, K, (0, 0, 2.0), 0.25)
    assert abs(d.data[h // 2, w // 2] - 1.75) < 1e-3
    lo, hi = oracle.fit_to_frustum(Tid, w, h, K, 0.4, 4.0)
    assert lo[2] == np.float32(0.4) and hi[2] == np.float32(4.0) and lo[0] < 0 < hi[0]
    vol = oracle.Volume(32, 32, 32, (-1, -1, 0), (1, 1, 4))
    sub = oracle.sub_bounding_volume(vol, (-0.3, -0.2, 1.0), (0.4, 0.9, 2.0))
    x0, y0, z0 = sub.origin
    assert np.array_equal(sub.boxmin, oracle.voxel_position(vol, x0, y0, z0))
    assert np.array_equal(sub.boxmax, oracle.voxel_position(vol, x0 + sub.w - 1, y0 + sub.h - 1, z0 + sub.d - 1))
    assert sub.boxmin[0] <= -0.3 + 2 / 31 and sub.boxmax[0] >= 0.4


def test_oracle_half_cells():
    """fp16 cells: F16C round-to-nearest-even conversions; first observation stores half(val), the
    running average rounds every intermediate (Sdf.h:52-58)."""
    v = oracle.VolumeH(16, 16, 16, *scenes.SCENES["full"][:2])
    oracle.sdf_reset(v, float("nan"))
    assert np.isnan(v.data[..., 0]).all() and (v.data[..., 1] == 0).all()
    K = scenes.intrinsics(80, 60)
    f, vbo, nrm = T.preprocess_oracle(scenes.render_depth("full", 80, 60, None, K), K)
    tr = scenes.trunc_dist(v.boxmin, v.boxmax, (16, 16, 16))
    v32 = T.make_volume(16, "full")
    n16 = oracle.sdf_fuse(v, f, nrm, scenes.identity_pose(), K, tr, 1000.0, 0.1)
    n32 = oracle.sdf_fuse(v32, f, nrm, scenes.identity_pose(), K, tr, 1000.0, 0.1)
    assert n16 == n32 > 0
    assert np.array_equal(v.data, v32.data.astype(np.float16), equal_nan=True)  # first observation = rounding only
    oracle.sdf_fuse(v, f, nrm, scenes.identity_pose(), K, tr, 1000.0, 0.1)
    w1 = v32.data[..., 1].astype(np.float16)
    upd = ~np.isnan(v.data[..., 0])
    assert np.array_equal(v.data[..., 1][upd], (w1[upd] + w1[upd]).astype(np.float16))
    rd, rn, ri = oracle.Image(80, 60), oracle.Image(80, 60, channels=4), oracle.Image(80, 60)
    st = oracle.raycast_sdf(rd, rn, ri, v, scenes.identity_pose(), K, 0.4, 8.0, tr, True)
    assert st["rays"] > 0


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_analytic_renderers_and_sdf_distance_vs_reference_headers():
    """RaycastBox / RaycastSphere / RaycastPlane depths and SdfDistance: the restatement against the same per-pixel
    arithmetic evaluated with the reference's header functions (first-hit depths: images start empty)."""
    R = C.CDLL(REF_SO)
    PF = C.POINTER(C.c_float)
    w, h = 96, 72
    K = scenes.intrinsics(w, h)
    T_wc = scenes.orbit_pose(3, 8)
    fp = lambda a: np.ascontiguousarray(a, np.float32).reshape(-1)
    t, k = fp(T_wc), fp(K)
    bmin, bmax, center, n_w, r = fp((-0.5, -0.4, 2.0)), fp((0.6, 0.5, 3.0)), fp((0.1, 0.0, 3.0)), fp((0.0, 0.1, -1.0 / 3.8)), 0.5
    rb, rs, rp = oracle.Image(w, h), oracle.Image(w, h), oracle.Image(w, h)
    R.ref_analytic_depths(rb.ref(), rs.ref(), rp.ref(), t.ctypes.data_as(PF), k.ctypes.data_as(PF), bmin.ctypes.data_as(PF),
                          bmax.ctypes.data_as(PF), center.ctypes.data_as(PF), C.c_float(r), n_w.ctypes.data_as(PF))
    ob = oracle.Image(w, h)
    oracle.raycast_box(ob, T_wc, K, bmin, bmax)
    assert T.nan_equal(ob.data, rb.data)
    for ref_img, fn in ((rs, lambda d: oracle.raycast_sphere(d, None, T_wc, K, center, r)), (rp, lambda d: oracle.raycast_plane(d, None, T_wc, K, n_w))):
        d = oracle.Image(w, h)
        d.data[...] = np.nan
        fn(d)
        want = np.where(ref_img.data > 0, ref_img.data, np.float32("nan"))   # the kernels keep only positive depths
        assert T.nan_equal(d.data, want)
    vol = T.make_volume(32, "room")
    T.fuse_frames_oracle(vol, "room", w, h, 2)
    depth = oracle.Image(w, h)
    depth.data[...] = scenes.render_depth("room", w, h, T_wc, K)
    depth.data[::9, ::7] = np.nan
    od, rd = oracle.Image(w, h), oracle.Image(w, h)
    oracle.sdf_distance(od, depth, vol, T_wc, K)
    R.ref_sdf_distance(rd.ref(), depth.ref(), vol.ref(), t.ctypes.data_as(PF), k.ctypes.data_as(PF))
    assert T.nan_equal(od.data, rd.data) and np.isfinite(od.data).any()


def depth_tool_inputs(w=96, h=72, cw=80, ch=64, seed=3):
    rng = np.random.default_rng(seed)
    K = scenes.intrinsics(w, h)
    depth = scenes.render_depth("room", w, h, scenes.orbit_pose(1, 8), K)
    vbo = oracle.Image(w, h, channels=4)
    d = oracle.Image(w, h)
    d.data[...] = depth
    d.data[::11, ::5] = np.nan
    oracle.depth_to_vbo(vbo, d, K)
    rgb = oracle.Image(cw, ch, np.uint8, 3)
    rgb.data[...] = rng.integers(0, 256, (ch, cw, 3), dtype=np.uint8)
    Kc = scenes.intrinsics(cw, ch)
    T_cd = np.array([[1, 0, 0, 0.025], [0, 1, 0, -0.01], [0, 0, 1, 0.0]], np.float64)
    KT = (np.array([[Kc[0], 0, Kc[2]], [0, Kc[1], Kc[3]], [0, 0, 1]], np.float64) @ T_cd).astype(np.float32)
    return vbo, rgb, KT


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_colour_vbo_vs_reference_headers():
    R = C.CDLL(REF_SO)
    vbo, rgb, KT = depth_tool_inputs()
    got, want = oracle.Image(vbo.w, vbo.h, np.uint8, 4), oracle.Image(vbo.w, vbo.h, np.uint8, 4)
    oracle.colour_vbo(got, vbo, rgb, KT)
    kt = np.ascontiguousarray(KT, np.float32).reshape(-1)
    R.ref_colour_vbo(want.ref(), vbo.ref(), rgb.ref(), kt.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(got.data, want.data) and (got.data[..., 3] == 255).mean() > 0.3 and (got.data[..., 3] == 0).any()


def test_oracle_is_clean_under_asan_and_ubsan():
    """Every oracle entry point on small ragged inputs under AddressSanitizer + UBSan (oracle/sanitize_check.c):
    the checker itself must not read out of bounds (GPU sanitizers are unavailable on the target pool)."""
    import subprocess
    out = subprocess.run(["make", "-C", os.path.join(T.ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    if "unrecognized" in out.stderr or "cannot find -lasan" in out.stderr or "libasan" in out.stderr and out.returncode != 0:
        pytest.skip("this toolchain has no AddressSanitizer runtime")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "sanitize_check:" in out.stdout


def texture_inputs(w=96, h=72, seed=4):
    """A rendered view (depth / normals / phong from the oracle's raycast of a fused room) and three RGB keyframes around it."""
    rng = np.random.default_rng(seed)
    K = scenes.intrinsics(w, h)
    vol = T.make_volume(40, "room")
    T.fuse_frames_oracle(vol, "room", w, h, 2)
    T_wd = scenes.orbit_pose(1, 8)
    bmin, bmax, near, far = scenes.SCENES["room"]
    tr = scenes.trunc_dist(bmin, bmax, (40, 40, 40))
    rd, rn, ri = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
    oracle.raycast_sdf(rd, rn, ri, vol, T_wd, K, near, far, tr, True)
    kfs = []
    for i, (cw, ch) in enumerate(((80, 60), (64, 64), (100, 40))):
        img = oracle.Image(cw, ch, np.uint8, 3)
        img.data[...] = rng.integers(0, 256, (ch, cw, 3), dtype=np.uint8)
        kfs.append((img, scenes.se3_inverse(scenes.orbit_pose(i, 8, yaw_deg=12.0, trans=0.2)), scenes.intrinsics(cw, ch)))
    return K, T_wd, rd, rn, ri, kfs


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_texture_depth_vs_reference_headers():
    R = C.CDLL(REF_SO)
    K, T_wd, rd, rn, ri, kfs = texture_inputs()
    w, h = rd.w, rd.h
    for keyframes, phong in ((kfs[:1], None), (kfs + [(None, np.zeros((3, 4)), np.ones(4))], ri), (kfs[1:2], ri)):
        got, want = oracle.Image(w, h, channels=4), oracle.Image(w, h, channels=4)
        oracle.texture_depth(got, keyframes, rd, rn, T_wd, K, phong)
        oracle.texture_depth(want, keyframes, rd, rn, T_wd, K, phong, fn=R.ref_texture_depth)
        assert T.nan_equal(got.data, want.data), T.mismatch_report(got.data, want.data)
        assert (got.data[..., 3] == 1).all() and (got.data[..., :3] > 0).any()


def _ref_lib():
    R = C.CDLL(REF_SO)
    R.ref_phong_shade.restype = C.c_float
    return R


def _pf(a):
    a = np.ascontiguousarray(a, np.float32).reshape(-1)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_bilateral_vs_reference_headers():
    """BilateralFilter, all four instantiations the path uses (cu_bilateral.cu:52-53,103-104): the restatement against
    the kernel's loop run on the reference's Image::InBounds / GetWithClampedRange -- same libm expf, so bit-equal."""
    R = _ref_lib()
    rng = np.random.default_rng(21)
    w, h = 75, 53
    K = scenes.intrinsics(w, h)
    raw = scenes.render_depth("room", w, h, scenes.orbit_pose(2, 8), K, noise_sigma=0.003, seed=5)
    raw[7:15, 20:31] = np.nan          # holes
    raw[30:33, :] = 0.1                # below minval
    raw[0, 0] = np.nan                 # invalid corner: clamped taps
    for gs, gr, size in ((1.5, 0.1, 3), (2.0, 0.05, 1), (0.8, 0.3, 4)):
        for minval in (0.2, None):
            src = oracle.Image.from_numpy(raw if minval is not None else np.nan_to_num(raw, nan=1.0), pitch_bytes=w * 4 + 32)
            got, want = oracle.Image(w, h), oracle.Image(w, h)
            oracle.bilateral(got, src, gs, gr, size, minval)
            R.ref_bilateral_f32(want.ref(), src.ref(), C.c_float(gs), C.c_float(gr), size, C.c_float(minval or 0.0), int(minval is not None))
            assert T.nan_equal(got.data, want.data), (gs, gr, size, minval, T.mismatch_report(got.data, want.data))
            if minval is not None:
                assert np.isnan(got.data).any() and np.isfinite(got.data).any()
    mm = rng.integers(0, 4000, (h, w)).astype(np.uint16)
    mm[10:14] = 100
    src = oracle.Image.from_numpy(mm)
    got, want = oracle.Image(w, h), oracle.Image(w, h)
    oracle.bilateral(got, src, 1.5, 80.0, 3, 200)
    R.ref_bilateral_u16(want.ref(), src.ref(), C.c_float(1.5), C.c_float(80.0), 3, C.c_ushort(200))
    assert T.nan_equal(got.data, want.data) and np.isnan(got.data).any()
    g8 = rng.integers(0, 256, (h, w)).astype(np.uint8)
    src = oracle.Image.from_numpy(g8)
    oracle.bilateral(got, src, 1.5, 20.0, 2)
    R.ref_bilateral_u8(want.ref(), src.ref(), C.c_float(1.5), C.c_float(20.0), 2)
    assert T.nan_equal(got.data, want.data) and np.isfinite(got.data).all()


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_normals_from_vbo_vs_reference_headers():
    """NormalsFromVbo (cu_normals.cu:12-38) on the reference's float4 operator-, length, make_float4: NaN vertices,
    zero-area triangles (0/0) and the zeroed last row / column included."""
    R = _ref_lib()
    w, h = 67, 45
    K = scenes.intrinsics(w, h)
    raw = scenes.render_depth("room", w, h, scenes.orbit_pose(3, 8), K, noise_sigma=0.002, seed=9)
    raw[5:9, 11:19] = np.nan
    vbo = oracle.Image(w, h, channels=4, pitch_bytes=w * 16 + 48)
    oracle.depth_to_vbo(vbo, oracle.Image.from_numpy(raw), K)
    vbo.data[20, 30] = vbo.data[20, 31] = vbo.data[21, 30]        # degenerate: a = b = 0
    got, want = oracle.Image(w, h, channels=4), oracle.Image(w, h, channels=4)
    oracle.normals_from_vbo(got, vbo)
    R.ref_normals_from_vbo(want.ref(), vbo.ref())
    assert T.nan_equal(got.data, want.data), T.mismatch_report(got.data, want.data)
    assert (got.data[-1] == 0).all() and (got.data[:, -1] == 0).all() and np.isnan(got.data[20, 30, :3]).all()
    inner = got.data[:-1, :-1]
    ok = np.isfinite(inner[..., 0])
    assert ok.sum() > 0.8 * w * h and (inner[..., 3] == 1).all()


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_phong_shade_vs_reference_source():
    """PhongShade compiled from the reference's own lines (cu_raycast.cu:14-28, extracted by oracle/Makefile at build
    time): the oracle's shade image equals it on every hit of a raycast, and misses are 0."""
    R = _ref_lib()
    w, h = 96, 72
    K = scenes.intrinsics(w, h)
    vol = T.make_volume(40, "room")
    T.fuse_frames_oracle(vol, "room", w, h, 2)
    bmin, bmax, near, far = scenes.SCENES["room"]
    tr = scenes.trunc_dist(bmin, bmax, (40, 40, 40))
    for T_wc in (scenes.orbit_pose(1, 8), scenes.orbit_pose(5, 8, yaw_deg=15.0, trans=0.15)):
        rd, rn, ri = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
        st = oracle.raycast_sdf(rd, rn, ri, vol, T_wc, K, near, far, tr, True)
        want = oracle.Image(w, h)
        want.data[...] = 7.0
        _, k = _pf(K)
        R.ref_raycast_shade(want.ref(), rd.ref(), rn.ref(), k)
        assert T.nan_equal(ri.data, want.data), T.mismatch_report(ri.data, want.data)
        assert st["hits"] > 1000 and (ri.data[np.isnan(rd.data)] == 0).all() and (ri.data[np.isfinite(rd.data)] > 0.0).all()
    # and the function itself on arbitrary (not unit, back-facing) vectors
    rng = np.random.default_rng(2)
    for _ in range(200):
        p, n = rng.normal(0, 1, 3).astype(np.float32), rng.normal(0, 1, 3).astype(np.float32)
        _, pp = _pf(p)
        _, nn = _pf(n)
        assert np.float32(R.ref_phong_shade(pp, nn)) == np.float32(oracle.phong_shade(p, n))


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_sdf_sphere_and_preamble_vs_reference_headers():
    """SdfSphere (cu_sdffusion.cu:175-195), ElementwiseScaleBias<float,float,float> (cu_operations.cu:39-57) and
    BoxHalfIgnoreInvalid<float,float,float> (cu_resample.cu:89-120) against the kernels' bodies run on the reference's
    VoxelPositionInUnits / length / SDF_t(float), ConvertPixel and InvalidValue<float>."""
    R = _ref_lib()
    for dims, pitch in (((32, 32, 32), None), ((20, 17, 13), 20 * 8 + 24)):
        a = oracle.Volume(*dims, (-1, -0.5, 0.25), (1, 1.5, 2.0), pitch_bytes=pitch)
        b = oracle.Volume(*dims, (-1, -0.5, 0.25), (1, 1.5, 2.0), pitch_bytes=pitch)
        for v in (a, b):
            oracle.sdf_reset(v, float("nan"))
        c, cp = _pf((0.05, 0.3, 1.1))
        oracle.sdf_sphere(a, c, 0.6)
        R.ref_sdf_sphere(b.ref(), cp, C.c_float(0.6))
        assert T.nan_equal(a.data, b.data) and (a.data[..., 1][np.isfinite(a.data[..., 0])] == 1).all()
        assert np.isnan(a.data[..., 0]).any() == (dims != (32, 32, 32))   # (dim/8)*8 extents
    rng = np.random.default_rng(8)
    w, h = 70, 50
    mm = rng.uniform(0, 6000, (h, w)).astype(np.float32)
    mm[3:6, 9:20] = np.nan
    src = oracle.Image.from_numpy(mm, pitch_bytes=w * 4 + 16)
    got, want = oracle.Image(w, h), oracle.Image(w, h)
    oracle.elementwise_scale_bias(got, src, 1.0 / 1000.0, 0.0)
    R.ref_elementwise_scale_bias_f32(want.ref(), src.ref(), C.c_float(1.0 / 1000.0), C.c_float(0.0))
    assert T.nan_equal(got.data, want.data)
    oracle.elementwise_scale_bias(got, src, 0.5, -1.25)
    R.ref_elementwise_scale_bias_f32(want.ref(), src.ref(), C.c_float(0.5), C.c_float(-1.25))
    assert T.nan_equal(got.data, want.data)
    m = got.data.copy()
    m[10:12, 10:12] = np.nan       # a fully invalid 2x2 cell -> NaN out
    m[20, 21] = np.inf             # non-finite counts as invalid
    src = oracle.Image.from_numpy(m)
    g2, w2 = oracle.Image(w // 2, h // 2), oracle.Image(w // 2, h // 2)
    oracle.box_half_ignore_invalid(g2, src)
    R.ref_box_half_ignore_invalid_f32(w2.ref(), src.ref())
    assert T.nan_equal(g2.data, w2.data) and np.isnan(g2.data[5, 5]) and np.isfinite(g2.data).sum() > 0.9 * g2.data.size
