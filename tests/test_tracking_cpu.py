"""CPU tests of SURVEY 8(f) row f-2 (projective point-plane ICP): the oracle's restatement against the
reference's own headers (oracle/_ref), the host-side solve (kangaroo_amd/tracking.py) against scipy, and the
coarse-to-fine loop recovering a known camera motion with the oracle standing in for the HIP operator."""
import ctypes as C
import os

import numpy as np
import pytest

import kfx_testlib as T
from kfx_testlib import oracle, scenes
from kangaroo_amd import tracking

REF_SO = os.path.join(T.ROOT, "oracle", "_ref", "libkfx_refhdr.so")


def vertex_maps(scene, w, h, T_wc, K, noise=0.0, seed=0):
    raw = scenes.render_depth(scene, w, h, T_wc, K, noise_sigma=noise, seed=seed)
    f, vbo, nrm = T.preprocess_oracle(raw, K)
    return f, vbo, nrm


def icp_inputs(scene, w, h, holes=False):
    """Live maps at pose 1, model maps at pose 0 of a small orbit; optional NaN holes and w != 1 normals."""
    K = scenes.intrinsics(w, h)
    T_wp, T_wl = scenes.orbit_pose(0, 24), scenes.orbit_pose(1, 24)
    _, Pr, Nr = vertex_maps(scene, w, h, T_wp, K)
    _, Pl, _ = vertex_maps(scene, w, h, T_wl, K)
    if holes:
        rng = np.random.default_rng(5)
        Pr.data[rng.random((h, w)) < 0.05] = np.float32("nan")
        Pl.data[rng.random((h, w)) < 0.05] = np.float32("nan")
        Nr.data[rng.random((h, w)) < 0.05, 3] = 0.0
    return K, Pl, Pr, Nr, T_wp, T_wl


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("w,h,holes", [(160, 120, False), (80, 60, True), (20, 15, True), (96, 72, True)])
def test_oracle_icp_matches_reference_headers(w, h, holes):
    """Bit-equality of per-block systems, final system and debug image with the loop that calls the
    reference's header arithmetic (Mat / MatUtils / Image / reweighting) step by step."""
    R = C.CDLL(REF_SO)
    PF = C.POINTER(C.c_float)
    R.ref_icp_point_plane.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, PF, PF, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    R.ref_icp_point_plane.restype = None
    K, Pl, Pr, Nr, T_wp, T_wl = icp_inputs("room", w, h, holes)
    # a guess slightly off identity so KT_lr / T_rl are general matrices
    T_lp = tracking.se3_exp([0.004, -0.003, 0.002, 0.003, -0.002, 0.001])
    KT = (tracking.k_matrix(K) @ T_lp[:3]).astype(np.float32)
    T_pl = tracking.se3_inv(T_lp)[:3].astype(np.float32)
    d1, d2 = oracle.Image(w, h, channels=4), oracle.Image(w, h, channels=4)
    got, gb = oracle.icp_point_plane(Pl, Pr, Nr, KT, T_pl, 0.1, d1, want_blocks=True)
    want, wb = oracle.icp_point_plane(Pl, Pr, Nr, KT, T_pl, 0.1, d2, want_blocks=True, fn=R.ref_icp_point_plane)
    assert int(got["obs"]) > w * h // 4
    assert gb.tobytes() == wb.tobytes()
    assert got.tobytes() == want.tobytes()
    assert T.nan_equal(d1.data, d2.data)
    # block geometry of InitDimFromOutputImage(dPl, 16, 16)
    bx, by, gx, gy = oracle.icp_block_dims(w, h)
    assert (bx, by) == (np.gcd(w, 16), np.gcd(h, 16)) and gx * bx == w and gy * by == h


def test_oracle_icp_sum_is_a_valid_reduction():
    """The fixed summation order is one of the orders thrust::reduce may take: the float32 tree result
    agrees with a float64 sum of the per-block systems to float32 summation accuracy."""
    w, h = 160, 120
    K, Pl, Pr, Nr, _, _ = icp_inputs("room", w, h, True)
    KT = (tracking.k_matrix(K) @ np.eye(4)[:3]).astype(np.float32)
    got, blocks = oracle.icp_point_plane(Pl, Pr, Nr, KT, np.eye(4, dtype=np.float32)[:3], 0.1, want_blocks=True)
    assert int(got["obs"]) == int(blocks["obs"].sum())
    for name in ("JTy", "JTJ"):
        ref = blocks[name].astype(np.float64).sum(axis=0)
        mag = np.abs(blocks[name].astype(np.float64)).sum(axis=0)
        assert np.all(np.abs(got[name] - ref) <= 1e-6 * mag + 1e-12)
    assert abs(float(got["sqErr"]) - blocks["sqErr"].astype(np.float64).sum()) <= 1e-6 * blocks["sqErr"].sum()


def test_full_piv_lu_and_exponentials():
    from scipy.linalg import expm
    rng = np.random.default_rng(3)
    for n in (3, 6):
        A = rng.standard_normal((n, n))
        A = A @ A.T + 0.5 * np.eye(n)
        b = rng.standard_normal(n)
        assert np.allclose(tracking.full_piv_lu_solve(A, b), np.linalg.solve(A, b), rtol=1e-12, atol=1e-12)
    # rank-deficient: consistent system, free variable set to zero
    A = np.array([[2.0, 0.0, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, 4.0]])
    assert np.allclose(tracking.full_piv_lu_solve(A, [2.0, 0.0, 2.0]), [1.0, 0.0, 0.5])
    assert np.allclose(tracking.full_piv_lu_solve(np.zeros((3, 3)), [1.0, 2.0, 3.0]), 0.0)
    for x in (np.zeros(6), np.array([0.1, -0.2, 0.3, 1e-12, 0.0, 0.0]), rng.standard_normal(6) * 0.3, rng.standard_normal(6) * 2.0):
        G = np.zeros((4, 4))
        G[:3, :3] = tracking.hat(x[3:])
        G[:3, 3] = x[:3]
        assert np.allclose(tracking.se3_exp(x), expm(G), rtol=1e-12, atol=1e-12)
        Tm = tracking.se3_exp(x)
        assert np.allclose(tracking.se3_inv(Tm) @ Tm, np.eye(4), atol=1e-12)
        assert np.allclose(tracking.so3_exp(x[3:]), expm(tracking.hat(x[3:])), rtol=1e-12, atol=1e-12)


def pyramid_maps(scene, w, h, T_wc, K, levels=4):
    """kin_d pyramid (BoxReduceIgnoreInvalid) + per-level DepthToVbo / NormalsFromVbo, main.cpp:209-215."""
    raw = scenes.render_depth(scene, w, h, T_wc, K)
    d = oracle.Image(w, h)
    oracle.bilateral(d, _img(raw), scenes.BILATERAL["gs"], scenes.BILATERAL["gr"], scenes.BILATERAL["size"], scenes.BILATERAL["minval"])
    ds, vs, ns, Ks = [d], [], [], []
    for l in range(levels):
        if l > 0:
            nxt = oracle.Image(w >> l, h >> l)
            oracle.box_half_ignore_invalid(nxt, ds[-1])
            ds.append(nxt)
        Kl = oracle.intrinsics_level(K, l)
        v, n = oracle.Image(w >> l, h >> l, channels=4), oracle.Image(w >> l, h >> l, channels=4)
        oracle.depth_to_vbo(v, ds[l], Kl)
        oracle.normals_from_vbo(n, v)
        vs.append(v)
        ns.append(n)
        Ks.append(Kl)
    return ds, vs, ns, Ks


def _img(arr):
    im = oracle.Image(arr.shape[1], arr.shape[0])
    im.data[...] = arr
    return im


def test_refine_pose_recovers_known_motion():
    """Model maps rendered at pose p, live maps at pose l (analytic scene, no noise): the coarse-to-fine
    loop of main.cpp:301-336 must bring T_lp close to T_wl^-1 T_wp, and the rmse must drop."""
    import oracle_ops as ops
    w, h = 160, 120
    K = scenes.intrinsics(w, h)
    T_wp = scenes.orbit_pose(0, 60)
    T_wl = scenes.orbit_pose(1, 60)
    _, ray_v, ray_n, Ks = pyramid_maps("room", w, h, T_wp, K)
    _, kin_v, _, _ = pyramid_maps("room", w, h, T_wl, K)
    truth = tracking.se3_inv(np.vstack([T_wl, [0, 0, 0, 1]])) @ np.vstack([T_wp, [0, 0, 0, 1]])
    log = []
    T_lp, rmse, good = tracking.refine_pose(ops, kin_v, ray_v, ray_n, Ks, None, its=(4, 3, 3, 3),
                                            on_iteration=lambda l, lss, Tm, r: log.append((l, r, lss.obs)))
    assert good and len(log) == 13 and all(o > 0 for _, _, o in log)
    err0 = np.linalg.norm(truth[:3, 3])                       # error of the identity guess
    err = np.linalg.norm((tracking.se3_inv(truth) @ T_lp)[:3, 3])
    rot = np.arccos(np.clip((np.trace((truth[:3, :3].T @ T_lp[:3, :3])) - 1) / 2, -1, 1))
    assert err < 0.15 * err0 and rot < np.deg2rad(0.05), (err0, err, rot, log)
    # the reference's own schedule (its = {1,0,2,3}: 3 rotation-only + 2 + 0 + 1 iterations)
    T_lp2, _, good2 = tracking.refine_pose(ops, kin_v, ray_v, ray_n, Ks, None)
    assert good2 and np.linalg.norm((tracking.se3_inv(truth) @ T_lp2)[:3, 3]) < 0.2 * err0


def test_tracking_pipeline_follows_the_orbit():
    """TrackingPipeline (raycast of the model -> ICP -> fuse at the refined pose) with the oracle operators:
    the estimated trajectory stays within a few millimetres of the true orbit, without being told the poses."""
    import oracle_ops as ops
    from kangaroo_amd.pipeline import TrackingPipeline
    N, w, h, frames = 64, 160, 120, 5
    bmin, bmax, near, far = scenes.SCENES["room"]
    pipe = TrackingPipeline(ops, (N, N, N), bmin, bmax, w, h, near=near, far=far)
    worst = 0.0
    for i in range(frames):
        T_true = scenes.orbit_pose(i, 60)
        pipe.raw.MemcpyFromHost(scenes.render_depth("room", w, h, T_true, pipe.K))
        T_est = pipe.step(T_wl_init=T_true if i == 0 else None)
        assert pipe.tracking_good
        worst = max(worst, float(np.linalg.norm(T_est[:3, 3] - T_true[:3, 3])))
    drift_if_static = float(np.linalg.norm(scenes.orbit_pose(frames - 1, 60)[:3, 3] - scenes.orbit_pose(0, 60)[:3, 3]))
    assert worst < 0.35 * drift_if_static, (worst, drift_if_static)


def test_tracking_pipeline_recovers_after_a_frame_without_depth():
    """main.cpp:223-242: a frame with no valid depth leaves the refinement without a single correspondence (rmse = sqrt(0 / 0)):
    the frame is not fused, the pose stays; on the NEXT frame the application starts over -- T_wl = identity, SdfReset(vol, NaN),
    the current frame fused -- and tracks on from there, in a world frame that sits at that frame's camera."""
    import oracle_ops as ops
    from kangaroo_amd.pipeline import TrackingPipeline
    N, w, h = 64, 160, 120
    bmin, bmax, near, far = scenes.SCENES["room"]
    pipe = TrackingPipeline(ops, (N, N, N), bmin, bmax, w, h, near=near, far=far)
    drop, anchor = 3, None
    for i in range(8):
        T_true = scenes.orbit_pose(i, 60)
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
            anchor = np.vstack([T_true, [0, 0, 0, 1]])   # the new world frame = this frame's camera
        assert pipe.tracking_good and np.isfinite(pipe.rmse), i
        T_abs = T_est if anchor is None else anchor @ T_est
        assert np.linalg.norm(T_abs[:3, 3] - T_true[:3, 3]) < 1e-2, (i, T_abs[:3, 3], T_true[:3, 3])
    assert pipe.resets == 1
    # the model was rebuilt from the frames after the drop-out only: cells the first three frames alone had observed are unknown again
    assert np.isnan(pipe.vol.data[..., 0]).any() and np.isfinite(pipe.vol.data[..., 0]).sum() > 0.2 * N ** 3


def test_pose_step_host_function_matches_the_matrix_expression():
    """kfx_pose_step (host code of libkfx: the tracked loop's T_wl = T_wl * T_lp^-1 and the float T_cw SdfFuse takes) against the
    4 x 4 matrix expression of tracking.py, for random rigid transforms: float64 to rounding, the float inverse exactly the
    rounded double inverse.  Runs without a GPU."""
    from kangaroo_amd import roo, tracking as tr
    rng = np.random.default_rng(5)
    for _ in range(50):
        T_wl, T_lp = np.eye(4), np.eye(4)
        for T in (T_wl, T_lp):
            q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            T[:3, :3] = q * np.sign(np.linalg.det(q))
            T[:3, 3] = rng.normal(size=3)
        got, inv = roo.PoseStep(T_wl, T_lp)
        want = T_wl @ tr.se3_inv(T_lp)
        assert np.abs(got - want).max() < 1e-14 and got[3].tolist() == [0.0, 0.0, 0.0, 1.0]
        assert inv.dtype == np.float32 and np.abs(inv - tr.se3_inv(got)[:3]).max() < 1e-6
        again, _ = roo.PoseStep(got, np.eye(4))
        assert again.tobytes() == got.tobytes()   # a step by the identity changes nothing
