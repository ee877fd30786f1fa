"""The brick summary (kfx_sdf_summary): tracked SdfFuse keeps per-8^3-brick value ranges as a by-product, tracked RaycastSdf
steps through uniform bricks without reading the volume.
  * the volume bits do not depend on tracking;
  * the summary is conservative against the actual volume contents, brick by brick;
  * exact numerics: tracked raycast images are bit-identical to the untracked ones (and so to the oracle);
  * fast numerics: observed free space is skipped too -- depth / normals within the fast-mode tolerance of the exact march;
  * views, ragged dimensions, untracked writers + invalidate()."""
import ctypes as C

import numpy as np
import pytest

import kfx_testlib as T
from kfx_testlib import oracle, scenes

pytestmark = pytest.mark.gpu


def export(roo, summ, tol):
    import torch
    from kangaroo_amd import _lib
    L = _lib.load()
    _lib.load_debug().kfx_debug_summary_export.restype = C.c_int
    _lib.load_debug().kfx_debug_summary_export.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
    dims = (C.c_int * 9)()
    assert _lib.load_debug().kfx_debug_summary_export(summ.handle, tol, None, None, dims, None) == 0
    nbx, nby, nbz = dims[0], dims[1], dims[2]
    n1, n2, n3 = nbx * nby * nbz, dims[3] * dims[4] * dims[5], dims[6] * dims[7] * dims[8]
    R = torch.empty((nbz, nby, nbx, 4), dtype=torch.float32, device="cuda")
    Dall = torch.empty(n1 + n2 + n3, dtype=torch.float32, device="cuda")
    assert _lib.load_debug().kfx_debug_summary_export(summ.handle, tol, C.c_void_p(R.data_ptr()), C.c_void_p(Dall.data_ptr()), dims, None) == 0
    torch.cuda.synchronize()
    D1 = Dall[:n1].view(nbz, nby, nbx)
    D2 = Dall[n1:n1 + n2].view(dims[5], dims[4], dims[3])
    # the coarse levels may only promise what every level-1 entry below them promises
    up = D2.repeat_interleave(4, 0).repeat_interleave(4, 1).repeat_interleave(4, 2)[:nbz, :nby, :nbx]
    uni = up > 0
    assert bool((D1[uni] > 0).all()) and bool(((D1[uni] - up[uni]).abs() <= tol * up[uni] + 0.0).all())
    assert bool(torch.isnan(D1[torch.isnan(up)]).all())
    return R, D1


def check_conservative(vol, R):
    """Every brick's summary must cover what the volume really holds."""
    import torch
    v = vol.tensor()[..., 0]
    d, h, w = v.shape
    nbz, nby, nbx = R.shape[:3]
    pad = torch.full((nbz * 8, nby * 8, nbx * 8), float("nan"), device=v.device)
    known = torch.zeros_like(pad, dtype=torch.bool)
    pad[:d, :h, :w] = v
    known[:d, :h, :w] = True
    br = pad.view(nbz, 8, nby, 8, nbx, 8).permute(0, 2, 4, 1, 3, 5).reshape(nbz, nby, nbx, 512)
    kn = known.view(nbz, 8, nby, 8, nbx, 8).permute(0, 2, 4, 1, 3, 5).reshape(nbz, nby, nbx, 512)
    isn = torch.isnan(br) & kn
    has = ~torch.isnan(br) & kn
    tmin = torch.where(has, br, torch.full_like(br, float("inf"))).amin(-1)
    tmax = torch.where(has, br, torch.full_like(br, float("-inf"))).amax(-1)
    state = R[..., 2].contiguous().view(torch.int32)
    all_nan, any_nan, any_val = ~has.any(-1), isn.any(-1), has.any(-1)
    s0, s1 = state == 0, state == 1
    assert not bool((s1 & any_val).any()), "brick marked all-NaN holds values"
    assert not bool((s0 & any_nan).any()), "brick marked all-values holds NaN"
    assert bool((R[..., 0][s0] <= tmin[s0]).all()) and bool((R[..., 1][s0] >= tmax[s0]).all()), "range does not cover the brick"
    # mixed / unknown bricks: whatever cells hold a value, the range covers them (the march's class 3 -- "NaN or +trunc" --
    # relies on it; an invalidated brick has the infinite range)
    s2 = (state == 2) & any_val
    assert bool((R[..., 0][s2] <= tmin[s2]).all()) and bool((R[..., 1][s2] >= tmax[s2]).all()), "range of a mixed brick does not cover its values"
    return dict(uniform_ranges=int(s0.sum()), all_nan=int(s1.sum()), mixed=int((state == 2).sum()), true_all_nan=int(all_nan.sum()))


@pytest.mark.parametrize("scene,N,w,h,dims", [("room", 128, 320, 240, None), ("full", 96, 160, 120, None), ("room", 0, 200, 150, (100, 84, 92))])
@pytest.mark.parametrize("math", ["exact", "fast"])
def test_gpu_tracked_fuse_and_raycast(roo, scene, N, w, h, dims, math):
    dims = dims or (N, N, N)
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(bmin, bmax, dims)
    prev = roo.set_math_mode(math)
    try:
        va, vb = roo.BoundedVolume(*dims, bmin, bmax), roo.BoundedVolume(*dims, bmin, bmax)
        summ = roo.SdfSummary(vb)
        roo.SdfReset(va, float("nan"))
        roo.SdfReset(vb, float("nan"), summary=summ)
        f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
        for i in range(4):
            T_wc = scenes.orbit_pose(i, 30)
            roo.BilateralFilter(f, T.upload_image(roo, scenes.render_depth(scene, w, h, T_wc, K)), **scenes.BILATERAL)
            roo.DepthToVbo(vbo, f, K)
            roo.NormalsFromVbo(nrm, vbo)
            T_cw = scenes.se3_inverse(T_wc)
            roo.SdfFuse(va, f, nrm, T_cw, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
            roo.SdfFuse(vb, f, nrm, T_cw, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, summary=summ)
            # (1) tracking never changes the volume
            assert T.nan_equal(va.MemcpyToHost(), vb.MemcpyToHost())
            # (2) the summary covers the volume's real contents
            R, D = export(roo, summ, 1e-5 if math == "fast" else 0.0)
            stats = check_conservative(vb, R)
            # (3) raycast with and without the summary
            a = [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
            b = [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
            roo.set_math_mode("exact")
            roo.RaycastSdf(*a, va, T_wc, K, near, far, tr, True)                      # the reference march
            roo.set_math_mode(math)
            roo.RaycastSdf(*b, vb, T_wc, K, near, far, tr, True, summary=summ)
            da, db = a[0].MemcpyToHost(), b[0].MemcpyToHost()
            na, nb = a[1].MemcpyToHost(), b[1].MemcpyToHost()
            n_uniform = int((D > 0).sum())
            if math == "exact":
                assert T.nan_equal(da, db) and T.nan_equal(na, nb) and T.nan_equal(a[2].MemcpyToHost(), b[2].MemcpyToHost())
                if i == 0 and dims[2] % 8 == 0:
                    assert n_uniform > 0      # first observation: +trunc everywhere in front of the surfaces, bit-identical
            else:
                hit_a, hit_b = np.isfinite(da), np.isfinite(db)
                assert (hit_a != hit_b).sum() <= max(3, 2e-4 * w * h), (hit_a != hit_b).sum()
                both = hit_a & hit_b
                assert both.sum() > 0.03 * w * h
                assert np.abs(da[both] - db[both]).max() < 1e-4, np.abs(da[both] - db[both]).max()
                cosang = np.clip(np.sum(na[both][:, :3].astype(np.float64) * nb[both][:, :3], axis=1), -1, 1)
                assert np.arccos(cosang).max() < 2e-3
                assert n_uniform > (0.05 if dims[0] % 8 == 0 else 0.02) * D.numel(), (n_uniform, D.numel(), stats)   # observed free space is recognised frame after frame
        assert stats["all_nan"] > 0 or scene == "full"
    finally:
        roo.set_math_mode(prev)


@pytest.mark.parametrize("scene", ["room", "full"])
def test_gpu_tracked_raycast_of_pyramid_levels_in_one_launch(roo, scene):
    """kfx_raycast_sdf_levels_tracked: the tracking loop's three renderings (640x480-like pyramid levels 0, 2, 3) from one
    launch with the summary consulted.  Every image -- and the vertex map of each level -- must equal the per-level tracked
    call, which in exact numerics equals the plain march bit for bit."""
    N, w, h = 128, 320, 240
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    prev = roo.set_math_mode("exact")
    try:
        vol = roo.BoundedVolume(N, N, N, bmin, bmax)
        summ = roo.SdfSummary(vol)
        roo.SdfReset(vol, float("nan"), summary=summ)
        f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
        for i in range(2):
            T_wc = scenes.orbit_pose(i, 30)
            roo.BilateralFilter(f, T.upload_image(roo, scenes.render_depth(scene, w, h, T_wc, K)), **scenes.BILATERAL)
            roo.DepthToVbo(vbo, f, K)
            roo.NormalsFromVbo(nrm, vbo)
            roo.SdfFuse(vol, f, nrm, scenes.se3_inverse(T_wc), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, summary=summ)
        levels = [0, 2, 3]
        Ks = [scenes.intrinsics_level(K, l) for l in levels]
        def images(l):
            return [roo.Image(w >> l, h >> l), roo.Image(w >> l, h >> l, "f32x4"), roo.Image(w >> l, h >> l), roo.Image(w >> l, h >> l, "f32x4")]
        one = [images(l) for l in levels]
        roo.RaycastSdfLevels([tuple(o) for o in one], vol, T_wc, Ks, near, far, tr, True, summary=summ)
        for math in ("exact", "fast"):          # the summary's tolerance follows the numerics mode: compare like with like
            roo.set_math_mode(math)
            roo.RaycastSdfLevels([tuple(o) for o in one], vol, T_wc, Ks, near, far, tr, True, summary=summ)
            for o, l, Kl in zip(one, levels, Ks):
                ref = images(l)
                roo.RaycastSdf(ref[0], ref[1], ref[2], vol, T_wc, Kl, near, far, tr, True, summary=summ)
                roo.DepthToVbo(ref[3], ref[0], Kl)
                for a, b in zip(o, ref):
                    assert T.nan_equal(a.MemcpyToHost(), b.MemcpyToHost()), (math, l)
                if math == "exact":
                    plain = images(l)
                    roo.RaycastSdf(plain[0], plain[1], plain[2], vol, T_wc, Kl, near, far, tr, True)
                    assert T.nan_equal(o[0].MemcpyToHost(), plain[0].MemcpyToHost()) and T.nan_equal(o[1].MemcpyToHost(), plain[1].MemcpyToHost())
                assert np.isfinite(o[0].MemcpyToHost()).sum() > 0.02 * (w >> l) * (h >> l)
    finally:
        roo.set_math_mode(prev)


def test_gpu_summary_views_and_untracked_writers(roo):
    """8-aligned views keep tracking, unaligned views and untracked writers drop to 'unknown' (correct, nothing skipped);
    SdfSphere + invalidate; all raycasts equal the untracked kernel bit for bit (exact numerics)."""
    N, w, h = 96, 160, 120
    bmin, bmax, near, far = scenes.SCENES["room"]
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    vol = roo.BoundedVolume(N, N, N, bmin, bmax)
    summ = roo.SdfSummary(vol)
    roo.SdfReset(vol, float("nan"), summary=summ)
    f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
    T_wc = scenes.orbit_pose(1, 30)
    roo.BilateralFilter(f, T.upload_image(roo, scenes.render_depth("room", w, h, T_wc, K)), **scenes.BILATERAL)
    roo.DepthToVbo(vbo, f, K)
    roo.NormalsFromVbo(nrm, vbo)

    def same_images(v):
        a = [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
        b = [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
        roo.RaycastSdf(*a, v, T_wc, K, near, far, tr, True)
        roo.RaycastSdf(*b, v, T_wc, K, near, far, tr, True, summary=summ)
        return all(T.nan_equal(x.MemcpyToHost(), y.MemcpyToHost()) for x, y in zip(a, b))

    assert same_images(vol)                                         # all NaN: every ray misses on both paths
    aligned = vol.SubVolume((16, 8, 24), (64, 80, 56))
    roo.SdfFuse(aligned, f, nrm, scenes.se3_inverse(T_wc), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, summary=summ)
    R, D = export(roo, summ, 0.0)
    st = check_conservative(vol, R)
    assert st["uniform_ranges"] > 0 and st["all_nan"] > 0 and st["mixed"] > 0   # mixed: partially observed bricks
    assert same_images(vol) and same_images(aligned)
    ragged = vol.SubVolume((3, 8, 24), (64, 80, 56))
    roo.SdfFuse(ragged, f, nrm, scenes.se3_inverse(T_wc), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, summary=summ)
    R, D = export(roo, summ, 0.0)
    assert check_conservative(vol, R)["mixed"] == R.shape[0] * R.shape[1] * R.shape[2] and int((D > 0).sum()) == 0
    assert same_images(vol) and same_images(ragged)
    roo.SdfReset(vol, float("nan"), summary=summ)
    roo.SdfSphere(vol, (0.0, 0.0, 3.0), 0.5)
    summ.invalidate()
    assert same_images(vol)
