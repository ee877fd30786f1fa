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


def export(roo, summ, tol, vref, fine_shift=4):
    """R (per-brick ranges) and the march's class tables, decoded to one int8 class per entry: {shift: tensor[nz, ny, nx]}."""
    import torch
    from kangaroo_amd import _lib
    f = _lib.load_debug().kfx_debug_summary_export
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
    dims = (C.c_int * 12)()
    assert f(summ.handle, tol, vref, fine_shift, None, None, dims, None) == 0
    nbx, nby, nbz = dims[0], dims[1], dims[2]
    R = torch.empty((nbz, nby, nbx, 4), dtype=torch.float32, device="cuda")
    Cw = torch.zeros(dims[9], dtype=torch.int32, device="cuda")
    assert f(summ.handle, tol, vref, fine_shift, C.c_void_p(R.data_ptr()), C.c_void_p(Cw.data_ptr()), dims, None) == 0
    torch.cuda.synchronize()
    W, H, D = summ.vol.w, summ.vol.h, summ.vol.d
    classes = {}
    for shift, (first, rw, ny) in ((fine_shift, dims[3:6]), (5, dims[6:9])):
        m = 1 << shift
        nx, nz = -(-W // m), -(-D // m)
        assert ny == -(-H // m) and rw == 2 * (-(-nx // 32))
        words = Cw[first:first + rw * ny * nz].view(nz, ny, rw // 2, 2).to(torch.int64) & 0xffffffff
        bits = (words.unsqueeze(-1) >> torch.arange(32, device="cuda")) & 1           # [nz, ny, rw/2, plane, 32]
        p0 = bits[..., 0, :].reshape(nz, ny, -1)[..., :nx]
        p1 = bits[..., 1, :].reshape(nz, ny, -1)[..., :nx]
        classes[shift] = (p0 | (p1 << 1)).to(torch.int8)
    published, n_coarse = dims[11], dims[10]
    assert published == int((classes[5] != 0).sum()) and n_coarse == classes[5].numel()   # the count the host steers by
    if fine_shift < 5:
        # the 32^3-cell level is the combination of the fine level's entries under it (clamped at the far edges)
        fine, m = classes[fine_shift].to(torch.int64), 32 >> fine_shift
        nzc, nyc, nxc = classes[5].shape
        iz = (torch.arange(nzc * m, device="cuda")).clamp(max=fine.shape[0] - 1)
        iy = (torch.arange(nyc * m, device="cuda")).clamp(max=fine.shape[1] - 1)
        ix = (torch.arange(nxc * m, device="cuda")).clamp(max=fine.shape[2] - 1)
        sub = fine[iz][:, iy][:, :, ix].view(nzc, m, nyc, m, nxc, m).permute(0, 2, 4, 1, 3, 5).reshape(nzc, nyc, nxc, -1)
        all_free, all_nan, all_either = (sub == 1).all(-1), (sub == 2).all(-1), (sub != 0).all(-1)
        want = torch.where(all_free, 1, torch.where(all_nan, 2, torch.where(all_either, 3, 0))).to(torch.int8)
        assert bool((want == classes[5]).all()), "32^3-cell classes differ from the combination of the fine entries: %d" % int((want != classes[5]).sum())
    return R, classes


def check_classes(vol, classes, tol, vref):
    """What the march relies on, against the volume's real contents: an entry of class 1 / 2 / 3 holds -- in its own cells and
    in the +1 cells a trilinear sample based in it reads -- only vref (within tol) / only NaN / only NaN or vref."""
    import torch
    import torch.nn.functional as F
    v = vol.tensor()[..., 0]
    d, h, w = v.shape
    flags = {1: (v - vref).abs() <= tol * vref, 2: torch.isnan(v)}
    flags[3] = flags[1] | flags[2]
    out = {}
    for shift, cls in classes.items():
        m = 1 << shift
        nz, ny, nx = cls.shape
        for k, fl in flags.items():
            bad = (~fl).float()[None, None]
            bad = F.pad(bad, (0, nx * m + 1 - w, 0, ny * m + 1 - h, 0, nz * m + 1 - d), value=0.0)   # cells outside the volume do not exist
            holds = F.max_pool3d(bad, kernel_size=m + 1, stride=m)[0, 0] == 0                  # every cell of [b m, b m + m] satisfies the flag
            wrong = (cls == k) & ~holds
            assert not bool(wrong.any()), "class %d entries of the 2^%d level that the volume contradicts: %d" % (k, shift, int(wrong.sum()))
        out[shift] = {k: int((cls == k).sum()) for k in range(4)}
    return out


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
            tol = 1e-5 if math == "fast" else 0.0
            R, classes = export(roo, summ, tol, tr, fine_shift=4)
            stats = check_conservative(vb, R)
            check_classes(vb, classes, tol, np.float32(tr))
            R, classes = export(roo, summ, tol, tr, fine_shift=3)
            counts = check_classes(vb, classes, tol, np.float32(tr))
            # (3) raycast with and without the summary
            a = [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
            b = [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
            roo.set_math_mode("exact")
            roo.RaycastSdf(*a, va, T_wc, K, near, far, tr, True)                      # the reference march
            roo.set_math_mode(math)
            roo.RaycastSdf(*b, vb, T_wc, K, near, far, tr, True, summary=summ)
            da, db = a[0].MemcpyToHost(), b[0].MemcpyToHost()
            na, nb = a[1].MemcpyToHost(), b[1].MemcpyToHost()
            n_free, n_entries = counts[3][1] + counts[3][3], classes[3].numel()   # 8^3-cell entries holding +trunc (or +trunc / NaN)
            if math == "exact":
                assert T.nan_equal(da, db) and T.nan_equal(na, nb) and T.nan_equal(a[2].MemcpyToHost(), b[2].MemcpyToHost())
                if i == 0 and dims[2] % 8 == 0:
                    assert n_free > 0      # first observation: +trunc everywhere in front of the surfaces, bit-identical
            else:
                hit_a, hit_b = np.isfinite(da), np.isfinite(db)
                assert (hit_a != hit_b).sum() <= max(3, 2e-4 * w * h), (hit_a != hit_b).sum()
                both = hit_a & hit_b
                assert both.sum() > 0.03 * w * h
                assert np.abs(da[both] - db[both]).max() < 1e-4, np.abs(da[both] - db[both]).max()
                cosang = np.clip(np.sum(na[both][:, :3].astype(np.float64) * nb[both][:, :3], axis=1), -1, 1)
                assert np.arccos(cosang).max() < 2e-3
                assert n_free > (0.05 if dims[0] % 8 == 0 else 0.02) * n_entries, (n_free, n_entries, stats)   # observed free space is recognised frame after frame
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
    R, classes = export(roo, summ, 0.0, tr)
    st = check_conservative(vol, R)
    check_classes(vol, classes, 0.0, np.float32(tr))
    assert st["uniform_ranges"] > 0 and st["all_nan"] > 0 and st["mixed"] > 0   # mixed: partially observed bricks
    assert same_images(vol) and same_images(aligned)
    ragged = vol.SubVolume((3, 8, 24), (64, 80, 56))
    roo.SdfFuse(ragged, f, nrm, scenes.se3_inverse(T_wc), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, summary=summ)
    R, classes = export(roo, summ, 0.0, tr)
    assert check_conservative(vol, R)["mixed"] == R.shape[0] * R.shape[1] * R.shape[2] and all(int((c != 0).sum()) == 0 for c in classes.values())
    assert same_images(vol) and same_images(ragged)
    roo.SdfReset(vol, float("nan"), summary=summ)
    roo.SdfSphere(vol, (0.0, 0.0, 3.0), 0.5)
    summ.invalidate()
    assert same_images(vol)


def test_gpu_unforced_choice_between_plain_and_table_march(tmp_path):
    """Without KFX_RAYCAST_SUMMARY the tracked RaycastSdf uses the tables only where at least a quarter of the 32^3-cell
    entries can be crossed without sampling (the count the last finished table build published).  Either way the images are
    those of the plain march (exact numerics: bit for bit) -- checked in a fresh process, the knob is read once."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import kfx_testlib as T
from kfx_testlib import scenes
from kangaroo_amd import roo
N, w, h = 96, 160, 120
for scene in ("room", "full"):
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h); tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    vol = roo.BoundedVolume(N, N, N, bmin, bmax); summ = roo.SdfSummary(vol)
    roo.SdfReset(vol, float("nan"), summary=summ)
    f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
    for i in range(5):
        T_wc = scenes.orbit_pose(i, 30)
        roo.BilateralFilter(f, T.upload_image(roo, scenes.render_depth(scene, w, h, T_wc, K)), **scenes.BILATERAL)
        roo.DepthToVbo(vbo, f, K); roo.NormalsFromVbo(nrm, vbo)
        roo.SdfFuse(vol, f, nrm, scenes.se3_inverse(T_wc), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, summary=summ)
        a = [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
        b = [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
        roo.RaycastSdf(*a, vol, T_wc, K, near, far, tr, True)
        for _ in range(2):      # the second call sees the count the first call's table build published
            roo.RaycastSdf(*b, vol, T_wc, K, near, far, tr, True, summary=summ)
            torch.cuda.synchronize()
            assert all(T.nan_equal(x.MemcpyToHost(), y.MemcpyToHost()) for x, y in zip(a, b)), (scene, i)
print("ok")
''' % (T.ROOT, __import__("os").path.join(T.ROOT, "tests"))
    env = dict(__import__("os").environ)
    env.pop("KFX_RAYCAST_SUMMARY", None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_gpu_tracked_march_counts_its_own_work(roo):
    """kfx_raycast_sdf_count_tracked (bench.py's roofline_raycast): the class-table march with every cell it reads marked in a
    bitmap.  Same rays, same hits as the plain march's count; fewer samples and cells (it crosses free space without reading the
    volume), every cell it reads is one the plain march reads too (same positions), table look-ups and table bytes reported."""
    N, w, h = 96, 200, 150
    bmin, bmax, near, far = scenes.SCENES["full"]
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    vol = roo.BoundedVolume(N, N, N, bmin, bmax)
    summ = roo.SdfSummary(vol)
    roo.SdfReset(vol, float("nan"), summary=summ)
    f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
    # one frame, exact numerics: a cell observed once in front of the wall holds the clamped +trunc bit for bit, so free space qualifies
    T_wc = scenes.orbit_pose(0, 30)
    roo.BilateralFilter(f, T.upload_image(roo, scenes.render_depth("full", w, h, T_wc, K)), **scenes.BILATERAL)
    roo.DepthToVbo(vbo, f, K)
    roo.NormalsFromVbo(nrm, vbo)
    roo.SdfFuse(vol, f, nrm, scenes.se3_inverse(T_wc), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, summary=summ)
    plain = roo.RaycastSdfCount(vol, w, h, T_wc, K, near, far, tr, True)
    tab = roo.RaycastSdfCount(vol, w, h, T_wc, K, near, far, tr, True, summary=summ)
    assert tab["rays"] == plain["rays"] > 0 and tab["hits"] == plain["hits"] > 0
    assert tab["table_bytes"] > 0 and tab["lookups"] > 0, tab          # (the suite forces the table march: tests/conftest.py)
    assert 0 < tab["samples"] < 0.5 * plain["samples"], (tab, plain)
    assert 0 < tab["U"] < plain["U"], (tab, plain)
    # the images of the two marches are the same bits (exact numerics), so the hits' gradient stencils are the same cells
    a = [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
    b = [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
    roo.RaycastSdf(*a, vol, T_wc, K, near, far, tr, True)
    roo.RaycastSdf(*b, vol, T_wc, K, near, far, tr, True, summary=summ)
    assert all(T.nan_equal(x.MemcpyToHost(), y.MemcpyToHost()) for x, y in zip(a, b))


def test_gpu_coarser_levels_derived_in_lds_skip_more_and_change_nothing(tmp_path):
    """The 64^3- and 128^3-cell levels every raycast workgroup derives in LDS from the 32^3-cell level (classes_stage, raycast.hip):
    exact numerics, 256^3 (four and two entries per axis) and a ragged 200 x 168 x 232 volume, both scenes, several frames --
    the images are the plain march's bit for bit with the levels on (default) and off (KFX_RAYCAST_TOP_LEVELS=0, read once per
    process), and with them the march needs fewer table look-ups for the same samples."""
    import subprocess
    import sys
    code = r'''
import sys, json, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import kfx_testlib as T
from kfx_testlib import scenes
from kangaroo_amd import roo
out = {}
for scene, dims, (w, h) in (("full", (256, 256, 256), (320, 240)), ("room", (256, 256, 256), (320, 240)), ("room", (200, 168, 232), (200, 150))):
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h); tr = scenes.trunc_dist(bmin, bmax, dims)
    vol = roo.BoundedVolume(*dims, bmin, bmax); summ = roo.SdfSummary(vol)
    roo.SdfReset(vol, float("nan"), summary=summ)
    f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
    look = samp = 0
    for i in range(3):
        T_wc = scenes.orbit_pose(3 * i, 30)
        roo.BilateralFilter(f, T.upload_image(roo, scenes.render_depth(scene, w, h, T_wc, K)), **scenes.BILATERAL)
        roo.DepthToVbo(vbo, f, K); roo.NormalsFromVbo(nrm, vbo)
        roo.SdfFuse(vol, f, nrm, scenes.se3_inverse(T_wc), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, full_extent=True, summary=summ)
        a = [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
        b = [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
        roo.RaycastSdf(*a, vol, T_wc, K, near, far, tr, True)
        roo.RaycastSdf(*b, vol, T_wc, K, near, far, tr, True, summary=summ)
        torch.cuda.synchronize()
        assert all(T.nan_equal(x.MemcpyToHost(), y.MemcpyToHost()) for x, y in zip(a, b)), (scene, dims, i)
        assert np.isfinite(a[0].MemcpyToHost()).sum() > 0.05 * w * h
        c = roo.RaycastSdfCount(vol, w, h, T_wc, K, near, far, tr, True, summary=summ)
        assert c["table_bytes"] > 0
        look += c["lookups"]; samp += c["samples"]
    out["%%s-%%d" %% (scene, dims[0])] = [look, samp]
print("RES", json.dumps(out))
''' % (T.ROOT, __import__("os").path.join(T.ROOT, "tests"))
    import json
    res = {}
    for levels in ("0", "2"):
        env = dict(__import__("os").environ, KFX_RAYCAST_SUMMARY="1", KFX_RAYCAST_TOP_LEVELS=levels)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0 and "RES" in out.stdout, out.stdout + out.stderr
        res[levels] = json.loads([l for l in out.stdout.splitlines() if l.startswith("RES")][0][4:])
    for key in res["0"]:
        (l0, s0), (l2, s2) = res["0"][key], res["2"][key]
        assert 0 < s2 <= s0, (key, res)          # positions at an entry's rim are sampled rather than skipped: fewer rims, a few samples fewer
        assert l2 <= l0, (key, res)
    assert res["2"]["full-256"][0] < 0.9 * res["0"]["full-256"][0], res   # wide free space: fewer look-ups


def test_gpu_unforced_choice_is_a_function_of_the_calls(tmp_path):
    """Round-3 advice: with KFX_RAYCAST_SUMMARY unset the tracked RaycastSdf chooses between the table march and the plain march
    from a count a table build published -- and in fast numerics the two agree within tolerance only, so the choice must not
    depend on how far the GPU happens to have got.  It is taken from the count of the build before the previous one (a ring of
    published counts): the same sequence of calls renders the same bits, run after run, with the host racing ahead or
    synchronising after every frame.  S_room at 96^3 crosses the quarter threshold while the stream develops, so both kernels
    are in play."""
    import subprocess
    import sys
    code = r'''
import sys, hashlib, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import kfx_testlib as T
from kfx_testlib import scenes
from kangaroo_amd import roo
from kangaroo_amd.pipeline import FramePipeline
sync_every_frame = sys.argv[1] == "1"
roo.set_math_mode("fast")
N, w, h = 96, 200, 150
out = []
for scene in ("room", "full"):
    bmin, bmax, near, far = scenes.SCENES[scene]
    pipe = FramePipeline(roo, (N, N, N), bmin, bmax, w, h, near=near, far=far, track=True)
    frames = [T.upload_image(roo, scenes.render_depth(scene, w, h, scenes.orbit_pose(i, 30), pipe.K)) for i in range(30)]
    keep = []
    for i in range(45):
        pipe.step(scenes.orbit_pose(i %% 30, 30), frames[i %% 30])
        if sync_every_frame:
            torch.cuda.synchronize()
        keep.append(pipe.ray_d.tensor().clone())
    torch.cuda.synchronize()
    hsh = hashlib.sha256()
    for t in keep:
        hsh.update(t.cpu().numpy().tobytes())
    out.append(hsh.hexdigest())
print("HASH", " ".join(out))
''' % (T.ROOT, __import__("os").path.join(T.ROOT, "tests"))
    env = dict(__import__("os").environ)
    env.pop("KFX_RAYCAST_SUMMARY", None)
    hashes = []
    for mode in ("0", "1", "0"):
        out = subprocess.run([sys.executable, "-c", code, mode], capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0 and "HASH" in out.stdout, out.stdout + out.stderr
        hashes.append([l for l in out.stdout.splitlines() if l.startswith("HASH")][0])
    assert hashes[0] == hashes[1] == hashes[2], hashes


@pytest.mark.parametrize("scene", ["full", "room"])
def test_gpu_frame_pipeline_auto_policy(roo, scene):
    """FramePipeline(track="auto"): starts with the summary, times three blocks of whole frames of the stream itself (tracked
    pair, plain pair, tracked pair again -- device events recorded by kfx_frame_step), decides once and carries on with the
    faster pair unless the tables win by the margin.  Whatever it decides, the volume equals the untracked pipeline's bit for
    bit and the images stay within the fast-mode tolerance of the plain march; reset() re-arms the calibration."""
    from kangaroo_amd.pipeline import FramePipeline
    N, w, h = 128, 320, 240
    bmin, bmax, near, far = scenes.SCENES[scene]
    prev = roo.set_math_mode("fast")
    try:
        auto = FramePipeline(roo, (N, N, N), bmin, bmax, w, h, near=near, far=far, track="auto", cal_first=4, cal_block=8)
        ref = FramePipeline(roo, (N, N, N), bmin, bmax, w, h, near=near, far=far, track=False)
        assert auto.kframe is not None and ref.kframe is not None, "the HIP operator set issues frames through kfx_frame_step"
        assert auto.track_policy == "auto" and auto.track and auto.track_decision is None
        states = []
        for i in range(40):
            T_wc = scenes.orbit_pose(i % 30, 30)
            raw = T.upload_image(roo, scenes.render_depth(scene, w, h, T_wc, auto.K))
            auto.step(T_wc, raw)
            ref.step(T_wc, raw)
            states.append(auto.track)
        d = auto.track_decision
        assert d is not None and auto._cal is None, "no decision after 40 frames"
        assert states[4:12] == [True] * 8 and states[12:20] == [False] * 8 and states[20:28] == [True] * 8, states
        assert auto.track == d["chosen"].startswith("table march") and (auto.summary is not None) == auto.track
        assert d["frames_per_block"] == 8 and d["frame_plain_ms"] > 0 and min(d["frame_tracked_ms"]) > 0
        assert d["sdf_fuse_plain_ms"] > 0 and d["sdf_fuse_tracked_ms"] > 0 and d["rest_of_frame_plain_ms"] > 0   # the plain SdfFuse is measured, not assumed
        assert auto.timing == 0   # ... with events recorded during the blocks only
        assert auto.track == (max(d["frame_tracked_ms"]) <= (1 - d["margin"]) * d["frame_plain_ms"])
        assert T.nan_equal(auto.vol.MemcpyToHost(), ref.vol.MemcpyToHost())
        da, dr = auto.ray_d.MemcpyToHost(), ref.ray_d.MemcpyToHost()
        ha, hr = np.isfinite(da), np.isfinite(dr)
        assert (ha != hr).sum() <= max(3, 2e-4 * w * h)
        both = ha & hr
        assert both.sum() > 0.03 * w * h and np.abs(da[both] - dr[both]).max() < 1e-4
        auto.reset()   # a new stream: the policy starts over (round-3 advice)
        assert auto.track and auto.track_decision is None and auto._cal is not None
    finally:
        roo.set_math_mode(prev)


def test_gpu_tracking_pipeline_auto_policy(roo):
    """TrackingPipeline(track="auto"): the loop with pose estimation times whole frames of the three blocks with the host clock
    (its pose read-back synchronises every frame), decides once, and tracks the orbit like the untracked loop: same poses to
    within the fast-mode tolerance of the renderings."""
    from kangaroo_amd.pipeline import TrackingPipeline
    N, w, h = 128, 320, 240
    scene = "room"
    bmin, bmax, near, far = scenes.SCENES[scene]
    prev = roo.set_math_mode("fast")
    try:
        auto = TrackingPipeline(roo, (N, N, N), bmin, bmax, w, h, near=near, far=far, device_icp=True, track="auto", cal_first=4, cal_block=8)
        ref = TrackingPipeline(roo, (N, N, N), bmin, bmax, w, h, near=near, far=far, device_icp=True, track=False)
        assert auto.track_policy == "auto" and auto.track and auto.kframe is None
        worst = 0.0
        for i in range(36):
            T_wc = scenes.orbit_pose(i % 30, 30)
            raw = T.upload_image(roo, scenes.render_depth(scene, w, h, T_wc, auto.K))
            Ta = auto.step(T_wc if i == 0 else None, raw)
            Tr = ref.step(T_wc if i == 0 else None, raw)
            worst = max(worst, float(np.abs(Ta[:3, 3] - Tr[:3, 3]).max()))
            assert auto.tracking_good and ref.tracking_good, i
        d = auto.track_decision
        assert d is not None and auto._cal is None, "no decision after 36 frames"
        assert auto.track == d["chosen"].startswith("table march") and (auto.summary is not None or not auto.track)
        assert d["frames_per_block"] == 8 and d["frame_plain_ms"] > 0 and d["clock"].startswith("host")
        assert worst < 2e-4, worst   # metres: the two loops see renderings that differ within the fast-mode tolerance
    finally:
        roo.set_math_mode(prev)


@pytest.mark.parametrize("unaligned_view", [False, True])
def test_gpu_summary_rebuild_is_exact(roo, unaligned_view):
    """kfx_sdf_summary_rebuild: after frames fused WITHOUT tracking, the rebuilt summary holds for every brick the exact range
    of its valued cells and the exact state -- at least as tight as what tracking keeps -- and the march through tables built
    from it renders the plain march's images bit for bit (exact numerics).  unaligned_view: the summary of a view whose first
    cell sits at an odd x of its parent (pointer 8-byte aligned only: the rebuild's scalar path, eight cells per row and brick
    -- round-4 advice: it read to the end of the row)."""
    import torch
    N, w, h = 96, 200, 150
    dims = (N, N - 12, N - 5)   # not multiples of 8: partial bricks on two axes
    bmin, bmax, near, far = scenes.SCENES["room"]
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(bmin, bmax, dims)
    if unaligned_view:
        parent = roo.BoundedVolume(dims[0] + 3, dims[1] + 2, dims[2] + 1, bmin, bmax)
        roo.SdfReset(parent, 0.25)   # (cells around the view hold a value no cell of the view will: a read beyond a brick shows)
        vol = parent.SubVolume((1, 2, 1), dims)
        assert vol.ptr % 16 == 8
    else:
        vol = roo.BoundedVolume(*dims, bmin, bmax)
    summ = roo.SdfSummary(vol)
    if unaligned_view:   # (SdfReset fills the contiguous span of a view, padding and the parent's cells in between included)
        vol.tensor()[...] = torch.tensor([float("nan"), 0.0], device="cuda")
    else:
        roo.SdfReset(vol, float("nan"))
    f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
    for i in range(3):
        T_wc = scenes.orbit_pose(i, 30)
        raw = T.upload_image(roo, scenes.render_depth("room", w, h, T_wc, K))
        roo.BilateralFilter(f, raw, **scenes.BILATERAL)
        roo.DepthToVbo(vbo, f, K)
        roo.NormalsFromVbo(nrm, vbo)
        roo.SdfFuse(vol, f, nrm, scenes.se3_inverse(T_wc), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, full_extent=True)
    summ.rebuild()
    R, classes = export(roo, summ, 0.0, tr)
    check_classes(vol, classes, 0.0, tr)
    stats = check_conservative(vol, R)
    assert stats["uniform_ranges"] > 0 and stats["all_nan"] > 0 and stats["mixed"] > 0, stats
    # exactness: every brick's range IS the range of its valued cells, every state the true one
    v = vol.tensor()[..., 0]
    d_, h_, w_ = v.shape
    nbz, nby, nbx = R.shape[:3]
    pad = torch.full((nbz * 8, nby * 8, nbx * 8), float("nan"), device=v.device)
    known = torch.zeros_like(pad, dtype=torch.bool)
    pad[:d_, :h_, :w_] = v
    known[:d_, :h_, :w_] = True
    br = pad.view(nbz, 8, nby, 8, nbx, 8).permute(0, 2, 4, 1, 3, 5).reshape(nbz, nby, nbx, 512)
    kn = known.view(nbz, 8, nby, 8, nbx, 8).permute(0, 2, 4, 1, 3, 5).reshape(nbz, nby, nbx, 512)
    has, isn = ~torch.isnan(br) & kn, torch.isnan(br) & kn
    tmin = torch.where(has, br, torch.full_like(br, float("inf"))).amin(-1)
    tmax = torch.where(has, br, torch.full_like(br, float("-inf"))).amax(-1)
    state = R[..., 2].contiguous().view(torch.int32)
    want = torch.where(has.any(-1), torch.where(isn.any(-1), 2, 0), 1).to(torch.int32)
    assert bool((state == want).all())
    assert bool((R[..., 0] == tmin).all()) and bool((R[..., 1] == tmax).all())
    T_wc = scenes.orbit_pose(2, 30)
    a = [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
    b = [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
    roo.RaycastSdf(*a, vol, T_wc, K, near, far, tr, True)
    roo.RaycastSdf(*b, vol, T_wc, K, near, far, tr, True, summary=summ)
    for x, y in zip(a, b):
        assert T.nan_equal(x.MemcpyToHost(), y.MemcpyToHost())


@pytest.mark.parametrize("trunc_factor", [0.4, 1.0, 6.0])
def test_gpu_table_march_with_unusual_truncation(roo, trunc_factor):
    """trunc_dist below the voxel size (the reference's step for a +trunc sample is then min_delta, not trunc: class 3 --
    "NaN or +trunc" -- must not be skipped), equal to it, and several voxels wide; and a raycast whose trunc_dist differs from
    the one the volume was fused with (no entry holds the raycast's value: only NaN space is skipped).  Exact numerics:
    images through the tables = images of the plain march, bit for bit."""
    N, w, h = 96, 200, 150
    scene = "room"
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h)
    voxel = (bmax[0] - bmin[0]) / (N - 1)
    tr = float(np.float32(trunc_factor * voxel))
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
        for ray_tr in (tr, float(np.float32(1.7 * tr))):
            a = [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
            b = [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
            roo.RaycastSdf(*a, vol, T_wc, K, near, far, ray_tr, True)
            roo.RaycastSdf(*b, vol, T_wc, K, near, far, ray_tr, True, summary=summ)
            for x, y in zip(a, b):
                assert T.nan_equal(x.MemcpyToHost(), y.MemcpyToHost()), (trunc_factor, i, ray_tr, T.mismatch_report(x.MemcpyToHost(), y.MemcpyToHost()))
            assert np.isfinite(a[0].MemcpyToHost()).sum() > (100 if trunc_factor < 1 else 0.02 * w * h)   # (a band thinner than the minimum step is mostly stepped over)


@pytest.mark.parametrize("scene", ["full", "room"])
def test_gpu_free_space_keeps_its_value_over_a_long_stream(roo, scene):
    """Fast numerics, 1500 frames of the orbit: a cell that is handed +trunc every frame must keep +trunc bit for bit (the
    running average is evaluated as old + (new - old) w / (w + ow)), so the class tables see as much free space at the end of
    the stream as after its first orbits.  With (w val + ow oval) * rcp(w + ow) the stored value crept away from trunc by the
    reciprocal's bias and after ~800 frames the table march had nothing left to skip."""
    import torch
    N, w, h = 128, 160, 120
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    prev = roo.get_math_mode()
    roo.set_math_mode("fast")
    try:
        vol = roo.BoundedVolume(N, N, N, bmin, bmax)
        summ = roo.SdfSummary(vol)
        roo.SdfReset(vol, float("nan"), summary=summ)
        frames = []
        for i in range(30):
            T_wc = scenes.orbit_pose(i, 30)
            f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
            roo.BilateralFilter(f, T.upload_image(roo, scenes.render_depth(scene, w, h, T_wc, K)), **scenes.BILATERAL)
            roo.DepthToVbo(vbo, f, K)
            roo.NormalsFromVbo(nrm, vbo)
            frames.append((scenes.se3_inverse(T_wc), f, nrm))
        counts = {}
        for k in range(1500):
            T_cw, f, nrm = frames[k % 30]
            roo.SdfFuse(vol, f, nrm, T_cw, K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, summary=summ)
            if k + 1 in (60, 1500):
                _, classes = export(roo, summ, 1e-5, tr, fine_shift=3)
                counts[k + 1] = check_classes(vol, classes, 1e-5, tr)
            if k + 1 == 60:
                free60 = (vol.tensor()[..., 0] == tr).clone()
        free1500 = vol.tensor()[..., 0] == tr
        assert int(free60.sum()) > N ** 3 // 20, "the scene should leave observed free space"
        assert bool((free1500 | ~free60).all()), "%d cells held +trunc after two orbits and no longer do" % int((free60 & ~free1500).sum())
        for shift in counts[60]:
            n60, n1500 = counts[60][shift][1] + counts[60][shift][3], counts[1500][shift][1] + counts[1500][shift][3]
            assert n1500 >= n60 and (n60 > 0 or shift != 3), (shift, counts)
    finally:
        roo.set_math_mode(prev)
