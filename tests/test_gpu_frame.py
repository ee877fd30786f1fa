"""kfx_frame (include/kfx.h): one frame of the application's loop (main.cpp:200-356, known poses) as ONE library call.

  * a step writes exactly what the separate operator calls write -- filtered depth, vertex / normal maps, volume, raycast
    images -- in both numerics modes, with and without the brick summary, through views and partial steps;
  * the timing ring: four device events per frame, read back as preprocess / SdfFuse / RaycastSdf / frame / period;
  * switching the summary off and on again (kfx_sdf_summary_rebuild) leaves a summary the march can trust."""
import numpy as np
import pytest

import kfx_testlib as T
from kfx_testlib import scenes

pytestmark = pytest.mark.gpu


def _operators_frame(roo, vol, imgs, K, T_wc, raw, tr, near, far, summary=None):
    f, vbo, nrm, rd, rn, ri = imgs
    roo.BilateralFilter(f, raw, **scenes.BILATERAL)
    roo.DepthToVbo(vbo, f, K)
    roo.NormalsFromVbo(nrm, vbo)
    kw = {"summary": summary} if summary is not None else {}
    roo.SdfFuse(vol, f, nrm, scenes.se3_inverse(T_wc), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, **kw)
    roo.RaycastSdf(rd, rn, ri, vol, T_wc, K, near, far, tr, True, **kw)


@pytest.mark.parametrize("math", ["exact", "fast"])
@pytest.mark.parametrize("track", [False, True])
def test_gpu_frame_step_equals_the_separate_calls(roo, math, track):
    import torch
    N, w, h = 96, 240, 180
    scene = "room"
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    prev = roo.set_math_mode(math)
    try:
        va, vb = roo.BoundedVolume(N, N, N, bmin, bmax), roo.BoundedVolume(N, N, N, bmin, bmax)
        mk = lambda: [roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4"), roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h)]
        ia, ib = mk(), mk()
        raw_slot = roo.Image(w, h)
        fr = roo.Frame(vb, raw_slot, ib[0], ib[1], ib[2], ib[3], ib[4], ib[5], K, scenes.BILATERAL, near, far, tr, scenes.MAX_W, scenes.MIN_COS_THETA,
                       timing_slots=16)
        summ = roo.SdfSummary(va) if track else None
        roo.SdfReset(va, float("nan"), summary=summ)
        fr.set_track(track)
        fr.reset()
        assert fr.track == track and fr.count == 0 and (fr.summary() is not None) == track
        for i in range(5):
            T_wc = scenes.orbit_pose(i, 30)
            raw = T.upload_image(roo, scenes.render_depth(scene, w, h, T_wc, K))
            _operators_frame(roo, va, ia, K, T_wc, raw, tr, near, far, summ)
            if i == 2:   # the frame's parts one by one, the raw image taken from the frame's own slot
                raw_slot.MemcpyFromHost(scenes.render_depth(scene, w, h, T_wc, K))
                fr.step(T_wc, None, None, fr.PREPROCESS)
                fr.step(T_wc, scenes.se3_inverse(T_wc), None, fr.FUSE)
                fr.step(T_wc, None, None, fr.RAYCAST)
            else:
                fr.step(T_wc, scenes.se3_inverse(T_wc), raw)
            torch.cuda.synchronize()
            for a, b in zip(ia, ib):
                assert T.nan_equal(a.MemcpyToHost(), b.MemcpyToHost()), (i, a.kind)
            assert T.nan_equal(va.MemcpyToHost(), vb.MemcpyToHost()), i
        assert fr.count == 7
        t = fr.timings(0, 7)
        assert t.shape == (7, 5) and np.all(t[:, :4] >= 0) and np.all(np.isfinite(t[:6, 4])) and np.isnan(t[6, 4])
        assert np.all(t[[0, 1, 5, 6], 1] > 0) and np.all(t[[0, 1, 5, 6], 2] > 0)          # whole frames: SdfFuse and RaycastSdf took time
        assert t[2, 1] < 0.5 * t[3, 1] and t[3, 2] < 0.5 * t[4, 2]                         # the partial steps ran their part only
        assert np.all(t[:6, 4] >= t[:6, 3] * 0.999)                                        # start to next start covers the frame
        # T_cw = NULL: the library inverts T_wc itself (double, rounded once): within an ulp or two of the float32 inverse
        T_wc = scenes.orbit_pose(7, 30)
        fr.step(T_wc, None, raw)
        with pytest.raises(roo.KfxError):
            fr.timings(0, 40)   # more frames than were stepped
        # kfx_frame_set_timing: only the two events around SdfFuse (what a loop that is being timed records), then none
        fr.set_timing(fr.EVENTS_FUSE)
        n0 = fr.count
        for i in range(3):
            fr.step(scenes.orbit_pose(i, 30), None, raw)
        fr.set_timing(fr.EVENTS_NONE)
        fr.step(T_wc, None, raw)
        t = fr.timings(n0, 4)
        assert np.all(t[:3, 1] > 0) and np.all(np.isnan(t[:3, [0, 2, 3]])), t      # SdfFuse window only
        assert np.all(t[:2, 4] > t[:2, 1]) and np.isnan(t[2, 4])                   # period: before-SdfFuse to before-SdfFuse; the next frame recorded nothing
        assert np.all(np.isnan(t[3]))
        # a frame with the SdfFuse events only followed by one with all four, read back without a synchronisation in between: the
        # period ends at the next frame's before-SdfFuse event (not at its first event), and timings() waits for that one
        fr.set_timing(fr.EVENTS_FUSE)
        n1 = fr.count
        fr.step(T_wc, None, raw)
        fr.set_timing(fr.EVENTS_ALL)
        for i in range(3):
            fr.step(scenes.orbit_pose(i, 30), None, raw)
        t = fr.timings(n1, 1)
        assert t[0, 4] > t[0, 1] > 0 and np.all(np.isnan(t[0, [0, 2, 3]])), t
    finally:
        roo.set_math_mode(prev)


def test_gpu_frame_track_off_and_on_again(roo):
    """The auto policy's block pattern: tracked frames, plain frames (the summary goes stale), tracked frames again after
    kfx_sdf_summary_rebuild -- the volume equals an always-plain pipeline's bit for bit and, in exact numerics, so do the images
    (the table march built from the rebuilt summary is the plain march)."""
    import torch
    from kangaroo_amd.pipeline import FramePipeline
    N, w, h = 96, 240, 180
    bmin, bmax, near, far = scenes.SCENES["room"]
    a = FramePipeline(roo, (N, N, N), bmin, bmax, w, h, near=near, far=far, track=True)
    b = FramePipeline(roo, (N, N, N), bmin, bmax, w, h, near=near, far=far, track=False)
    assert a.kframe is not None and a.track and not b.track
    for i in range(12):
        if i == 4:
            a.set_track(False)
            assert a.summary is None and not a.kframe.track
        if i == 8:
            a.set_track(True)
            assert a.summary is not None and a.kframe.track
        T_wc = scenes.orbit_pose(i, 30)
        raw = T.upload_image(roo, scenes.render_depth("room", w, h, T_wc, a.K))
        a.step(T_wc, raw)
        b.step(T_wc, raw)
        torch.cuda.synchronize()
        assert T.nan_equal(a.vol.MemcpyToHost(), b.vol.MemcpyToHost()), i
        for x, y in ((a.ray_d, b.ray_d), (a.ray_n, b.ray_n), (a.ray_i, b.ray_i)):
            assert T.nan_equal(x.MemcpyToHost(), y.MemcpyToHost()), i
