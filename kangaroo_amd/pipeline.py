"""Headless KinectFusion frame loop on the `roo::` operators (known poses, no ICP).

Mirrors the per-frame call order of the reference application
(applications/kinectfusion/main.cpp:200-356):

    BilateralFilter -> DepthToVbo -> NormalsFromVbo          (main.cpp:209-215)
    [RaycastSdf of the model from the current pose]           (main.cpp:286)
    SdfFuse of the new frame                                  (main.cpp:345-356)

`FramePipeline` is the single-GPU loop.  `SlabPipeline` partitions the volume into Z-slabs,
one per rank (one process per GPU): SdfFuse is embarrassingly parallel per voxel, so every
rank integrates its own slab (+ ghost planes, recomputed redundantly -- the update is
deterministic per voxel, so neighbours hold bit-identical ghosts without any exchange) and
ray-casts its own slab; the per-rank images are merged with one min-reduction over
(depth, rank) keys and one sum-reduction of the winner's payload (RCCL over xGMI through
torch.distributed; MiB-scale, latency-bound messages).

The operator set is injected (`ops`): the product passes `kangaroo_amd.roo` (HIP kernels behind
the C ABI); the CPU tests inject an oracle-backed stand-in from tests/ to exercise the
partitioning and compositing logic under gloo without a GPU.
"""
import numpy as np

from . import scenes


def vbo_normals(ops, vbo, normals, depth, K):
    """DepthToVbo + NormalsFromVbo (main.cpp:213-214); one launch where the operator set offers the fused entry point
    (identical outputs), else the two operators."""
    if hasattr(ops, "DepthToVboNormals"):
        ops.DepthToVboNormals(vbo, normals, depth, K)
    else:
        ops.DepthToVbo(vbo, depth, K)
        ops.NormalsFromVbo(normals, vbo)


class FramePipeline:
    # track="auto": the stream itself is the benchmark.  From frame CAL_FIRST on, three blocks of CAL_BLOCK frames each:
    # tracked pair (SdfFuse keeping the summary + march through the class tables), plain pair (the summary goes stale), tracked
    # pair again (summary rebuilt from the volume, kfx_sdf_summary_rebuild) -- whole frames, timed start to start.  The tables
    # stay only if BOTH tracked blocks beat the plain block's median frame by CAL_MARGIN; otherwise the plain pair runs from
    # then on.  Nothing is assumed about what tracking costs SdfFuse: the plain SdfFuse is in the plain block's frames.
    CAL_FIRST, CAL_BLOCK, CAL_MARGIN = 8, 60, 0.05
    USE_FRAME = True   # issue a frame through ONE library call (ops.Frame = kfx_frame) where the operator set has it

    def __init__(self, ops, dims, boxmin, boxmax, w, h, K=None, near=0.4, far=8.0, bilateral=None,
                 trunc_factor=scenes.TRUNC_DIST_FACTOR, max_w=scenes.MAX_W, mincostheta=scenes.MIN_COS_THETA,
                 contiguous_images=False, track=False, cal_first=None, cal_block=None, cal_margin=None, timing_slots=None):
        """track: keep a brick summary of the volume (ops.SdfSummary) current in SdfFuse and let RaycastSdf step through
        uniformly free / never-observed space without reading the volume (same volume bits; images bit-identical in
        exact numerics, within the fast-mode tolerance in fast numerics).  True / False, or "auto": start with the summary,
        time whole frames of the stream itself with and without it and keep whichever makes the frame shorter (see CAL_*):
        the table march wins where rays cross much free space (S_full: RaycastSdf 0.15 -> 0.05 ms) and loses where the
        longest rays graze silhouettes (S_room), and which it is depends on the scene, not on anything known up front.
        During the calibration blocks the images come from the march of the block (bit-identical in exact numerics, within
        the fast-mode tolerance otherwise); reset() re-arms the calibration for the new stream, recalibrate() at any frame."""
        self.ops = ops
        self.track_policy = "auto" if track == "auto" else ("on" if track else "off")
        self.track = bool(track) and hasattr(ops, "SdfSummary")
        if self.track_policy == "auto" and not self.track:
            self.track_policy = "off"
        self.cal_first = self.CAL_FIRST if cal_first is None else int(cal_first)
        self.cal_block = self.CAL_BLOCK if cal_block is None else int(cal_block)
        self.cal_margin = self.CAL_MARGIN if cal_margin is None else float(cal_margin)
        self.track_decision = None   # auto: dict(chosen=..., ...) once decided
        self.frames_done = 0         # frames stepped since construction / reset
        self._cal = None
        self.summary = None
        self.kframe = None
        self.dims = tuple(int(d) for d in dims)
        self.w, self.h = int(w), int(h)
        self.K = scenes.intrinsics(w, h) if K is None else np.asarray(K, np.float32)
        self.near, self.far = float(near), float(far)
        self.bil = dict(scenes.BILATERAL if bilateral is None else bilateral)
        self.max_w, self.mincostheta = float(max_w), float(mincostheta)
        # trunc_dist = factor * length(VoxelSizeUnits()) of the FULL volume (main.cpp:221)
        self.trunc = scenes.trunc_dist(boxmin, boxmax, self.dims, trunc_factor)
        self.vol = self._alloc_volume(boxmin, boxmax)
        pf = (lambda e: self.w * e) if contiguous_images else (lambda e: None)
        I = ops.Image
        self.raw = I(w, h, "f32", pitch=pf(4))
        self.filtered = I(w, h, "f32", pitch=pf(4))
        self.vbo = I(w, h, "f32x4", pitch=pf(16))
        self.normals = I(w, h, "f32x4", pitch=pf(16))
        self.ray_d = I(w, h, "f32", pitch=pf(4))
        self.ray_n = I(w, h, "f32x4", pitch=pf(16))
        self.ray_i = I(w, h, "f32", pitch=pf(4))
        if self.USE_FRAME and hasattr(ops, "Frame") and getattr(self.vol, "kind", "f32") == "f32":
            self.kframe = ops.Frame(self.vol, self.raw, self.filtered, self.vbo, self.normals, self.ray_d, self.ray_n, self.ray_i, self.K,
                                    self.bil, self.near, self.far, self.trunc, self.max_w, self.mincostheta,
                                    timing_slots=max(256, 3 * self.cal_block + 16, int(timing_slots or 0)))
        # the frame's device events are markers between launches and cost the stream a few microseconds each (all four: 2.7 % of a
        # 0.42 ms frame): none are recorded unless the caller asks (set_timing) or the auto policy is timing its blocks
        self.timing = 0
        if self.kframe is not None:
            self.kframe.set_timing(self.timing)
        track_now, self.track = self.track, False
        self.set_track(track_now)
        self.reset()

    # -- overridable pieces ------------------------------------------------------
    def _alloc_volume(self, boxmin, boxmax):
        return self.ops.BoundedVolume(self.dims[0], self.dims[1], self.dims[2], boxmin, boxmax)

    def set_track(self, on):
        """Switch between the tracked pair of kernels and the plain pair.  Switching on (re)builds the summary from what the
        volume holds, so it may follow any number of untracked frames."""
        on = bool(on) and hasattr(self.ops, "SdfSummary")
        if self.kframe is not None:
            self.kframe.set_track(on)
            self.summary = self.kframe.summary() if on else None
        elif on and not self.track:
            if self.summary is None:
                self.summary = self.ops.SdfSummary(self.vol)
            self.summary.rebuild()
        self.track = on

    def set_timing(self, mask):
        """Which device events kfx_frame_step records from now on (ops.Frame.EVENTS_ALL / EVENTS_FUSE / EVENTS_NONE): what
        self.kframe.timings() can answer afterwards."""
        self.timing = int(mask)
        if self.kframe is not None:
            self.kframe.set_timing(self.timing)

    def recalibrate(self, first=None):
        """track="auto": time the three blocks again, starting `first` frames from now (default: at once) -- e.g. once a
        stream has reached its steady state."""
        if self.track_policy != "auto":
            return
        self.track_decision = None
        self._cal = {"first": self.frames_done + (0 if first is None else int(first)), "t": [], "clock": None}
        self.set_track(True)

    def reset(self):
        """SdfReset(vol, NaN): 'never observed' = (NaN, 0) (main.cpp:229).  A new stream: track="auto" starts over with the
        summary and calibrates on the new stream."""
        self.frames_done = 0
        if self.track_policy == "auto":
            self.track_decision = None
            # cal_first < 0: no calibration until the caller asks for one (recalibrate()); the tracked pair runs meanwhile
            self._cal = {"first": self.cal_first, "t": [], "clock": None} if self.cal_first >= 0 else None
            if not self.track:
                self.set_track(True)
        if self.kframe is not None:
            self.kframe.reset()
        elif self.track:
            self.ops.SdfReset(self.vol, float("nan"), summary=self.summary)
        else:
            self.ops.SdfReset(self.vol, float("nan"))

    def preprocess(self, raw_image=None):
        if self.kframe is not None:
            self.kframe.step(_IDENTITY, None, raw_image, self.kframe.PREPROCESS)
            return
        o = self.ops
        src = self.raw if raw_image is None else raw_image
        o.BilateralFilter(self.filtered, src, self.bil["gs"], self.bil["gr"], self.bil["size"], self.bil["minval"])
        vbo_normals(o, self.vbo, self.normals, self.filtered, self.K)

    # -- the auto policy ------------------------------------------------------------
    def _policy_before(self):
        """Called at the start of a frame while a calibration is pending: puts the pipeline into the state of the frame's block."""
        c, B = self._cal, self.cal_block
        k = self.frames_done - c["first"]   # this frame's index within the blocks
        if k == 0 and self.kframe is not None:
            c["kfirst"] = self.kframe.count
            self.kframe.set_timing(self.timing | self.kframe.EVENTS_FUSE)   # the SdfFuse window and the frame period, for all blocks alike
        if k == B:
            self.set_track(False)
        elif k == 2 * B:
            self.set_track(True)
        if self.kframe is None:   # host clock (loops that synchronise every frame: the tracking loop's pose read-back)
            import time
            c["clock"] = time.perf_counter()

    def _policy_after(self):
        """Called at the end of a frame (frames_done counts it already): once the three blocks and one more frame -- whose start
        ends the last period -- have been issued, read their times and decide."""
        c, B = self._cal, self.cal_block
        k = self.frames_done - 1 - c["first"]
        if self.kframe is None and 0 <= k < 3 * B and c["clock"] is not None:
            import time
            c["t"].append((time.perf_counter() - c["clock"]) * 1e3)
        if k < 3 * B:
            return
        parts = None
        if self.kframe is not None:
            if self.kframe.count - c.get("kfirst", -1) != self.frames_done - c["first"]:
                # the caller issued parts of frames by themselves in between (preprocess / fuse / raycast): the ring's frames are
                # not the blocks' frames -- start over from here
                self._cal = {"first": self.frames_done, "t": [], "clock": None}
                self.set_track(True)
                return
            t = self.kframe.timings(c["kfirst"], 3 * B)   # waits for the last block's frames only; what is queued behind keeps the GPU busy
            self.kframe.set_timing(self.timing)
            period = t[:, 4].astype(np.float64)
            parts = t
        else:
            period = np.asarray(c["t"][:3 * B], np.float64)
        med = [float(np.median(period[i * B:(i + 1) * B])) for i in range(3)]
        keep = max(med[0], med[2]) <= (1.0 - self.cal_margin) * med[1]
        d = {"chosen": "table march (tracked SdfFuse)" if keep else "plain march",
             "frame_tracked_ms": [round(med[0], 5), round(med[2], 5)], "frame_plain_ms": round(med[1], 5),
             "frames_per_block": B, "first_frame": c["first"], "margin": self.cal_margin,
             "rule": "tables stay iff max(tracked block medians) <= (1 - margin) x plain block median (whole frames)",
             "clock": "device events, frame start to next frame start" if self.kframe is not None else "host clock around step()"}
        if parts is not None:   # what the frames were made of: SdfFuse, and the rest (preprocess, table build, RaycastSdf, gaps)
            fuse, rest = parts[:, 1].astype(np.float64), period - parts[:, 1]
            for name, v in (("sdf_fuse", fuse), ("rest_of_frame", rest)):
                d[name + "_tracked_ms"] = round(float(np.median(np.concatenate([v[:B], v[2 * B:]]))), 5)
                d[name + "_plain_ms"] = round(float(np.median(v[B:2 * B])), 5)
        self.track_decision = d
        self._cal = None
        if not keep:
            self.set_track(False)

    def _timed_fuse(self, integrate):
        """integrate(kw) launches the frame's SdfFuse with kw = {"summary": ...} or {} (loops that issue their own operators)."""
        integrate({"summary": self.summary} if self.track else {})

    def _timed_raycast(self, render):
        """render(kw) launches the frame's rendering(s) of the model."""
        render({"summary": self.summary} if self.track else {})

    def fuse(self, T_wc):
        if self.kframe is not None:
            self.kframe.step(T_wc, scenes.se3_inverse(T_wc), None, self.kframe.FUSE)
            return
        self._timed_fuse(lambda kw: self.ops.SdfFuse(self.vol, self.filtered, self.normals, scenes.se3_inverse(T_wc), self.K, self.trunc,
                                                     self.max_w, self.mincostheta, **kw))

    def raycast(self, T_wc):
        if self.kframe is not None:
            self.kframe.step(T_wc, None, None, self.kframe.RAYCAST)
            return
        self._timed_raycast(lambda kw: self.ops.RaycastSdf(self.ray_d, self.ray_n, self.ray_i, self.vol, T_wc, self.K, self.near, self.far,
                                                           self.trunc, True, **kw))

    def step(self, T_wc, raw_image=None):
        """One frame: preprocess the new depth image, integrate it, render the model."""
        cal = self._cal is not None and self.frames_done >= self._cal["first"]
        if cal:
            self._policy_before()
        if self.kframe is not None:
            self.kframe.step(T_wc, scenes.se3_inverse(T_wc), raw_image)
        else:
            self.preprocess(raw_image)
            self.fuse(T_wc)
            self.raycast(T_wc)
        self.frames_done += 1
        if cal:
            self._policy_after()


_IDENTITY = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], np.float32)


class TrackingPipeline(FramePipeline):
    """The reference application's loop with pose estimation switched on (main.cpp:200-356): depth pyramid,
    per-level vertex / normal maps, raycast of the model at the last pose on every level that has ICP
    iterations, coarse-to-fine projective point-plane ICP (kangaroo_amd/tracking.py around the HIP operator),
    then SdfFuse at the refined pose when tracking is good."""

    LEVELS = 4
    USE_FRAME = False   # its frame has the pose estimation between rendering and integration: the operators are issued here

    def __init__(self, ops, dims, boxmin, boxmax, w, h, its=None, icp_c=0.1, max_rmse=0.10, device_icp=False, one_raycast=None, **kw):
        """device_icp: run the whole refinement loop on the GPU (ops.IcpRefine, one synchronisation per frame) instead
        of one PoseRefinementProjectiveIcpPointPlane call + host solve per iteration.  one_raycast: render all pyramid
        levels with one launch (ops.RaycastSdfLevels; default: when the operator set has it)."""
        from . import tracking
        super().__init__(ops, dims, boxmin, boxmax, w, h, **kw)
        self.tracking = tracking
        self.device_icp = bool(device_icp) and hasattr(ops, "IcpRefine")
        self.one_raycast = hasattr(ops, "RaycastSdfLevels") if one_raycast is None else (bool(one_raycast) and hasattr(ops, "RaycastSdfLevels"))
        self.its = tuple(tracking.DEFAULT_ITS if its is None else its)
        self.icp_c, self.max_rmse = float(icp_c), float(max_rmse)
        L = self.LEVELS
        P = ops.Pyramid
        self.kin_d, self.kin_v, self.kin_n = P(w, h, L, "f32"), P(w, h, L, "f32x4"), P(w, h, L, "f32x4")
        # a second set for the NEXT frame's pre-amble (step(..., next_image=...)): enqueued while the pose of this frame is on
        # its way to the host, it must not overwrite the maps SdfFuse is about to read
        self._kin_back = None
        self._prefetched = None   # the image whose pre-amble the back set holds
        self._bound = {}          # SdfFuseBound per set of maps
        self.pyr_d, self.pyr_i = P(w, h, L, "f32"), P(w, h, L, "f32")
        self.pyr_n, self.pyr_v = P(w, h, L, "f32x4"), P(w, h, L, "f32x4")
        self.K_levels = [scenes.intrinsics_level(self.K, l) for l in range(L)]
        # main.cpp:110-111: debug image and a scratch image of w * sizeof(LeastSquaresSystem<float,12>) x h bytes
        self.debug = ops.Image(w, h, "f32x4")
        self.scratch = ops.Image(w * 232, h, "u8")
        self.T_wl = np.eye(4)
        self.frame = 0
        self.rmse, self.tracking_good = 0.0, True
        self.resets = 0   # recoveries after tracking was lost altogether (main.cpp:223-242)

    def preprocess(self, raw_image=None, into=None):
        o = self.ops
        src = self.raw if raw_image is None else raw_image
        kin_d, kin_v, kin_n = (self.kin_d, self.kin_v, self.kin_n) if into is None else into
        o.BilateralFilter(kin_d[0], src, self.bil["gs"], self.bil["gr"], self.bil["size"], self.bil["minval"])
        if hasattr(o, "DepthPyramidVboNormals"):   # the pyramid and its maps from one launch (same images)
            o.DepthPyramidVboNormals(kin_d, kin_v, kin_n, self.K_levels)
            return
        o.BoxReduceIgnoreInvalid(kin_d)
        for l in range(self.LEVELS):
            vbo_normals(o, kin_v[l], kin_n[l], kin_d[l], self.K_levels[l])

    def _prefetch(self, image):
        """The pre-amble of `image` into the back set of maps (called between enqueueing the refinement and waiting for its pose)."""
        if self._kin_back is None:
            P, L = self.ops.Pyramid, self.LEVELS
            self._kin_back = (P(self.w, self.h, L, "f32"), P(self.w, self.h, L, "f32x4"), P(self.w, self.h, L, "f32x4"))
        self.preprocess(image, into=self._kin_back)
        self._prefetched = image

    def step(self, T_wl_init=None, raw_image=None, next_image=None):
        """One frame.  The first frame is fused at T_wl_init (identity if None); later frames are tracked
        against the model.  Returns the current T_wl (4x4 float64).
        next_image: the depth image the NEXT step will be given (already in device memory).  With the device-resident
        refinement its pre-amble -- which does not depend on any pose -- is enqueued behind the refinement, before this thread
        waits for the pose: the device works on it while the thread wakes up instead of idling until SdfFuse arrives.  The
        next step finds its maps ready when it is given the same image object; same images, same poses."""
        o, tr = self.ops, self.tracking
        cal = self._cal is not None and self.frames_done >= self._cal["first"]
        if cal:   # track="auto": whole frames of the three blocks, host clock around the step (the pose read-back synchronises)
            self._policy_before()
        if raw_image is not None and self._prefetched is raw_image:
            (self.kin_d, self.kin_v, self.kin_n), self._kin_back = self._kin_back, (self.kin_d, self.kin_v, self.kin_n)
        else:
            self.preprocess(raw_image)
        self._prefetched = None
        # main.cpp:223-242, `if (Pushed(reset) || !std::isfinite(f_rmse))`: when the last refinement found no correspondence at all
        # (rmse = sqrt(0 / 0): a frame without depth, a model out of view) the application starts over -- T_wl = identity (the world
        # frame restarts at the current camera; T_wl_init if the caller gives one), the volume back to "never observed", the current
        # frame fused -- and the frame then goes on like any other: it is tracked against the model it has just founded.
        recover = self.frame > 0 and not np.isfinite(self.rmse)
        if recover:
            self.T_wl = np.eye(4)
            self._reset_model()
            self.rmse, self.tracking_good = 0.0, True
            self.resets += 1
        if self.frame == 0 or recover:
            if T_wl_init is not None:
                self.T_wl = np.vstack([np.asarray(T_wl_init, np.float64).reshape(3, 4), [0, 0, 0, 1]])
            self._fuse_at(self.T_wl)
        if self.frame > 0:
            T34 = self.T_wl[:3].astype(np.float32)
            lv = [l for l in range(self.LEVELS) if self.its[l] > 0]

            def render(kw):
                if self.one_raycast:   # the per-level RaycastSdf + DepthToVbo calls as one launch: same images, overlapping marches
                    o.RaycastSdfLevels([(self.pyr_d[l], self.pyr_n[l], self.pyr_i[l], self.pyr_v[l]) for l in lv], self.vol, T34,
                                       [self.K_levels[l] for l in lv], self.near, self.far, self.trunc, True, **kw)
                else:
                    for l in lv:
                        o.RaycastSdf(self.pyr_d[l], self.pyr_n[l], self.pyr_i[l], self.vol, T34, self.K_levels[l], self.near,
                                     self.far, self.trunc, True, **kw)
                        o.DepthToVbo(self.pyr_v[l], self.pyr_d[l], self.K_levels[l])
            self._timed_raycast(render)
            if self.device_icp:
                hook = (lambda: self._prefetch(next_image)) if next_image is not None else None
                T_lp, self.rmse, _, self.tracking_good = o.IcpRefine(self.kin_v, self.pyr_v, self.pyr_n, self.K_levels, self.its,
                                                                     self.icp_c, self.max_rmse, self.scratch, self.debug, before_wait=hook)
            else:
                T_lp, self.rmse, self.tracking_good = tr.refine_pose(o, self.kin_v, self.pyr_v, self.pyr_n, self.K_levels,
                                                                     self.scratch, self.debug, self.its, self.icp_c, self.max_rmse)
            if self.tracking_good:
                if hasattr(o, "PoseStep") and hasattr(o, "SdfFuseBound"):
                    # the device idles from the pose's arrival to the SdfFuse launch: the update in one host call, the launch with
                    # its arguments bound beforehand
                    self.T_wl, T_cw = o.PoseStep(self.T_wl, T_lp)
                    self._fuse_bound(T_cw)
                else:
                    self.T_wl = self.T_wl @ tr.se3_inv(T_lp)
                    self._fuse_at(self.T_wl)
        self.frame += 1
        self.frames_done += 1
        if cal:
            self._policy_after()
        return self.T_wl

    def _fuse_bound(self, T_cw):
        """SdfFuse of the frame's maps at T_cw (3x4 float32) through a call whose other arguments were marshalled once (per set of
        maps and per tracked / plain choice)."""
        key = (id(self.kin_d[0]), bool(self.track), id(self.summary) if self.track else 0)
        b = self._bound.get(key)
        if b is None:
            b = self._bound[key] = self.ops.SdfFuseBound(self.vol, self.kin_d[0], self.kin_n[0], self.K, self.trunc, self.max_w, self.mincostheta,
                                                         summary=self.summary if self.track else None)
        b(T_cw)

    def _fuse_at(self, T_wl):
        T_cw = self.tracking.se3_inv(T_wl)[:3].astype(np.float32)
        self._timed_fuse(lambda kw: self.ops.SdfFuse(self.vol, self.kin_d[0], self.kin_n[0], T_cw, self.K, self.trunc, self.max_w,
                                                     self.mincostheta, **kw))

    def _reset_model(self):
        """SdfReset(vol, NaN) (main.cpp:229), the brick summary set to match."""
        if self.track:
            self.ops.SdfReset(self.vol, float("nan"), summary=self.summary)
        else:
            self.ops.SdfReset(self.vol, float("nan"))


def slab_range(d, rank, world):
    """Z-planes [z0, z1) owned by `rank`: contiguous, sizes differ by at most one plane."""
    base, rem = divmod(d, world)
    z0 = rank * base + min(rank, rem)
    return z0, z0 + base + (1 if rank < rem else 0)


class SlabPipeline(FramePipeline):
    """Z-slab partition across ranks.  Rank r stores planes
    [z0 - G, z1 + G) clipped to the volume (G ghost planes each side) as an ordinary
    BoundedVolume whose bbox is VoxelPositionInUnits of its first / last stored plane -- the
    view BoundedVolume::SubBoundingVolume produces (BoundedVolume.h:156-164)."""

    GHOST = 2  # >= 1 for the trilinear z+1 corner and the gradient's z-1 / z+1 cells (Volume.h:240-289)
    USE_FRAME = False   # slabs: the operators take slab arguments and collectives sit between them

    def __init__(self, ops, dist, dims, boxmin, boxmax, w, h, halo="exchange", raycast="exact", kind="f32", overlap=False,
                 inputs="replicate", images="all", merge="direct", driver="python", comm=None, tiles=0, unchecked=False, pipeline=3, ghost=None, **kw):
        """driver = "c": every frame is ONE library call per rank (kfx_slab_frame_step, include/kfx_slab.h: the launches AND the
        collectives are enqueued by the library through `comm`, a kangaroo_amd.slab.Comm -- RCCL for one process per GPU; default:
        Comm.torch(dist), the collectives of the process group the caller has set up); driver = "python": this class issues the
        operators and torch.distributed collectives one by one (the cross-check, and what the CPU tests run on the oracle-backed
        operator set).  Same bits either way.  tiles: row-tiles of the exact hand-over (C driver; 0 = the library's default);
        unchecked: do not fail when the exact march leaves rays open (loop-back measurements of ONE rank: scripts/slab_host_floor.py).
        ghost: ghost planes per side; None = 2, or -- driver "c", exact raycast, recomputed ghost planes -- kfx_slab_exact_ghost's width
        (as far as a hit can fall back behind the sample that found it + the gradient stencil: every rank finalises the hits it
        finds and the hand-over drops its last stage, include/kfx_slab.h), where every rank owns that many planes.

        halo = "exchange": every rank integrates only the planes it owns and the ghost planes are
        refreshed from the two neighbours after each SdfFuse (point-to-point send/recv: one xGMI link
        per direction); halo = "recompute": every rank integrates its ghost planes itself (the update is
        deterministic per voxel, so no traffic is needed) -- the cross-check of the exchange path."""
        assert halo in ("exchange", "recompute")
        assert raycast in ("composite", "exact", "exact_allreduce")
        # inputs (SURVEY.md 8(e) "input distribution"): "replicate" = every rank filters the frame and derives the normal map
        # itself (no traffic); "broadcast" = rank 0 does, and its filtered depth + normal map (20 B per pixel) are broadcast
        assert inputs in ("replicate", "broadcast")
        self.inputs = inputs
        # images (composite mode): "all" = every rank ends up with the merged images (the second collective is an all-reduce);
        # "root" = only rank 0 does (a reduce: half the traffic of the payload's all-reduce) -- a known-pose stream has one
        # consumer of the rendering, and the tracking loop can solve on rank 0 and broadcast the 4 x 4 pose (TrackingSlabPipeline)
        assert images in ("all", "root")
        if images == "root" and raycast != "composite":
            raise ValueError("SlabPipeline: images='root' is an option of the composite raycast")
        self.images = images
        # merge (composite mode): how the per-slab images become one.  "direct" = every rank owns one strip of the image: one
        # all-to-all sends each strip to its owner, the owner keeps the nearest hit per pixel, one all-gather (images = "root": a
        # gather) returns the merged strips -- each byte crosses one link once per phase, all links of the xGMI mesh at once;
        # "allreduce" = a MIN all-reduce of (depth, rank) keys + a SUM all-reduce of the winners' payload (whole images through
        # whatever ring / tree the library builds).  Same winner per pixel, same images.
        assert merge in ("direct", "allreduce")
        self.merge = merge
        # overlap (composite mode, known-pose streams): the merge of frame k's per-slab images -- two latency-bound
        # all-reduces and three small kernels -- runs on a second stream while the main stream already preprocesses and
        # integrates frame k + 1; step() then returns before ray_d / ray_n / ray_i are merged: they are valid only after
        # wait_composite() (bench.py's sync_all calls it; the next raycast_into waits for it by itself).
        self.overlap = bool(overlap)
        # The overlapped merge issues its all-reduces from a side stream while the main stream may issue the ghost-plane
        # send / recv of the next SdfFuse: two NCCL call sequences whose relative order can differ between ranks (and torch may
        # route point-to-point through a communicator of its own) -- the classic collective-ordering deadlock.  Not allowed.
        if self.overlap and raycast == "composite" and (halo == "exchange" or inputs == "broadcast"):
            raise ValueError("SlabPipeline: overlap=True needs halo='recompute' and inputs='replicate' (an overlapped merge next to the "
                             "ghost-plane exchange or the input broadcast would interleave collectives in rank-dependent order)")
        # overlap with raycast = "exact" (driver "c"): frames pipelined across the ranks -- frame k's final exchange of the finalised
        # pixels runs on the frame object's side stream through a second communicator while the main stream carries frame k + 1
        # (include/kfx_slab.h); `pipeline` image sets are in flight, step() returns before the rendering exists, and
        # wait_composite() makes ray_d / ray_n / ray_i the last frame's images.  Same bits as without overlap.
        if self.overlap and raycast != "composite" and (driver != "c" or raycast != "exact"):
            raise ValueError("SlabPipeline: overlap=True with the exact raycast (pipelined frames) is a feature of driver='c'")
        self.pipeline = max(2, min(int(pipeline), 4))
        if kw.get("track"):
            raise ValueError("SlabPipeline: track=True (brick summary) is a single-volume feature; slabs march without it")
        self._side = self._merged = None
        self.halo = halo
        self.raycast_mode = raycast
        self.kind = kind            # "f32": SDF_t cells; "f16": SDF_h cells (config C5: 2048^3 over 8 GPUs = 4 GiB per rank)
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.full_boxmin = np.asarray(boxmin, np.float32)
        self.full_boxmax = np.asarray(boxmax, np.float32)
        d = int(dims[2])
        if ghost is None:
            ghost = self.GHOST
            if driver == "c" and raycast == "exact" and halo == "recompute" and self.world > 1 and kind == "f32":
                from . import slab as S
                Kk = kw.get("K")
                Kk = scenes.intrinsics(w, h) if Kk is None else Kk
                wide = S.exact_ghost(dims, boxmin, boxmax, scenes.trunc_dist(boxmin, boxmax, dims, kw.get("trunc_factor", scenes.TRUNC_DIST_FACTOR)), Kk, w, h)
                if d // self.world >= wide:
                    ghost = wide
        self.GHOST = int(ghost)   # (shadows the class default for this pipeline)
        # exchange_halos takes the ghost planes from the immediate neighbours: every rank must own at least GHOST planes
        assert self.world == 1 or d // self.world >= self.GHOST, "slabs thinner than the ghost width (%d planes / %d ranks)" % (d, self.world)
        self.z0, self.z1 = slab_range(d, self.rank, self.world)
        self.s0, self.s1 = max(self.z0 - self.GHOST, 0), min(self.z1 + self.GHOST, d)
        kw["contiguous_images"] = True  # collectives operate on dense tensors
        assert driver in ("python", "c")
        self.driver, self.sframe, self.comm = driver, None, comm
        timing_slots = kw.pop("timing_slots", None)
        super().__init__(ops, dims, boxmin, boxmax, w, h, **kw)
        if driver == "c":
            if raycast == "exact_allreduce" or images != "all" or kind != "f32":
                raise ValueError("SlabPipeline(driver='c'): raycast 'exact' or 'composite', images='all', fp32 cells")
            from . import slab as S
            if self.comm is None:
                self.comm = S.Comm.torch(dist)
            lay = S.layout(d, float(self.full_boxmin[2]), float(self.full_boxmax[2]), self.rank, self.world, self.GHOST)
            assert (lay.z0, lay.z1, lay.s0, lay.s1) == (self.z0, self.z1, self.s0, self.s1)
            # the image sets 1 .. of the pipelined exact raycast (set 0: ray_d / ray_n / ray_i), there from the start so that
            # configure(overlap=True) can switch to pipelined frames on the same object (bench.py times the variants on one pipeline)
            I = ops.Image
            extra = [(I(w, h, "f32", pitch=w * 4), I(w, h, "f32x4", pitch=w * 16), I(w, h, "f32", pitch=w * 4)) for _ in range(self.pipeline - 1)] if self.world > 1 else []
            self.sframe = S.SlabFrame(self.comm, self.vol, lay, self.raw, self.filtered, self.vbo, self.normals, self.ray_d, self.ray_n, self.ray_i, self.K,
                                      self.bil, self.near, self.far, self.trunc, self.max_w, self.mincostheta, halo=halo, raycast=raycast, merge=merge,
                                      inputs=inputs, overlap=overlap, tiles=tiles, unchecked=unchecked, timing_slots=int(timing_slots or 256),
                                      pipe_images=extra)

    def configure(self, **kw):
        """Change policies between frames (halo, raycast, merge, inputs, overlap, tiles): bench.py times the variants on one pipeline."""
        names = {"halo": "halo", "raycast": "raycast_mode", "merge": "merge", "inputs": "inputs", "overlap": "overlap"}
        self.wait_composite()
        if self.sframe is not None:
            self.sframe.configure(**kw)
            self.ray_d, self.ray_n, self.ray_i = self.sframe.image_sets[0]   # (pipelined frames re-point these at the last frame's set)
        for k, v in kw.items():
            if k in names:
                setattr(self, names[k], v)

    def step(self, T_wc, raw_image=None):
        if self.sframe is not None:
            self.sframe.step(T_wc, scenes.se3_inverse(T_wc), raw_image)
            self.frames_done += 1
            return
        super().step(T_wc, raw_image)

    def _alloc_volume(self, boxmin, boxmax):
        W, H, D = self.dims
        f = np.float32
        size_z = f(boxmax[2]) - f(boxmin[2])
        # plane positions exactly as VoxelPositionInUnits computes them (BoundedVolume.h:115-125)
        zlo = f(boxmin[2]) + size_z * f(self.s0) / f(D - 1)
        zhi = f(boxmin[2]) + size_z * f(self.s1 - 1) / f(D - 1)
        lo = np.array([boxmin[0], boxmin[1], zlo], f)
        hi = np.array([boxmax[0], boxmax[1], zhi], f)
        if self.kind != "f32":
            return self.ops.BoundedVolume(W, H, self.s1 - self.s0, lo, hi, kind=self.kind)
        return self.ops.BoundedVolume(W, H, self.s1 - self.s0, lo, hi)

    def preprocess(self, raw_image=None):
        if self.sframe is not None:
            self.sframe.step(_IDENTITY, None, raw_image, 1)   # KFX_FRAME_PREPROCESS (the broadcast included)
            return
        if self.inputs == "broadcast" and self.world > 1:
            if self.rank == 0:
                super().preprocess(raw_image)
            self.dist.broadcast(self.filtered.tensor(), src=0)
            self.dist.broadcast(self.normals.tensor(), src=0)
        else:
            super().preprocess(raw_image)

    def fuse(self, T_wc, T_cw=None):
        """T_cw: world -> camera transform to use instead of the float32 inverse of T_wc (the tracking loop inverts
        its pose in float64, main.cpp:345-356)."""
        if self.sframe is not None:
            self.sframe.step(T_wc, scenes.se3_inverse(T_wc) if T_cw is None else T_cw, None, 2)   # KFX_FRAME_FUSE (+ ghost planes)
            return
        # a slab's plane count is usually not a multiple of 8: full_extent="slab" integrates exactly the voxels roo::SdfFuse
        # integrates on the whole volume (x / y extents (dim/8)*8, planes below (D/8)*8), whatever the partition
        D = self.dims[2]
        zmin, zmax = float(self.full_boxmin[2]), float(self.full_boxmax[2])
        if self.halo == "recompute" or self.world == 1:
            target, first = self.vol, self.s0
        else:
            target, first = self.vol.ZSlab(self.z0 - self.s0, self.z1 - self.s0), self.z0  # owned planes only
        # slab entry point: voxel positions by the FULL volume's expression, so every plane is integrated
        # bit-identically to the same plane of a single-GPU volume
        self.ops.SdfFuse(target, self.filtered, self.normals, scenes.se3_inverse(T_wc) if T_cw is None else T_cw, self.K, self.trunc,
                         self.max_w, self.mincostheta, full_extent="slab", slab=(D, first, zmin, zmax))
        if target is not self.vol:
            self.exchange_halos()

    def _p2p(self, ops):
        """One batched group of point-to-point operations, complete on return as far as the current stream is concerned.
        RCCL (backend "nccl") is stream-ordered: the requests' wait() chains the current stream behind the transfers.  gloo
        with device tensors -- ranks sharing one GPU in the tests -- goes through host copies made here, with the device
        synchronised on both sides of the batch."""
        if not ops:
            return
        dist = self.dist
        if ops[0].tensor.is_cuda and dist.get_backend() != "nccl":
            # gloo's own path for device tensors takes ~0.1 s per message: stage through host copies here (test transport only)
            import torch
            torch.cuda.synchronize()
            host = [(op, op.tensor.cpu() if op.op is dist.isend else torch.empty(op.tensor.shape, dtype=op.tensor.dtype)) for op in ops]
            for req in dist.batch_isend_irecv([dist.P2POp(op.op, t, op.peer) for op, t in host]):
                req.wait()
            for op, t in host:
                if op.op is not dist.isend:
                    op.tensor.copy_(t)
            torch.cuda.synchronize()
            return
        for req in dist.batch_isend_irecv(ops):
            req.wait()

    def exchange_halos(self):
        """Refresh the ghost planes from the neighbours' owned planes (contiguous img_pitch-sized
        slices): rank r sends its first / last G owned planes down / up and receives the matching
        planes of its neighbours.  One batched group of point-to-point operations."""
        dist = self.dist
        ops = []
        lo_ghost = self.z0 - self.s0              # planes [s0, z0) come from rank-1
        hi_ghost = self.s1 - self.z1              # planes [z1, s1) come from rank+1
        own0, own1 = self.z0 - self.s0, self.z1 - self.s0   # local indices of the owned range
        if self.rank > 0 and lo_ghost > 0:
            ops.append(dist.P2POp(dist.isend, self.vol.planes(own0, own0 + lo_ghost), self.rank - 1))
            ops.append(dist.P2POp(dist.irecv, self.vol.planes(0, lo_ghost), self.rank - 1))
        if self.rank < self.world - 1 and hi_ghost > 0:
            ops.append(dist.P2POp(dist.isend, self.vol.planes(own1 - hi_ghost, own1), self.rank + 1))
            ops.append(dist.P2POp(dist.irecv, self.vol.planes(own1, own1 + hi_ghost), self.rank + 1))
        self._p2p(ops)

    def raycast(self, T_wc):
        """raycast = "composite": every rank marches its own slab from the slab's entry point and the nearest
        hit wins (one MIN + one SUM all-reduce; rays re-enter each slab with a fresh step, so depths can differ
        from the single-volume march in the last bits).  raycast = "exact": the march state travels with the
        ray from slab to slab (raycast_exact), bit-identical to RaycastSdf on the whole volume."""
        if self.sframe is not None:
            self.sframe.step(T_wc, None, None, 4)   # KFX_FRAME_RAYCAST (+ the hand-over / the merge)
            return
        self.raycast_into(self.ray_d, self.ray_n, self.ray_i, self.K, T_wc)

    def raycast_into(self, d, n, i, K, T_wc):
        """The model rendered from T_wc with intrinsics K into the images d / n / i (any size: pyramid levels),
        identical on every rank afterwards."""
        if self.raycast_mode == "exact":
            self.raycast_exact(T_wc, d, n, i, K)
            return
        if self.raycast_mode == "exact_allreduce":
            self.raycast_exact_allreduce(T_wc, d, n, i, K)
            return
        self.wait_composite()   # the previous frame's merge still reads these images
        self.ops.RaycastSdf(d, n, i, self.vol, T_wc, K, self.near, self.far, self.trunc, True)
        if self.world > 1:
            if self.overlap and (self.halo == "exchange" or self.inputs == "broadcast"):
                raise RuntimeError("SlabPipeline: overlapped merge with halo='exchange' or inputs='broadcast' (see __init__)")
            if self.overlap and hasattr(self.ops, "CompositePack"):
                import torch
                if self._side is None:
                    self._side = torch.cuda.Stream()
                marched = torch.cuda.Event()
                marched.record()
                with torch.cuda.stream(self._side):
                    self._side.wait_event(marched)
                    self.composite(d, n, i)
                    self._merged = torch.cuda.Event()
                    self._merged.record(self._side)
            else:
                self.composite(d, n, i)

    def wait_composite(self):
        """Make the current stream wait for an overlapped merge (no-op otherwise)."""
        if self.sframe is not None:
            self.sframe.wait()
            if self.overlap and self.raycast_mode == "exact" and self.world > 1 and self.sframe.count > 0:
                self.ray_d, self.ray_n, self.ray_i = self.sframe.images()   # the set the last frame was rendered into
            return
        if self._merged is not None:
            import torch
            torch.cuda.current_stream().wait_event(self._merged)
            self._merged = None

    def raycast_levels_into(self, outputs, K_levels, T_wc):
        """Several renderings of the model from one pose (the tracking loop's pyramid levels): outputs = [(d, n, i), ...].
        Composite mode renders them with one launch where the operator set has RaycastSdfLevels and merges them with ONE
        pair of collectives (the strips -- or keys and payloads -- of all levels side by side) instead of a pair per level; the
        images equal those of per-level raycast_into calls.  Exact mode hands the march over level by level."""
        if self.raycast_mode == "exact":
            for (d, n, i), K in zip(outputs, K_levels):
                self.raycast_exact(T_wc, d, n, i, K)
            return
        if hasattr(self.ops, "RaycastSdfLevels") and len(outputs) > 1:
            self.ops.RaycastSdfLevels(outputs, self.vol, T_wc, K_levels, self.near, self.far, self.trunc, True)
        else:
            for (d, n, i), K in zip(outputs, K_levels):
                self.ops.RaycastSdf(d, n, i, self.vol, T_wc, K, self.near, self.far, self.trunc, True)
        if self.world == 1:
            return
        if self.merge == "direct" and hasattr(self.ops, "CompositeStripsPack"):
            self.composite_direct_levels(outputs)
            return
        if not hasattr(self.ops, "CompositePack"):
            for d, n, i in outputs:
                self.composite(d, n, i)
            return
        import torch
        sizes = [d.w * d.h for d, _, _ in outputs]
        total = sum(sizes)
        key = self._scratch("keys", (total,), torch.int64, outputs[0][0])
        payload = self._scratch("payloads", (total * 4,), torch.float32, outputs[0][0])
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
        parts = [(key[offs[k]:offs[k + 1]], payload[4 * offs[k]:4 * offs[k + 1]]) for k in range(len(outputs))]
        for (d, n, i), (kk, _) in zip(outputs, parts):
            self.ops.CompositePack(d, n, i, kk, self.rank)
        self.dist.all_reduce(key, op=self.dist.ReduceOp.MIN)
        for (d, n, i), (kk, pp) in zip(outputs, parts):
            self.ops.CompositeSelect(d, n, i, kk, pp, self.rank)
        self._sum_payload(payload)
        for (d, n, i), (kk, pp) in zip(outputs, parts):
            self.ops.CompositeUnpack(d, n, i, kk, pp)

    def _sum_payload(self, payload):
        """The winners' normals / shade, summed over the ranks: on every rank, or (images = "root") on rank 0 only."""
        if self.images == "root":
            self.dist.reduce(payload, dst=0, op=self.dist.ReduceOp.SUM)
        else:
            self.dist.all_reduce(payload, op=self.dist.ReduceOp.SUM)

    def _scratch(self, name, shape, dtype, like):
        """Per-shape device scratch tensors (march state, composite key / payload), allocated once."""
        import torch
        cache = self.__dict__.setdefault("_scratch_cache", {})
        key = (name, tuple(shape))
        if key not in cache:
            cache[key] = torch.empty(shape, dtype=dtype, device=like.tensor().device)
        return cache[key]

    def raycast_exact(self, T_wc, d=None, n=None, i=None, K=None):
        """The march state travels with the ray (the exact variant): world + 1 stages of kfx_raycast_sdf_slab
        -- a rank advances the rays whose current sample lies in the planes it owns --, between them the march planes
        (lambda, last_sdf, delta, status, touched) go to the two NEIGHBOUR ranks only, and a rank adopts the rays a
        neighbour advanced that are still under way (status 0 marching, 3 hit with its normal pending).  A stale copy is
        never acted on: its position lies in planes of a rank the ray has left.  A ray entering at one end needs `world`
        stages, one more lets a hit on a slab boundary get its normal from the neighbour owning the gradient's base plane.
        At the end ONE all-reduce sums the results of the ranks that finalised each pixel (integer bit patterns; rank 0
        answers for rays that never enter the box).  No host synchronisation between the stages.  Depth, normals and
        shade equal RaycastSdf on the whole volume bit for bit."""
        import torch
        dist, o = self.dist, self.ops
        d, n, i = (self.ray_d, self.ray_n, self.ray_i) if d is None else (d, n, i)
        K = self.K if K is None else K
        w, h = d.w, d.h
        D = self.dims[2]
        slab = (D, self.s0, float(self.full_boxmin[2]), float(self.full_boxmax[2]))
        st = self._scratch("state", (9, h, w), torch.float32, d)
        march = st[0:5].view(torch.int32)   # contiguous planes 0..4
        stages = self.world + 1 if self.world > 1 else 1
        fin = torch.zeros((h, w), dtype=torch.bool, device=st.device)
        from_lo = self._scratch("from_lo", (5, h, w), torch.int32, d) if self.rank > 0 else None
        from_hi = self._scratch("from_hi", (5, h, w), torch.int32, d) if self.rank < self.world - 1 else None
        for stage in range(stages):
            o.RaycastSdfSlab(st, stage == 0, self.vol, slab, self.z0, self.z1, w, h, T_wc, K, self.near, self.far, self.trunc, True)
            if self.world == 1:
                break
            status, touched = st[3], march[4] != 0
            fin |= touched & ((status == 1) | (status == 2))
            if stage == 0 and self.rank == 0:
                fin |= ~touched & (status == 2)
            if stage + 1 == stages:
                break
            ops = []
            if from_lo is not None:
                ops += [dist.P2POp(dist.isend, march, self.rank - 1), dist.P2POp(dist.irecv, from_lo, self.rank - 1)]
            if from_hi is not None:
                ops += [dist.P2POp(dist.isend, march, self.rank + 1), dist.P2POp(dist.irecv, from_hi, self.rank + 1)]
            self._p2p(ops)
            taken = torch.zeros_like(fin)
            for buf in (from_lo, from_hi):
                if buf is None:
                    continue
                bstat = buf[3].view(torch.float32)
                take = (buf[4] != 0) & ((bstat == 0) | (bstat == 3)) & ~taken
                march[0:4].copy_(torch.where(take.unsqueeze(0), buf[0:4], march[0:4]))
                taken |= take
        if self.world > 1:
            allst = st.view(torch.int32)
            contrib = torch.stack([allst[0], allst[3], allst[5], allst[6], allst[7], allst[8]])
            contrib = torch.where(fin.unsqueeze(0), contrib, torch.zeros_like(contrib)).contiguous()
            dist.all_reduce(contrib, op=dist.ReduceOp.SUM)
            for k, p in enumerate((0, 3, 5, 6, 7, 8)):
                allst[p].copy_(contrib[k])
            final = (st[3] == 1) | (st[3] == 2)
            assert bool(final.all()), "slab march: %d rays without a final status after world + 1 stages" % int((~final).sum())
        self.rounds = stages
        o.RaycastStateToImages(d, n, i, st)

    def raycast_exact_allreduce(self, T_wc, d=None, n=None, i=None, K=None):
        """The cross-check of raycast_exact: the same kernels with the state merged over ALL ranks after every round.  In a
        round exactly one rank advances a given ray (the owner of the trilinear base plane of its current sample), so the
        merge is one integer SUM all-reduce of the touched pixels' march planes; a host-side termination test follows every
        round (at most world + 2 of them).  The normals (planes 5-8, written by one rank per pixel) are merged once at the end."""
        import torch
        dist, o = self.dist, self.ops
        d, n, i = (self.ray_d, self.ray_n, self.ray_i) if d is None else (d, n, i)
        K = self.K if K is None else K
        w, h = d.w, d.h
        D = self.dims[2]
        slab = (D, self.s0, float(self.full_boxmin[2]), float(self.full_boxmax[2]))
        st = self._scratch("state", (9, h, w), torch.float32, d)
        march = st[0:5].view(torch.int32)   # contiguous planes 0..4
        rounds = 0
        while True:
            o.RaycastSdfSlab(st, rounds == 0, self.vol, slab, self.z0, self.z1, w, h, T_wc, K,
                             self.near, self.far, self.trunc, True)
            rounds += 1
            if self.world > 1:
                touched = march[4] != 0
                contrib = torch.where(touched.unsqueeze(0), march, torch.zeros_like(march))
                dist.all_reduce(contrib, op=dist.ReduceOp.SUM)
                march.copy_(torch.where((contrib[4] != 0).unsqueeze(0), contrib, march))
            status = st[3]
            if not bool(((status == 0) | (status == 3)).any()):
                break
            assert rounds <= self.world + 3, "slab march did not terminate"
        if self.world > 1:
            out = st[5:9].view(torch.int32)
            dist.all_reduce(out, op=dist.ReduceOp.SUM)
        self.rounds = rounds
        o.RaycastStateToImages(d, n, i, st)

    def _all_to_all(self, recv, send):
        """recv[r] = rank r's send[self.rank] (dense tensors of shape (world, ...))."""
        dist = self.dist
        if dist.get_backend() == "nccl":
            dist.all_to_all_single(recv, send)   # RCCL: one grouped send / recv per peer, every link of the mesh at once
            return
        if send.is_cuda:   # gloo with device tensors (the tests' ranks sharing one GPU): its point-to-point path takes ~0.1 s per
            import torch   # message, its all-reduce does not -- every rank's strips summed as integers into one table (bits survive)
            table = torch.zeros((self.world,) + tuple(send.shape), dtype=torch.int32, device=send.device)
            table[self.rank] = send.view(torch.int32)
            dist.all_reduce(table)
            recv.copy_(table[:, self.rank].view(torch.float32))
            return
        recv[self.rank].copy_(send[self.rank])
        ops = []
        for k in range(1, self.world):   # gloo, host tensors (the CPU tests)
            to, frm = (self.rank + k) % self.world, (self.rank - k) % self.world
            ops += [dist.P2POp(dist.isend, send[to], to), dist.P2POp(dist.irecv, recv[frm], frm)]
        self._p2p(ops)

    def _gather_strips(self, full, part):
        """full[r] = rank r's part: on every rank, or (images = "root") on rank 0 only."""
        dist = self.dist
        if dist.get_backend() == "nccl":
            if self.images == "root":
                dist.gather(part, [full[r] for r in range(self.world)] if self.rank == 0 else None, dst=0)
            else:
                dist.all_gather_into_tensor(full, part)
            return
        if part.is_cuda:   # (gloo with device tensors: as in _all_to_all; every rank ends up with the strips)
            import torch
            table = torch.zeros(tuple(full.shape), dtype=torch.int32, device=part.device)
            table[self.rank] = part.view(torch.int32).reshape(table[self.rank].shape)
            dist.all_reduce(table)
            full.copy_(table.view(torch.float32))
            return
        full[self.rank].copy_(part)
        ops = []
        for k in range(1, self.world):
            to, frm = (self.rank + k) % self.world, (self.rank - k) % self.world
            if self.images != "root" or to == 0:
                ops.append(dist.P2POp(dist.isend, part, to))
            if self.images != "root" or self.rank == 0:
                ops.append(dist.P2POp(dist.irecv, full[frm], frm))
        self._p2p(ops)

    def composite_direct_levels(self, outputs):
        """The direct-send merge of several images at once (pyramid levels: outputs = [(d, n, i), ...]): a rank's strips of all
        levels lie side by side in one buffer, so the whole set costs ONE all-to-all and ONE all-gather."""
        import torch
        o, W = self.ops, self.world
        planes = 5
        S = [o.CompositeStripPixels(d.w, d.h, W) for d, _, _ in outputs]
        offs = [0]
        for s_l in S:
            offs.append(offs[-1] + planes * s_l)
        total = offs[-1]
        like = outputs[0][0]
        send = self._scratch("strips_send", (W, total), torch.float32, like)
        recv = self._scratch("strips_recv", (W, total), torch.float32, like)
        merged = self._scratch("strips_merged", (total,), torch.float32, like)
        for (d, n, i), off in zip(outputs, offs):
            o.CompositeStripsPack(d, n, i, send, W, offset=off, rank_stride=total)
        self._all_to_all(recv, send)
        for s_l, off in zip(S, offs):
            o.CompositeStripsMerge(recv, merged, s_l, W, offset=off, rank_stride=total, merged_offset=off)
        self._gather_strips(recv, merged)      # (recv is free again: it becomes the gathered strips)
        if self.images != "root" or self.rank == 0:
            for (d, n, i), off in zip(outputs, offs):
                o.CompositeStripsUnpack(d, n, i, recv, W, offset=off, rank_stride=total)

    def composite_direct(self, d, n, i):
        """The direct-send merge (include/kfx.h, kfx_composite_strips_*): strips to their owners, nearest hit per pixel, strips
        back.  Operator sets without the kernels (the oracle-backed CPU stand-in of the tests) use the tensor expressions."""
        if hasattr(self.ops, "CompositeStripsPack"):
            self.composite_direct_levels([(d, n, i)])
            return
        import torch
        W = self.world
        w, h = d.w, d.h
        planes = 5
        dt, nt, it = d.tensor(), n.tensor(), i.tensor()
        P = w * h
        S = ((P + W - 1) // W + 63) // 64 * 64   # kfx_composite_strip_pixels
        hit = torch.isfinite(dt)
        img = torch.zeros((planes, W * S), dtype=torch.float32, device=dt.device)
        img[0] = float("inf")
        img[0, :P] = torch.where(hit, dt, torch.full_like(dt, float("inf"))).reshape(-1)
        for c in range(3):
            img[1 + c, :P] = torch.where(hit, nt[..., c], torch.zeros_like(dt)).reshape(-1)
        img[4, :P] = torch.where(hit, it, torch.zeros_like(it)).reshape(-1)
        send = img.reshape(planes, W, S).permute(1, 0, 2).contiguous()
        recv = torch.empty_like(send)
        self._all_to_all(recv, send)
        bits = recv[:, 0].contiguous().view(torch.int32).to(torch.int64)
        win = torch.argmin((bits << 8) | torch.arange(W, device=dt.device).reshape(W, 1), dim=0)
        merged = torch.gather(recv, 0, win.reshape(1, 1, S).expand(1, planes, S))[0].contiguous()
        full = torch.empty_like(send)
        self._gather_strips(full, merged)
        if self.images == "root" and self.rank != 0:
            return
        out = full.permute(1, 0, 2).reshape(planes, W * S)[:, :P]
        any_hit = torch.isfinite(out[0]).reshape(h, w)
        dt.copy_(torch.where(any_hit, out[0].reshape(h, w), torch.full_like(dt, float("nan"))))
        for c in range(3):
            nt[..., c].copy_(out[1 + c].reshape(h, w))
        nt[..., 3].copy_(any_hit.to(torch.float32))
        it.copy_(out[4].reshape(h, w))

    def composite(self, d=None, n=None, i=None):
        """Nearest hit over all slabs (merge = "direct": composite_direct above).  key = depth bits (positive floats order like ints) in the
        high word, rank in the low byte; misses use +inf.  One MIN all-reduce picks the winner,
        one SUM all-reduce broadcasts its normal / shade (four floats per pixel; the depth and the hit flag travel in the key).  With the HIP
        operator set the per-pixel glue is three fused kernels; operator sets without them (the
        oracle-backed CPU stand-in of the tests) use the equivalent tensor expressions below."""
        import torch
        dist = self.dist
        d, n, i = (self.ray_d, self.ray_n, self.ray_i) if d is None else (d, n, i)
        w, h = d.w, d.h
        if self.merge == "direct":
            self.composite_direct(d, n, i)
            return
        if hasattr(self.ops, "CompositePack"):
            key = self._scratch("key", (w * h,), torch.int64, d)
            payload = self._scratch("payload", (w * h * 4,), torch.float32, d)
            self.ops.CompositePack(d, n, i, key, self.rank)
            dist.all_reduce(key, op=dist.ReduceOp.MIN)
            self.ops.CompositeSelect(d, n, i, key, payload, self.rank)
            self._sum_payload(payload)
            self.ops.CompositeUnpack(d, n, i, key, payload)   # (images = "root": only rank 0's payload is the sum)
            return
        dt, nt, it = d.tensor(), n.tensor(), i.tensor()
        hit = torch.isfinite(dt)
        bits = torch.where(hit, dt, torch.full_like(dt, float("inf"))).contiguous().view(torch.int32).to(torch.int64)
        key = (bits << 8) | self.rank
        dist.all_reduce(key, op=dist.ReduceOp.MIN)
        mine = hit & ((key & 0xFF) == self.rank) & ((key >> 8) == bits)
        payload = torch.zeros((h, w, 4), dtype=torch.float32, device=dt.device)   # {n.x, n.y, n.z, shade}; n.w = hit ? 1 : 0 comes out of the key
        payload[..., 0:3] = torch.where(mine.unsqueeze(-1), nt[..., 0:3], torch.zeros_like(nt[..., 0:3]))
        payload[..., 3] = torch.where(mine, it, torch.zeros_like(it))
        self._sum_payload(payload)
        win_bits = (key >> 8).to(torch.int32)
        any_hit = win_bits < 0x7F800000
        dt.copy_(torch.where(any_hit, win_bits.view(torch.float32), torch.full_like(dt, float("nan"))))
        nt[..., 0:3].copy_(payload[..., 0:3])
        nt[..., 3].copy_(any_hit.to(torch.float32))
        it.copy_(payload[..., 3])


class TrackingSlabPipeline(SlabPipeline):
    """Tracked KinectFusion on Z-slabs: the loop of TrackingPipeline with the model spread over the ranks.  The
    raycast of every pyramid level is merged across ranks (raycast_into), after which all ranks hold the same model
    images; the ICP normal equations are image-space work of < 0.1 ms and are evaluated redundantly on every rank
    (deterministic: every rank derives the same pose without a broadcast); SdfFuse then integrates each rank's slab.
    With raycast="exact" the poses and the fused slabs equal the single-GPU TrackingPipeline bit for bit."""

    LEVELS = TrackingPipeline.LEVELS

    def __init__(self, ops, dist, dims, boxmin, boxmax, w, h, its=None, icp_c=0.1, max_rmse=0.10, **kw):
        from . import tracking
        if kw.get("inputs", "replicate") != "replicate":
            raise ValueError("TrackingSlabPipeline: inputs='broadcast' is not implemented (every rank builds the frame's pyramids itself)")
        if kw.get("overlap"):
            raise ValueError("TrackingSlabPipeline: overlap=True is for known-pose streams (the tracker reads the merged images at once)")
        if kw.get("driver", "python") != "python":
            # the C frame object is bound to SlabPipeline's filtered / normals images at construction; this loop integrates the
            # pyramids' level 0 (kin_d[0] / kin_n[0]) and renders pyramid levels, which kfx_slab_frame_step knows nothing about
            raise ValueError("TrackingSlabPipeline: driver='c' is for known-pose streams (kfx_slab_frame_step integrates the images it was created "
                             "with, not the tracker's pyramids); use driver='python'")
        super().__init__(ops, dist, dims, boxmin, boxmax, w, h, **kw)
        self.tracking = tracking
        self.its = tuple(tracking.DEFAULT_ITS if its is None else its)
        self.icp_c, self.max_rmse = float(icp_c), float(max_rmse)
        L, P = self.LEVELS, ops.Pyramid
        self.kin_d, self.kin_v, self.kin_n = P(w, h, L, "f32"), P(w, h, L, "f32x4"), P(w, h, L, "f32x4")
        self.pyr_d, self.pyr_i = P(w, h, L, "f32"), P(w, h, L, "f32")
        self.pyr_n, self.pyr_v = P(w, h, L, "f32x4"), P(w, h, L, "f32x4")
        self.K_levels = [scenes.intrinsics_level(self.K, l) for l in range(L)]
        self.debug = ops.Image(w, h, "f32x4")
        self.scratch = ops.Image(w * 232, h, "u8")
        self.T_wl = np.eye(4)
        self.frame = 0
        self.rmse, self.tracking_good = 0.0, True
        self.resets = 0

    preprocess = TrackingPipeline.preprocess

    def step(self, T_wl_init=None, raw_image=None):
        o, tr = self.ops, self.tracking
        self.preprocess(raw_image)
        recover = self.frame > 0 and not np.isfinite(self.rmse)   # main.cpp:223-242, as TrackingPipeline.step (every rank sees the same rmse)
        if recover:
            self.T_wl = np.eye(4)
            self.ops.SdfReset(self.vol, float("nan"))
            self.rmse, self.tracking_good = 0.0, True
            self.resets += 1
        if self.frame == 0 or recover:
            if T_wl_init is not None:
                self.T_wl = np.vstack([np.asarray(T_wl_init, np.float64).reshape(3, 4), [0, 0, 0, 1]])
            self._fuse_at(self.T_wl)
        if self.frame > 0:
            T34 = self.T_wl[:3].astype(np.float32)
            lv = [l for l in range(self.LEVELS) if self.its[l] > 0]
            self.raycast_levels_into([(self.pyr_d[l], self.pyr_n[l], self.pyr_i[l]) for l in lv], [self.K_levels[l] for l in lv], T34)
            for l in lv:
                o.DepthToVbo(self.pyr_v[l], self.pyr_d[l], self.K_levels[l])
            if self.images == "root" and self.world > 1:
                # only rank 0 holds the merged model images: it solves, the others receive the pose (and the verdict)
                import torch
                msg = torch.zeros(18, dtype=torch.float64)
                if self.rank == 0:
                    T_lp, self.rmse, self.tracking_good = tr.refine_pose(o, self.kin_v, self.pyr_v, self.pyr_n, self.K_levels,
                                                                         self.scratch, self.debug, self.its, self.icp_c, self.max_rmse)
                    msg[:16] = torch.from_numpy(np.asarray(T_lp, np.float64).reshape(16))
                    msg[16], msg[17] = float(self.rmse), 1.0 if self.tracking_good else 0.0
                if self.dist.get_backend() == "nccl":
                    msg = msg.to(self.pyr_d[0].tensor().device)
                self.dist.broadcast(msg, src=0)
                msg = msg.cpu()
                T_lp = msg[:16].numpy().reshape(4, 4).copy()
                self.rmse, self.tracking_good = float(msg[16]), bool(msg[17] != 0)
            else:
                T_lp, self.rmse, self.tracking_good = tr.refine_pose(o, self.kin_v, self.pyr_v, self.pyr_n, self.K_levels,
                                                                     self.scratch, self.debug, self.its, self.icp_c, self.max_rmse)
            if self.tracking_good:
                if hasattr(o, "PoseStep"):   # the same host arithmetic as TrackingPipeline.step (kfx_pose_step): the same poses, bit for bit
                    self.T_wl, T_cw = o.PoseStep(self.T_wl, T_lp)
                    self._fuse_at(self.T_wl, T_cw)
                else:
                    self.T_wl = self.T_wl @ tr.se3_inv(T_lp)
                    self._fuse_at(self.T_wl)
        self.frame += 1
        return self.T_wl

    def _fuse_at(self, T_wl, T_cw=None):
        # SlabPipeline.fuse integrates self.filtered / self.normals at T_wc; point them at pyramid level 0
        self.filtered, self.normals = self.kin_d[0], self.kin_n[0]
        self.fuse(T_wl[:3].astype(np.float32), T_cw=self.tracking.se3_inv(T_wl)[:3].astype(np.float32) if T_cw is None else T_cw)
