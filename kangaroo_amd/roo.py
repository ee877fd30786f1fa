"""Python mirror of the reference's `roo::` interface for the KinectFusion hot path.

Same names and argument order as the reference's C++ free functions
(include/kangaroo/cu_sdffusion.h:13-26, cu_raycast.h:13-14, cu_bilateral.h:9-19,
cu_normals.h:9-10, cu_depth_tools.h:19-21), same container semantics (pitched, non-owning
views passed by value; `Image`/`BoundedVolume` here own torch storage the way the
reference's `...,TargetDevice,Manage>` objects own cudaMallocPitch memory).  Every op
calls the HIP kernels through the C ABI in libkfx.so -- there is no other execution path.

torch is used only for device memory and stream plumbing.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import KfxError, KfxImage, KfxVolume  # noqa: F401

_ELEM = {
    "f32": (np.float32, 1), "f32x4": (np.float32, 4), "u16": (np.uint16, 1), "u8": (np.uint8, 1),
    "u8x3": (np.uint8, 3),   # Image<uchar3>: the RGB frame of the colour path
    "u8x4": (np.uint8, 4),   # Image<uchar4>: ColourVbo output
}
PITCH_ALIGN = 256  # same policy as kfx_alloc_pitched


def _stream(stream):
    if stream is not None:
        return C.c_void_p(int(stream))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _fp(a, n):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))
    if a.size != n:
        raise ValueError("expected %d floats, got %d" % (n, a.size))
    return a.ctypes.data_as(_lib.PF), a


class Image:
    """roo::Image<T, TargetDevice, Manage>: pitched 2-D device image (Image.h:43-621)."""

    def __init__(self, w, h, kind="f32", device="cuda", pitch=None, _storage=None, _offset=0):
        self.w, self.h, self.kind = int(w), int(h), kind
        self.np_dtype, self.channels = _ELEM[kind]
        self.elem = np.dtype(self.np_dtype).itemsize * self.channels
        if _storage is None:
            self.pitch = int(pitch) if pitch else (self.w * self.elem + PITCH_ALIGN - 1) // PITCH_ALIGN * PITCH_ALIGN
            self.storage = torch.zeros(max(self.pitch * self.h, 1), dtype=torch.uint8, device=device)
            self.offset = 0
        else:
            self.pitch, self.storage, self.offset = int(pitch), _storage, int(_offset)

    @property
    def ptr(self):
        return self.storage.data_ptr() + self.offset

    def view(self):
        return KfxImage(self.pitch, self.ptr, self.w, self.h)

    def ref(self):
        self._v = self.view()
        return C.byref(self._v)

    def SubImage(self, x, y, w, h):
        """Image::SubImage (Image.h:423-454): same pitch, offset pointer."""
        assert x + w <= self.w and y + h <= self.h
        return Image(w, h, self.kind, pitch=self.pitch, _storage=self.storage,
                     _offset=self.offset + y * self.pitch + x * self.elem)

    def MemcpyFromHost(self, arr):
        """Image::MemcpyFromHost (Image.h:181-197)."""
        arr = np.ascontiguousarray(arr, dtype=self.np_dtype)
        rowbytes = self.w * self.elem
        assert arr.nbytes == rowbytes * self.h, (arr.shape, self.w, self.h, self.kind)
        host = torch.from_numpy(arr.view(np.uint8).reshape(self.h, rowbytes))
        dst = self.storage[self.offset: self.offset + self.pitch * (self.h - 1) + rowbytes]
        dst = torch.as_strided(dst, (self.h, rowbytes), (self.pitch, 1))
        dst.copy_(host)
        return self

    def MemcpyFromPinned(self, host):
        """Image::MemcpyFromHost from page-locked memory, asynchronous on the current stream: `host` is a pinned uint8 torch
        tensor of h x (w * element size) bytes (pinned_like() makes one).  The copy is ordered before the launches that follow
        on the stream; the caller keeps `host` unchanged until it has run."""
        rowbytes = self.w * self.elem
        dst = self.storage[self.offset: self.offset + self.pitch * (self.h - 1) + rowbytes]
        dst = torch.as_strided(dst, (self.h, rowbytes), (self.pitch, 1))
        dst.copy_(host, non_blocking=True)
        return self

    def pinned_like(self, arr):
        """A page-locked host copy of `arr` in the layout MemcpyFromPinned takes."""
        arr = np.ascontiguousarray(arr, dtype=self.np_dtype)
        rowbytes = self.w * self.elem
        assert arr.nbytes == rowbytes * self.h
        return torch.from_numpy(arr.view(np.uint8).reshape(self.h, rowbytes).copy()).pin_memory()

    def MemcpyToHost(self):
        """Image::MemcpyToHost (Image.h:199-213) -> numpy (h, w[, 4])."""
        rowbytes = self.w * self.elem
        src = self.storage[self.offset: self.offset + self.pitch * (self.h - 1) + rowbytes]
        src = torch.as_strided(src, (self.h, rowbytes), (self.pitch, 1))
        out = src.cpu().numpy().copy().view(self.np_dtype)
        return out.reshape(self.h, self.w, self.channels) if self.channels > 1 else out.reshape(self.h, self.w)

    def tensor(self):
        """Strided torch view (h, w[, ch]) of the pixels, for device-side comparisons."""
        tdt = {np.float32: torch.float32, np.uint16: torch.uint16, np.uint8: torch.uint8}[self.np_dtype]
        isz = np.dtype(self.np_dtype).itemsize
        flat = self.storage[self.offset:].view(tdt) if (self.offset % isz == 0) else None
        if self.channels > 1:
            return torch.as_strided(flat, (self.h, self.w, self.channels), (self.pitch // isz, self.channels, 1))
        return torch.as_strided(flat, (self.h, self.w), (self.pitch // isz, 1))


class BoundedVolume:
    """roo::BoundedVolume<SDF_t, TargetDevice, Manage> (BoundedVolume.h:10-170): x-fastest AoS
    {val, w} cells, row `pitch`, slice `img_pitch = pitch*h` (Memory.h:70-78)."""

    def __init__(self, w, h, d, boxmin=(-1, -1, -1), boxmax=(1, 1, 1), device="cuda", pitch=None,
                 _storage=None, _offset=0, _img_pitch=None, kind="f32"):
        """kind = "f32": SDF_t {float val; float w;} (8 B); kind = "f16": SDF_h {half val; half w;} (4 B);
        kind = "c32": BoundedVolume<float>, the grey-level colour volume (4 B)."""
        self.kind = kind
        self.ELEM = {"f32": 8, "f16": 4, "c32": 4}[kind]
        self.w, self.h, self.d = int(w), int(h), int(d)
        self.boxmin = np.asarray(boxmin, np.float32).copy()
        self.boxmax = np.asarray(boxmax, np.float32).copy()
        if _storage is None:
            self.pitch = int(pitch) if pitch else (self.w * self.ELEM + PITCH_ALIGN - 1) // PITCH_ALIGN * PITCH_ALIGN
            self.img_pitch = self.pitch * self.h
            self.storage = torch.zeros(max(self.img_pitch * self.d, 1), dtype=torch.uint8, device=device)
            self.offset = 0
        else:
            self.pitch, self.img_pitch, self.storage, self.offset = int(pitch), int(_img_pitch), _storage, int(_offset)

    @property
    def ptr(self):
        return self.storage.data_ptr() + self.offset

    def view(self):
        v = KfxVolume(self.pitch, self.ptr, self.w, self.h, self.img_pitch, self.d)
        for i in range(3):
            v.boxmin[i] = float(self.boxmin[i])
            v.boxmax[i] = float(self.boxmax[i])
        return v

    def ref(self):
        self._v = self.view()
        return C.byref(self._v)

    def IsValid(self):
        """BoundedVolume::IsValid (BoundedVolume.h:83-87)."""
        return self.w >= 8 and self.h >= 8 and self.d >= 8

    def VoxelSizeUnits(self):
        """BoundedVolume::VoxelSizeUnits (BoundedVolume.h:67-76), float32."""
        return ((self.boxmax - self.boxmin) / np.array([self.w - 1, self.h - 1, self.d - 1], np.float32)).astype(np.float32)

    def VoxelPositionInUnits(self, x, y, z):
        """BoundedVolume::VoxelPositionInUnits (BoundedVolume.h:115-125), float32."""
        s = (self.boxmax - self.boxmin).astype(np.float32)
        f = np.float32
        return np.array([self.boxmin[0] + s[0] * f(x) / f(self.w - 1), self.boxmin[1] + s[1] * f(y) / f(self.h - 1),
                         self.boxmin[2] + s[2] * f(z) / f(self.d - 1)], np.float32)

    def SubVolume(self, start, size):
        """Volume::SubVolume (Volume.h:305-311): same pitches, offset pointer."""
        off = self.offset + start[2] * self.img_pitch + start[1] * self.pitch + start[0] * self.ELEM
        return BoundedVolume(size[0], size[1], size[2], self.boxmin, self.boxmax, pitch=self.pitch,
                             _storage=self.storage, _offset=off, _img_pitch=self.img_pitch, kind=self.kind)

    def SubBoundingVolume(self, rmin, rmax):
        """BoundedVolume::SubBoundingVolume (BoundedVolume.h:137-165) in float32 host arithmetic."""
        f = np.float32
        rmin, rmax = np.asarray(rmin, f), np.asarray(rmax, f)
        size = (self.boxmax - self.boxmin).astype(f)
        min_fv = ((rmin - self.boxmin) / size).astype(f)
        max_fv = ((rmax - self.boxmin) / size).astype(f)
        dims1 = np.array([self.w - 1, self.h - 1, self.d - 1], f)
        with np.errstate(invalid="ignore"):
            min_v = np.maximum(dims1 * min_fv, f(0)).astype(np.int32)           # float -> int truncation
            max_v = np.minimum(np.ceil(dims1 * max_fv), dims1).astype(np.int32)
        size_v = np.maximum((max_v - min_v) + 1, 0)
        sub = self.SubVolume(tuple(int(v) for v in min_v), tuple(int(v) for v in size_v))
        sub.boxmin = self.VoxelPositionInUnits(*min_v)
        sub.boxmax = self.VoxelPositionInUnits(*max_v)
        return sub

    def ZSlab(self, z0, z1):
        """View of slices [z0, z1) with the bbox of its first/last plane (what SubBoundingVolume
        produces for an axis-aligned slab, BoundedVolume.h:156-164)."""
        sub = self.SubVolume((0, 0, z0), (self.w, self.h, z1 - z0))
        sub.boxmin = self.VoxelPositionInUnits(0, 0, z0)
        sub.boxmax = self.VoxelPositionInUnits(self.w - 1, self.h - 1, z1 - 1)
        return sub

    def planes(self, z0, z1):
        """Contiguous uint8 view of the z-slices [z0, z1) (img_pitch bytes each), for halo exchange."""
        assert self.img_pitch == self.pitch * self.h
        return self.storage[self.offset + z0 * self.img_pitch: self.offset + z1 * self.img_pitch]

    def tensor(self):
        """Strided torch view (d, h, w, 2) of the cells (float32 or float16); (d, h, w, 1) for a colour volume."""
        if self.kind == "c32":
            flat = self.storage[self.offset:].view(torch.float32)
            return torch.as_strided(flat, (self.d, self.h, self.w, 1), (self.img_pitch // 4, self.pitch // 4, 1, 1))
        if self.kind == "f16":
            flat = self.storage[self.offset:].view(torch.float16)
            return torch.as_strided(flat, (self.d, self.h, self.w, 2), (self.img_pitch // 2, self.pitch // 2, 2, 1))
        flat = self.storage[self.offset:].view(torch.float32)
        return torch.as_strided(flat, (self.d, self.h, self.w, 2), (self.img_pitch // 4, self.pitch // 4, 2, 1))

    def MemcpyToHost(self):
        return self.tensor().cpu().numpy().copy()

    def MemcpyFromHost(self, arr):
        self.tensor().copy_(torch.from_numpy(np.ascontiguousarray(arr, np.float16 if self.kind == "f16" else np.float32)))
        return self


def FitToFrustum(T_wc, w, h, K, near, far):
    """BoundingBox::FitToFrustum (BoundingBox.h:72-96), float32 host arithmetic."""
    f = np.float32
    T = np.asarray(T_wc, f).reshape(3, 4)
    K = np.asarray(K, f)
    c = T[:, 3]
    lo = np.full(3, np.finfo(f).max, f)
    hi = np.full(3, -np.finfo(f).max, f)
    for dist in (f(near), f(far)):
        for (u, v) in ((0, 0), (w, 0), (0, h), (w, h)):
            r = np.array([(f(u) - K[2]) / K[0], (f(v) - K[3]) / K[1], f(1)], f)
            rw = np.array([T[i, 0] * r[0] + T[i, 1] * r[1] + T[i, 2] * r[2] for i in range(3)], f)
            p = (c + dist * rw).astype(f)
            hi = np.maximum(p, hi)
            lo = np.minimum(p, lo)
    return lo, hi


MATH_EXACT, MATH_FAST = 0, 1


def set_math_mode(mode):
    """kfx_set_math_mode: 'exact' (default, bit-identical to the oracle) or 'fast' (perf build)."""
    m = {"exact": MATH_EXACT, "fast": MATH_FAST}.get(mode, mode)
    prev = _lib.load().kfx_set_math_mode(int(m))
    if prev < 0:
        _lib.check(prev)
    return "fast" if prev == MATH_FAST else "exact"


def get_math_mode():
    return "fast" if _lib.load().kfx_get_math_mode() == MATH_FAST else "exact"


# ---- operators ------------------------------------------------------------------

class SdfSummary:
    """kfx_sdf_summary of a BoundedVolume (fp32 cells): per 8 x 8 x 8 cells the range of the stored values, kept current by
    SdfFuse(..., summary=) / SdfReset(..., summary=) and used by RaycastSdf(..., summary=) to step through uniformly free
    or never-observed space without reading the volume.  Views of the volume may be passed to those calls; after any
    other write to the volume call invalidate()."""

    def __init__(self, vol):
        assert vol.kind == "f32"
        self.vol = vol
        self.handle = C.c_void_p()
        _lib.check(_lib.load().kfx_sdf_summary_create(C.byref(self.handle), vol.ref()))

    def invalidate(self, stream=None):
        _lib.check(_lib.load().kfx_sdf_summary_invalidate(self.handle, _stream(stream)))

    def rebuild(self, stream=None):
        """kfx_sdf_summary_rebuild: recompute every brick from what the volume holds (after writers that do not track)."""
        _lib.check(_lib.load().kfx_sdf_summary_rebuild(self.handle, _stream(stream)))

    def __del__(self):
        try:
            if self.handle is not None and self.handle.value:
                _lib.load().kfx_sdf_summary_destroy(self.handle)
                self.handle = None
        except Exception:   # interpreter shutdown: the process is going away with its device memory
            pass


class _FrameSummary:
    """The summary a kfx_frame owns, in the shape the operators' summary= argument takes (handle only; never destroyed here)."""

    def __init__(self, handle):
        self.handle = C.c_void_p(handle)

    def invalidate(self, stream=None):
        _lib.check(_lib.load().kfx_sdf_summary_invalidate(self.handle, _stream(stream)))

    def rebuild(self, stream=None):
        _lib.check(_lib.load().kfx_sdf_summary_rebuild(self.handle, _stream(stream)))


class Frame:
    """kfx_frame (include/kfx.h): one frame of the application's loop (main.cpp:200-356, known poses) -- BilateralFilter ->
    DepthToVbo -> NormalsFromVbo -> SdfFuse -> RaycastSdf -- enqueued by ONE library call on the current stream, with device
    events around its parts.  The images and the volume are the caller's containers (kept alive here); results are
    bit-identical to calling the operators one by one."""

    PREPROCESS, FUSE, RAYCAST, ALL = 1, 2, 4, 7
    EVENTS_ALL, EVENTS_FUSE, EVENTS_NONE = 15, 6, 0   # kfx_frame_set_timing: all four events / the two around SdfFuse / none
    FIELDS = 5   # timings(): preprocess, SdfFuse, RaycastSdf, whole frame, period to the next frame's start (ms)

    def __init__(self, vol, raw, filtered, vbo, normals, ray_d, ray_n, ray_i, K, bilateral, near, far, trunc_dist, max_w, mincostheta,
                 full_extent=False, timing_slots=0):
        assert vol.kind == "f32"
        self._keep = (vol, raw, filtered, vbo, normals, ray_d, ray_n, ray_i)
        cfg = _lib.KfxFrameConfig()
        cfg.vol = vol.view()
        cfg.raw, cfg.filtered, cfg.vbo, cfg.normals = raw.view(), filtered.view(), vbo.view(), normals.view()
        cfg.ray_depth, cfg.ray_norm, cfg.ray_img = ray_d.view(), ray_n.view(), ray_i.view()
        for i, v in enumerate(np.asarray(K, np.float32).reshape(4)):
            cfg.K[i] = float(v)
        cfg.bilateral_gs, cfg.bilateral_gr = float(bilateral["gs"]), float(bilateral["gr"])
        cfg.bilateral_minval, cfg.bilateral_size = float(bilateral["minval"]), int(bilateral["size"])
        cfg.near, cfg.far, cfg.trunc_dist, cfg.max_w, cfg.mincostheta = float(near), float(far), float(trunc_dist), float(max_w), float(mincostheta)
        cfg.fuse_flags = 1 if full_extent else 0
        cfg.timing_slots = int(timing_slots)
        self.timing_slots = int(timing_slots)
        self.handle = C.c_void_p()
        _lib.check(_lib.load().kfx_frame_create(C.byref(self.handle), C.byref(cfg)))
        L = _lib.load()
        self._step, self._h = L.kfx_frame_step, self.handle

    def __del__(self):
        try:
            if self.handle is not None and self.handle.value:
                _lib.load().kfx_frame_destroy(self.handle)
                self.handle = None
        except Exception:   # interpreter shutdown
            pass

    def reset(self, stream=None):
        _lib.check(_lib.load().kfx_frame_reset(self.handle, _stream(stream)))

    def set_track(self, on, stream=None):
        _lib.check(_lib.load().kfx_frame_set_track(self.handle, 1 if on else 0, _stream(stream)))

    @property
    def track(self):
        return bool(_lib.load().kfx_frame_get_track(self.handle))

    @property
    def count(self):
        return int(_lib.load().kfx_frame_count(self.handle))

    def summary(self):
        h = _lib.load().kfx_frame_summary(self.handle)
        return _FrameSummary(h) if h else None

    def set_timing(self, mask):
        """Which events the following steps record (EVENTS_*): every event is a marker between two launches and costs the stream
        a few microseconds; fields of timings() that need an unrecorded event are NaN."""
        _lib.check(_lib.load().kfx_frame_set_timing(self.handle, int(mask)))

    def step(self, T_wc, T_cw=None, raw=None, parts=0, stream=None):
        t, _t = _fp(T_wc, 12)
        if T_cw is not None:
            ti, _ti = _fp(T_cw, 12)
        else:
            ti = None
        e = self._step(self._h, raw.ref() if raw is not None else None, t, ti, parts, _stream(stream))
        if e:
            _lib.check(e)

    def timings(self, first, n):
        """(n, 5) float32 array in ms for frames first .. first + n - 1: preprocess, SdfFuse, RaycastSdf, whole frame, period
        (start of the frame to the start of the next; NaN for the most recent frame).  Waits for the last of them only."""
        out = np.empty((max(n, 0), self.FIELDS), np.float32)
        if n > 0:
            _lib.check(_lib.load().kfx_frame_timings(self.handle, int(first), int(n), out.ctypes.data_as(_lib.PF)))
        return out


def SdfFuse(vol, depth, norm, T_cw, K, trunc_dist, maxw, mincostheta, full_extent=False, stream=None, slab=None, summary=None):
    """roo::SdfFuse (cu_sdffusion.h:13-14).  slab = (full_d, z_offset, full_zmin, full_zmax) integrates `vol`
    as planes [z_offset, z_offset + d) of a larger volume (kfx_sdf_fuse_slab), bit-identically to the same
    planes of the monolithic volume.  summary: keep this SdfSummary current (kfx_sdf_fuse_tracked, same volume bits)."""
    t, _t = _fp(T_cw, 12)
    k, _k = _fp(K, 4)
    if summary is not None:
        assert slab is None and vol.kind == "f32"
        _lib.check(_lib.load().kfx_sdf_fuse_tracked(vol.ref(), summary.handle, depth.ref(), norm.ref(), t, k, trunc_dist, maxw, mincostheta,
                                                    1 if full_extent else 0, _stream(stream)))
        return
    if slab is not None:
        sl = _lib.KfxSlab(int(slab[0]), int(slab[1]), float(slab[2]), float(slab[3]))
        fn = _lib.load().kfx_sdf_fuse_slab_h if vol.kind == "f16" else _lib.load().kfx_sdf_fuse_slab
        # full_extent="slab": the reference's extents on the whole volume (KFX_FUSE_SLAB_EXTENT)
        _lib.check(fn(vol.ref(), C.byref(sl), depth.ref(), norm.ref(), t, k, trunc_dist, maxw,
                                                 mincostheta, 2 if full_extent == "slab" else (1 if full_extent else 0), _stream(stream)))
        return
    fn = _lib.load().kfx_sdf_fuse_h if vol.kind == "f16" else _lib.load().kfx_sdf_fuse
    _lib.check(fn(vol.ref(), depth.ref(), norm.ref(), t, k, trunc_dist, maxw, mincostheta,
                  1 if full_extent else 0, _stream(stream)))


class SdfFuseBound:
    """SdfFuse with everything but the pose bound ahead of time (the tracked loop: the launch follows the pose's arrival on the
    host, and every microsecond of argument marshalling in between is a microsecond the device idles).  call(T_cw: 12 float32)."""

    def __init__(self, vol, depth, norm, K, trunc_dist, maxw, mincostheta, full_extent=False, stream=None, summary=None):
        assert vol.kind == "f32"
        L = _lib.load()
        self._keep = (vol, depth, norm, summary)
        self._v, self._d, self._n = vol.view(), depth.view(), norm.view()
        self._k, self._ka = _fp(K, 4)
        self._T = (C.c_float * 12)()
        # the stream is resolved at every call (torch's current stream may have changed since construction: the launch must stay
        # ordered behind the maps it reads and before the raycast that follows); everything else is marshalled once
        tail = (float(trunc_dist), float(maxw), float(mincostheta), 1 if full_extent else 0)
        if summary is not None:
            self._call = lambda: L.kfx_sdf_fuse_tracked(C.byref(self._v), summary.handle, C.byref(self._d), C.byref(self._n), self._T, self._k, *tail, _stream(stream))
        else:
            self._call = lambda: L.kfx_sdf_fuse(C.byref(self._v), C.byref(self._d), C.byref(self._n), self._T, self._k, *tail, _stream(stream))

    def __call__(self, T_cw):
        if T_cw.dtype != np.float32 or T_cw.size != 12 or not T_cw.flags.c_contiguous:
            T_cw = np.ascontiguousarray(np.asarray(T_cw, np.float32).reshape(12))
        C.memmove(self._T, T_cw.ctypes.data, 48)
        _lib.check(self._call())


def PoseStep(T_wl, T_lp):
    """kfx_pose_step: (T_wl * T_lp^-1 as 4x4 float64, its inverse as 3x4 float32) -- main.cpp:337 / :345 on the host, term by term."""
    a = np.ascontiguousarray(T_wl, np.float64)
    b = np.ascontiguousarray(T_lp, np.float64)
    out = np.eye(4)
    inv = np.empty((3, 4), np.float32)
    PD = C.POINTER(C.c_double)
    _lib.check(_lib.load().kfx_pose_step(a.ctypes.data_as(PD), b.ctypes.data_as(PD), out.ctypes.data_as(PD), inv.ctypes.data_as(_lib.PF)))
    return out, inv


def SdfFuseCount(vol, depth, norm, T_cw, K, trunc_dist, mincostheta, full_extent=False, stream=None):
    """Diagnostics: number of voxels SdfFuse would update for this frame (kfx_sdf_fuse_count)."""
    t, _t = _fp(T_cw, 12)
    k, _k = _fp(K, 4)
    cnt = torch.zeros(1, dtype=torch.int64, device=vol.storage.device)
    _lib.check(_lib.load().kfx_sdf_fuse_count(vol.ref(), depth.ref(), norm.ref(), t, k, trunc_dist, mincostheta,
                                              1 if full_extent else 0, C.c_void_p(cnt.data_ptr()), _stream(stream)))
    return int(cnt.item())


def RaycastSdfCount(vol, w, h, T_wc, K, near, far, trunc_dist, subpix=True, stream=None, summary=None):
    """Diagnostics (kfx_raycast_sdf_count): what RaycastSdf's march for a w x h image reads -- dict(samples, rays, hits, U =
    distinct voxels touched).  summary: the march RaycastSdf(..., summary=) runs instead (kfx_raycast_sdf_count_tracked), with
    lookups = table look-ups and table_bytes = what a workgroup stages of the class tables (0: it fell back to the plain march)."""
    t, _t = _fp(T_wc, 12)
    k, _k = _fp(K, 4)
    bitmap = torch.zeros((vol.w * vol.h * vol.d + 31) // 32, dtype=torch.int32, device=vol.storage.device)
    if summary is not None:
        cnt = torch.zeros(6, dtype=torch.int64, device=vol.storage.device)
        _lib.check(_lib.load().kfx_raycast_sdf_count_tracked(vol.ref(), summary.handle, w, h, t, k, near, far, trunc_dist, 1 if subpix else 0,
                                                             C.c_void_p(bitmap.data_ptr()), C.c_void_p(cnt.data_ptr()), _stream(stream)))
        c = cnt.tolist()
        return dict(samples=c[0], rays=c[1], hits=c[2], U=c[3], lookups=c[4], table_bytes=c[5])
    cnt = torch.zeros(4, dtype=torch.int64, device=vol.storage.device)
    fn = _lib.load().kfx_raycast_sdf_count_h if vol.kind == "f16" else _lib.load().kfx_raycast_sdf_count
    _lib.check(fn(vol.ref(), w, h, t, k, near, far, trunc_dist, 1 if subpix else 0,
                  C.c_void_p(bitmap.data_ptr()), C.c_void_p(cnt.data_ptr()), _stream(stream)))
    c = cnt.tolist()
    return dict(samples=c[0], rays=c[1], hits=c[2], U=c[3])


def RaycastSdf(depth, norm, img, vol, T_wc, K, near, far, trunc_dist, subpix=True, stream=None, summary=None):
    """roo::RaycastSdf (cu_raycast.h:13-14).  summary: take the steps through uniform bricks from this SdfSummary
    (kfx_raycast_sdf_tracked: bit-identical images in exact numerics, within the fast-mode tolerance in fast numerics)."""
    t, _t = _fp(T_wc, 12)
    k, _k = _fp(K, 4)
    if summary is not None:
        _lib.check(_lib.load().kfx_raycast_sdf_tracked(depth.ref(), norm.ref(), img.ref(), vol.ref(), summary.handle, t, k, near, far,
                                                       trunc_dist, 1 if subpix else 0, _stream(stream)))
        return
    fn = _lib.load().kfx_raycast_sdf_h if vol.kind == "f16" else _lib.load().kfx_raycast_sdf
    _lib.check(fn(depth.ref(), norm.ref(), img.ref(), vol.ref(), t, k, near, far, trunc_dist, 1 if subpix else 0,
                  _stream(stream)))


def RaycastSdfLevels(outputs, vol, T_wc, K_levels, near, far, trunc_dist, subpix=True, stream=None, summary=None):
    """kfx_raycast_sdf_levels: the tracking loop's per-level RaycastSdf calls (main.cpp:280-288) as one launch.
    outputs = [(depth, norm, img[, vbo]), ...] per level, K_levels the matching intrinsics; images identical to per-level
    calls.  A fourth image per level receives DepthToVbo(depth, K) from the same launch (main.cpp:286).
    summary: an SdfSummary of the volume (kfx_raycast_sdf_levels_tracked), as in RaycastSdf."""
    n = len(outputs)
    assert n == len(K_levels)
    views = [[o[k].view() for o in outputs] for k in range(3)]   # KfxImage structs, kept alive over the call
    ptrs = [(_lib.PI * n)(*[C.pointer(v) for v in views[k]]) for k in range(3)]
    vviews = [o[3].view() if len(o) > 3 and o[3] is not None else None for o in outputs]
    vptrs = (_lib.PI * n)(*[C.pointer(v) if v is not None else None for v in vviews]) if any(v is not None for v in vviews) else None
    t, _t = _fp(T_wc, 12)
    k, _k = _fp(np.concatenate([np.asarray(K, np.float32).reshape(4) for K in K_levels]) if n else np.zeros(0, np.float32), 4 * n)
    if summary is not None:
        _lib.check(_lib.load().kfx_raycast_sdf_levels_tracked(n, ptrs[0], ptrs[1], ptrs[2], vptrs, vol.ref(), summary.handle, t, k, near, far,
                                                              trunc_dist, 1 if subpix else 0, _stream(stream)))
        return
    fn = _lib.load().kfx_raycast_sdf_levels_h if vol.kind == "f16" else _lib.load().kfx_raycast_sdf_levels
    _lib.check(fn(n, ptrs[0], ptrs[1], ptrs[2], vptrs, vol.ref(), t, k, near, far, trunc_dist, 1 if subpix else 0, _stream(stream)))


def BilateralFilter(dOut, dIn, gs, gr, size, minval=None, stream=None):
    """roo::BilateralFilter<float,Ti> (cu_bilateral.h:9-19); minval=None selects the overload
    without the validity threshold."""
    L = _lib.load()
    if dIn.kind == "f32":
        _lib.check(L.kfx_bilateral_f32(dOut.ref(), dIn.ref(), gs, gr, size, 0.0 if minval is None else minval,
                                       0 if minval is None else 1, _stream(stream)))
    elif dIn.kind == "u16":
        if minval is None:
            raise TypeError("BilateralFilter<float,unsigned short> is only instantiated with minval (cu_bilateral.cu:104)")
        _lib.check(L.kfx_bilateral_u16(dOut.ref(), dIn.ref(), gs, gr, size, int(minval), _stream(stream)))
    elif dIn.kind == "u8":
        if minval is not None:
            raise TypeError("BilateralFilter<float,unsigned char> is only instantiated without minval (cu_bilateral.cu:53)")
        _lib.check(L.kfx_bilateral_u8(dOut.ref(), dIn.ref(), gs, gr, size, _stream(stream)))
    else:
        raise TypeError(dIn.kind)


def DepthToVbo(dVbo, dDepth, K, scale=1.0, stream=None):
    """roo::DepthToVbo<T> (cu_depth_tools.h:19-21)."""
    k, _k = _fp(K, 4)
    L = _lib.load()
    fn = {"f32": L.kfx_depth_to_vbo_f32, "u16": L.kfx_depth_to_vbo_u16}[dDepth.kind]
    _lib.check(fn(dVbo.ref(), dDepth.ref(), k, scale, _stream(stream)))


def NormalsFromVbo(dN, dV, stream=None):
    """roo::NormalsFromVbo (cu_normals.h:9-10)."""
    _lib.check(_lib.load().kfx_normals_from_vbo(dN.ref(), dV.ref(), _stream(stream)))


def SdfReset(vol, trunc_dist, stream=None, summary=None):
    """roo::SdfReset(BoundedVolume<SDF_t>, float) (cu_sdffusion.h:20).  summary: the whole volume's SdfSummary, set to match."""
    if summary is not None:
        _lib.check(_lib.load().kfx_sdf_reset_tracked(vol.ref(), summary.handle, trunc_dist, _stream(stream)))
        return
    fn = _lib.load().kfx_sdf_reset_h if vol.kind == "f16" else _lib.load().kfx_sdf_reset
    _lib.check(fn(vol.ref(), trunc_dist, _stream(stream)))


def SdfSphere(vol, center, r, stream=None):
    """roo::SdfSphere (cu_sdffusion.h:26)."""
    c, _c = _fp(center, 3)
    fn = _lib.load().kfx_sdf_sphere_h if vol.kind == "f16" else _lib.load().kfx_sdf_sphere
    _lib.check(fn(vol.ref(), c, r, _stream(stream)))


def ElementwiseScaleBias(b, a, s, offset=0.0, stream=None):
    """roo::ElementwiseScaleBias<float,float,float> (cu_operations.h; main.cpp:208): b = s*a + offset."""
    _lib.check(_lib.load().kfx_elementwise_scale_bias_f32(b.ref(), a.ref(), s, offset, _stream(stream)))


def BoxHalfIgnoreInvalid(out, inp, stream=None):
    """roo::BoxHalfIgnoreInvalid<float,float,float> (cu_resample.h): NaN-aware 2x2 mean."""
    _lib.check(_lib.load().kfx_box_half_ignore_invalid_f32(out.ref(), inp.ref(), _stream(stream)))


class Pyramid:
    """roo::Pyramid<T, Levels, TargetDevice, Manage> (Pyramid.h:9-137): level l is (w >> l) x (h >> l)."""

    def __init__(self, w, h, levels, kind="f32"):
        self.imgs = [Image(w >> l, h >> l, kind) for l in range(levels) if (w >> l) > 0 and (h >> l) > 0]

    def __getitem__(self, l):
        return self.imgs[l]

    def __len__(self):
        return len(self.imgs)


def BoxReduceIgnoreInvalid(pyramid, stream=None):
    """roo::BoxReduceIgnoreInvalid<T,Levels,UpType> (reduce.h:48-59): fill levels 1.. from level 0."""
    for l in range(1, len(pyramid)):
        BoxHalfIgnoreInvalid(pyramid[l], pyramid[l - 1], stream)


def CompositePack(depth, norm, img, key, rank, stream=None):
    """kfx_composite_pack: key (int64 tensor, w*h) = (depth bits << 8) | rank, +inf for misses."""
    _lib.check(_lib.load().kfx_composite_pack(depth.ref(), norm.ref(), img.ref(), C.c_void_p(key.data_ptr()), rank, _stream(stream)))


def CompositeSelect(depth, norm, img, key, payload, rank, stream=None):
    """kfx_composite_select: payload (float32 tensor, w*h*5) = winner's {normal, shade}, zero elsewhere."""
    _lib.check(_lib.load().kfx_composite_select(depth.ref(), norm.ref(), img.ref(), C.c_void_p(key.data_ptr()),
                                                C.c_void_p(payload.data_ptr()), rank, _stream(stream)))


def CompositeUnpack(depth, norm, img, key, payload, stream=None):
    """kfx_composite_unpack: write the merged depth / normal / shade images."""
    _lib.check(_lib.load().kfx_composite_unpack(depth.ref(), norm.ref(), img.ref(), C.c_void_p(key.data_ptr()),
                                                C.c_void_p(payload.data_ptr()), _stream(stream)))


STRIP_PLANES = 5   # KFX_COMPOSITE_STRIP_PLANES


def CompositeStripPixels(w, h, world):
    """kfx_composite_strip_pixels: pixels per strip of the direct-send composite (the last strip is padded)."""
    return int(_lib.load().kfx_composite_strip_pixels(w, h, world))


def CompositeStripsPack(depth, norm, img, send, world, offset=0, rank_stride=0, stream=None):
    """kfx_composite_strips_pack: this rank's images cut into strips; strip j's 5 planes of S floats start at float
    offset + j * rank_stride of the tensor `send` (rank_stride 0: dense, world x 5 x S)."""
    _lib.check(_lib.load().kfx_composite_strips_pack(depth.ref(), norm.ref(), img.ref(), C.c_void_p(send.data_ptr() + 4 * offset), rank_stride, world,
                                                     _stream(stream)))


def CompositeStripsMerge(recv, merged, strip_pixels, world, offset=0, rank_stride=0, merged_offset=0, stream=None):
    """kfx_composite_strips_merge: merged (5 x S at float merged_offset) = per pixel the nearest of the world copies in recv
    (rank r's at float offset + r * rank_stride)."""
    _lib.check(_lib.load().kfx_composite_strips_merge(C.c_void_p(recv.data_ptr() + 4 * offset), C.c_void_p(merged.data_ptr() + 4 * merged_offset),
                                                      strip_pixels, rank_stride, world, _stream(stream)))


def CompositeStripsUnpack(depth, norm, img, strips, world, offset=0, rank_stride=0, stream=None):
    """kfx_composite_strips_unpack: the gathered strips back into the depth / normal / shade images."""
    _lib.check(_lib.load().kfx_composite_strips_unpack(depth.ref(), norm.ref(), img.ref(), C.c_void_p(strips.data_ptr() + 4 * offset), rank_stride, world,
                                                       _stream(stream)))


def RaycastSdfSlab(state, init, vol, slab, own_lo, own_hi, w, h, T_wc, K, near, far, trunc_dist, subpix=True, stream=None):
    """kfx_raycast_sdf_slab: one round of the exact multi-GPU march; `state` is a dense float32 tensor (9, h, w)."""
    assert state.dtype == torch.float32 and state.is_contiguous() and tuple(state.shape) == (9, h, w)
    t, _t = _fp(T_wc, 12)
    k, _k = _fp(K, 4)
    sl = _lib.KfxSlab(int(slab[0]), int(slab[1]), float(slab[2]), float(slab[3]))
    fn = _lib.load().kfx_raycast_sdf_slab_h if vol.kind == "f16" else _lib.load().kfx_raycast_sdf_slab
    _lib.check(fn(C.c_void_p(state.data_ptr()), 1 if init else 0, vol.ref(), C.byref(sl), own_lo, own_hi,
                                                w, h, t, k, near, far, trunc_dist, 1 if subpix else 0, _stream(stream)))


def RaycastStateToImages(depth, norm, img, state, stream=None):
    """kfx_raycast_state_to_images: final march state -> depth / normal / shade images."""
    _lib.check(_lib.load().kfx_raycast_state_to_images(depth.ref(), norm.ref(), img.ref(), C.c_void_p(state.data_ptr()), _stream(stream)))


class LeastSquaresSystem:
    """roo::LeastSquaresSystem<float,6> (Mat.h:483-520) on the host: JTy (6,), JTJ expanded to the symmetric
    6x6 matrix (SymMat -> Mat conversion, Mat.h:359-372), sqErr, obs; `raw` keeps the 21 unique elements."""

    def __init__(self, JTy, JTJ21, sqErr, obs):
        self.JTy = np.array(JTy, np.float32)
        self.raw = np.array(JTJ21, np.float32)
        self.JTJ = np.zeros((6, 6), np.float32)
        i = 0
        for r in range(6):
            for c in range(r + 1):
                self.JTJ[r, c] = self.JTJ[c, r] = self.raw[i]
                i += 1
        self.sqErr = np.float32(sqErr)
        self.obs = int(obs)


def PoseRefinementProjectiveIcpPointPlane(dPl, dPr, dNr, KT_lr, T_rl, c, dWorkspace, dDebug=None, stream=None):
    """cu_model_refinement.h:58-64 -> kfx_icp_point_plane.  dPl: live vertex map, dPr / dNr: model vertex map and
    normals (float4 images), dWorkspace: u8 image of >= (w/gcd(w,16))*(h/gcd(h,16))*116 bytes, dDebug: float4 image
    or None.  Returns the summed LeastSquaresSystem (blocking, as the reference's thrust::reduce is)."""
    kt, _kt = _fp(KT_lr, 12)
    t, _t = _fp(T_rl, 12)
    out = _lib.KfxLss6()
    _lib.check(_lib.load().kfx_icp_point_plane(dPl.ref(), dPr.ref(), dNr.ref(), kt, t, c, dWorkspace.ref(),
                                               dDebug.ref() if dDebug is not None else None, C.byref(out), _stream(stream)))
    return LeastSquaresSystem(list(out.JTy), list(out.JTJ), out.sqErr, out.obs)


def SdfFuseColor(vol, colorVol, depth, norm, T_cw, K, img, T_iw, Kimg, trunc_dist, max_w, mincostheta, full_extent=False, stream=None):
    """SdfFuse(vol, colorVol, depth, norm, T_cw, K, img, T_iw, Kimg, trunc_dist, max_w, mincostheta)
    (cu_sdffusion.h colour overload, cu_sdffusion.cu:70-138).  colorVol: BoundedVolume(kind="c32"), img: Image("u8x3")."""
    t, _t = _fp(T_cw, 12)
    k, _k = _fp(K, 4)
    ti, _ti = _fp(T_iw, 12)
    ki, _ki = _fp(Kimg, 4)
    _lib.check(_lib.load().kfx_sdf_fuse_color(vol.ref(), colorVol.ref(), depth.ref(), norm.ref(), t, k, img.ref(), ti, ki, trunc_dist,
                                              max_w, mincostheta, 1 if full_extent else 0, _stream(stream)))


def RaycastSdfColor(depth, norm, img, vol, colorVol, T_wc, K, near, far, trunc_dist, subpix=True, stream=None):
    """RaycastSdf(depth, norm, img, vol, colorVol, T_wc, K, near, far, trunc_dist, subpix) (cu_raycast.cu:119-196)."""
    t, _t = _fp(T_wc, 12)
    k, _k = _fp(K, 4)
    _lib.check(_lib.load().kfx_raycast_sdf_color(depth.ref(), norm.ref(), img.ref(), vol.ref(), colorVol.ref(), t, k, near, far,
                                                 trunc_dist, 1 if subpix else 0, _stream(stream)))


def ColorReset(colorVol, stream=None):
    """SdfReset(BoundedVolume<float>) = Fill(0.5) (cu_sdffusion.cu:166-169)."""
    _lib.check(_lib.load().kfx_color_reset(colorVol.ref(), _stream(stream)))


def RaycastBox(imgd, T_wc, K, boxmin, boxmax, stream=None):
    """RaycastBox(imgd, T_wc, K, bbox) (cu_raycast.h, cu_raycast.cu:202-240)."""
    t, _t = _fp(T_wc, 12)
    k, _k = _fp(K, 4)
    a, _a = _fp(boxmin, 3)
    b, _b = _fp(boxmax, 3)
    _lib.check(_lib.load().kfx_raycast_box(imgd.ref(), t, k, a, b, _stream(stream)))


def RaycastSphere(imgd, img, T_wc, K, center, r, stream=None):
    """RaycastSphere(imgd, img, T_wc, K, center, r) (cu_raycast.cu:246-279); img may be None."""
    t, _t = _fp(T_wc, 12)
    k, _k = _fp(K, 4)
    c, _c = _fp(center, 3)
    _lib.check(_lib.load().kfx_raycast_sphere(imgd.ref(), img.ref() if img is not None else None, t, k, c, r, _stream(stream)))


def RaycastPlane(imgd, img, T_wc, K, n_w, stream=None):
    """RaycastPlane(imgd, img, T_wc, K, n_w) (cu_raycast.cu:285-310): the plane n_w . x = -1."""
    t, _t = _fp(T_wc, 12)
    k, _k = _fp(K, 4)
    n, _n = _fp(n_w, 3)
    _lib.check(_lib.load().kfx_raycast_plane(imgd.ref(), img.ref(), t, k, n, _stream(stream)))


def SdfDistance(dist, depth, vol, T_wc, K, trunc_distance=0.0, stream=None):
    """SdfDistance(dist, depth, vol, T_wc, K, trunc_distance) (cu_sdffusion.cu:200-225)."""
    t, _t = _fp(T_wc, 12)
    k, _k = _fp(K, 4)
    _lib.check(_lib.load().kfx_sdf_distance(dist.ref(), depth.ref(), vol.ref(), t, k, trunc_distance, _stream(stream)))


def Disp2Depth(dIn, dOut, fu, fBaseline, fMinDisp=0.0, stream=None):
    """Disp2Depth(dIn, dOut, fu, fBaseline, fMinDisp) (cu_depth_tools.h:11)."""
    _lib.check(_lib.load().kfx_disp2depth(dIn.ref(), dOut.ref(), fu, fBaseline, fMinDisp, _stream(stream)))


def FilterBadKinectData(dFiltered, dKinectDepth, stream=None):
    """FilterBadKinectData(dFiltered, dKinectDepth) (cu_depth_tools.h:14-17): float or unsigned short millimetres."""
    L = _lib.load()
    fn = L.kfx_filter_bad_kinect_u16 if dKinectDepth.kind == "u16" else L.kfx_filter_bad_kinect_f32
    _lib.check(fn(dFiltered.ref(), dKinectDepth.ref(), _stream(stream)))


def ColourVbo(dId, dPd, dIc, KT_cd, stream=None):
    """ColourVbo(dId, dPd, dIc, KT_cd) (cu_depth_tools.h:30): dId Image("u8x4"), dPd float4 vertices, dIc Image("u8x3")."""
    t, _t = _fp(KT_cd, 12)
    _lib.check(_lib.load().kfx_colour_vbo(dId.ref(), dPd.ref(), dIc.ref(), t, _stream(stream)))


def BilateralFilterGuided(dOut, dIn, dImg, gs, gr, gc, size, stream=None):
    """BilateralFilter(dOut, dIn, dImg, gs, gr, gc, size) (cu_bilateral.h:21-25): joint bilateral, guide f32 or u8."""
    L = _lib.load()
    fn = L.kfx_bilateral_guided_u8 if dImg.kind == "u8" else L.kfx_bilateral_guided_f32
    _lib.check(fn(dOut.ref(), dIn.ref(), dImg.ref(), gs, gr, gc, size, _stream(stream)))


def DepthToVboNormals(vbo, nrm, depth, K, scale=1.0, stream=None):
    """DepthToVbo<float> + NormalsFromVbo in one launch (kfx_depth_to_vbo_normals_f32): identical outputs."""
    k, _k = _fp(K, 4)
    _lib.check(_lib.load().kfx_depth_to_vbo_normals_f32(vbo.ref(), nrm.ref(), depth.ref(), k, scale, _stream(stream)))


def DepthPyramidVboNormals(depth, vbo, nrm, K_levels, scale=1.0, stream=None):
    """kfx_depth_pyramid_vbo_normals_f32: BoxReduceIgnoreInvalid(depth) followed by DepthToVbo + NormalsFromVbo on every level
    (main.cpp:211-218), one launch; depth / vbo / nrm: per-level lists (or pyramids), depth[0] the input.  Same images as the
    per-level calls.  Levels of size zero (and what follows them) are left alone, as BoxReduceIgnoreInvalid leaves them."""
    n = 0
    while n < len(depth) and n < 4 and depth[n].w > 0 and depth[n].h > 0:
        n += 1
    if n == 0:
        return
    arrs = [(_lib.KfxImage * n)(*[im[l].view() for l in range(n)]) for im in (depth, vbo, nrm)]
    k, _k = _fp(np.concatenate([np.asarray(K_levels[l], np.float32).reshape(4) for l in range(n)]), 4 * n)
    _lib.check(_lib.load().kfx_depth_pyramid_vbo_normals_f32(arrs[0], arrs[1], arrs[2], k, n, scale, _stream(stream)))
    if len(depth) > n and n == 4 and depth[n].w > 0 and depth[n].h > 0:   # (more than four levels: the rest level by level)
        for l in range(n, len(depth)):
            if depth[l].w == 0 or depth[l].h == 0:
                break
            BoxHalfIgnoreInvalid(depth[l], depth[l - 1], stream)
            DepthToVboNormals(vbo[l], nrm[l], depth[l], K_levels[l], scale, stream)


def IcpRefine(kin_v, ray_v, ray_n, K_levels, its, icp_c, max_rmse, dWorkspace, dDebug=None, stream=None, before_wait=None):
    """kfx_icp_refine: the coarse-to-fine loop of main.cpp:301-337 enqueued as one kernel chain with the 6x6 solves on
    the device.  Per-level lists are indexed by pyramid level (0 = full resolution), as in the application; levels are
    processed from the coarsest down, the coarsest one rotation-only.  Returns (T_lp 4x4 float64, rmse, obs, good).
    before_wait: a callable run once everything is enqueued and before the pose is waited for (kfx_icp_refine_then): what it
    enqueues -- the next frame's pre-amble -- keeps the device busy while this thread wakes up."""
    n = len(K_levels)
    order = list(range(n - 1, -1, -1))
    arr = (_lib.KfxIcpLevel * n)()
    for slot, l in enumerate(order):
        arr[slot].Pl, arr[slot].Pr, arr[slot].Nr = kin_v[l].view(), ray_v[l].view(), ray_n[l].view()
        for i in range(4):
            arr[slot].K[i] = float(K_levels[l][i])
        arr[slot].iterations = int(its[l])
        arr[slot].rotation_only = 1 if (l == n - 1 and n > 1) else 0
    T = (C.c_double * 12)()
    rmse, obs, good = C.c_float(), C.c_uint(), C.c_int()
    if before_wait is not None:
        raised = []

        def hook(_user):   # (a Python exception must not unwind through C: it is re-raised after the call)
            try:
                before_wait()
            except BaseException as e:   # noqa: BLE001
                raised.append(e)
        cb = C.CFUNCTYPE(None, C.c_void_p)(hook)
        st = _lib.load().kfx_icp_refine_then(arr, n, icp_c, max_rmse, dWorkspace.ref(), dDebug.ref() if dDebug is not None else None, T,
                                             C.byref(rmse), C.byref(obs), C.byref(good), C.cast(cb, C.c_void_p), None, _stream(stream))
        if raised:
            raise raised[0]
        _lib.check(st)
    else:
        _lib.check(_lib.load().kfx_icp_refine(arr, n, icp_c, max_rmse, dWorkspace.ref(), dDebug.ref() if dDebug is not None else None, T,
                                              C.byref(rmse), C.byref(obs), C.byref(good), _stream(stream)))
    T4 = np.eye(4)
    T4[:3, :] = np.frombuffer(T, np.float64).reshape(3, 4)
    return T4, float(rmse.value), int(obs.value), bool(good.value)


def TextureDepth(img, keyframes, depth, norm, T_wd, Kdepth, phong=None, stream=None):
    """TextureDepth (cu_depth_tools.h:33-38).  keyframes: list of (Image("u8x3") or None, T_iw 3x4, K) -- one entry and
    phong=None for the single-keyframe form, up to 10 entries and a Phong image for the blended form."""
    n = len(keyframes)
    arr = (_lib.KfxKeyframe * max(n, 1))()
    for i, (kimg, T_iw, K) in enumerate(keyframes):
        for j, v in enumerate(np.asarray(K, np.float32).reshape(-1)):
            arr[i].K[j] = float(v)
        for j, v in enumerate(np.asarray(T_iw, np.float32).reshape(-1)):
            arr[i].T_iw[j] = float(v)
        arr[i].img = kimg.view() if kimg is not None else _lib.KfxImage(0, None, 0, 0)
    t, _t = _fp(T_wd, 12)
    k, _k = _fp(Kdepth, 4)
    _lib.check(_lib.load().kfx_texture_depth(img.ref(), arr, n, depth.ref(), norm.ref(), phong.ref() if phong is not None else None, t, k,
                                             _stream(stream)))
