"""ctypes front-end of the CPU parity oracle (oracle/kfx_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package `kangaroo_amd` never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class KfoImage(C.Structure):
    # roo::Image field order, Image.h:617-620
    _fields_ = [("pitch", C.c_size_t), ("ptr", C.c_void_p), ("w", C.c_size_t), ("h", C.c_size_t)]


class KfoVolume(C.Structure):
    # roo::BoundedVolume field order, Volume.h:363-369 + BoundedVolume.h:168
    _fields_ = [("pitch", C.c_size_t), ("ptr", C.c_void_p), ("w", C.c_size_t), ("h", C.c_size_t),
                ("img_pitch", C.c_size_t), ("d", C.c_size_t),
                ("boxmin", C.c_float * 3), ("boxmax", C.c_float * 3)]


class KfoSlab(C.Structure):
    _fields_ = [("full_d", C.c_size_t), ("z_offset", C.c_size_t), ("full_zmin", C.c_float), ("full_zmax", C.c_float)]


class KfoLss6(C.Structure):
    _fields_ = [("JTy", C.c_float * 6), ("JTJ", C.c_float * 21), ("sqErr", C.c_float), ("obs", C.c_uint)]


class KfoKeyframe(C.Structure):
    _fields_ = [("K", C.c_float * 4), ("T_iw", C.c_float * 12), ("img", KfoImage)]


class KfoRaycastStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("steps", C.c_uint64), ("hits", C.c_uint64)]


_SO_OVERRIDE = None


def use_native_build():
    """Switch this module to a -O3 -march=native build of the same source (oracle/_native/, built on the spot for the
    host it runs on; same -ffp-contract=off arithmetic).  Used by bench.py's cpu_baseline leg only -- the parity tests
    keep the portable build.  Returns a description of the build in use."""
    global _SO_OVERRIDE, _LIB
    so = os.path.join(_HERE, "_native", "libkfx_oracle_native.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        C.CDLL(so)
    except (subprocess.CalledProcessError, OSError):
        return "gcc -O2 -march=x86-64-v3 -ffp-contract=off -fopenmp (portable parity build; native rebuild failed)"
    _SO_OVERRIDE, _LIB = so, None
    return "gcc -O3 -march=native -ffp-contract=off -fopenmp"


def build(force=False):
    if _SO_OVERRIDE:
        return _SO_OVERRIDE
    so = os.path.join(_HERE, "libkfx_oracle.so")
    src = os.path.join(_HERE, "kfx_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libkfx_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        PI, PV, PF = C.POINTER(KfoImage), C.POINTER(KfoVolume), C.POINTER(C.c_float)
        L.kfo_bilateral_f32.argtypes = [PI, PI, C.c_float, C.c_float, C.c_int, C.c_float, C.c_int, C.c_int]
        L.kfo_bilateral_u16.argtypes = [PI, PI, C.c_float, C.c_float, C.c_int, C.c_ushort, C.c_int]
        L.kfo_bilateral_u8.argtypes = [PI, PI, C.c_float, C.c_float, C.c_int, C.c_int]
        L.kfo_depth_to_vbo_f32.argtypes = [PI, PI, PF, C.c_float]
        L.kfo_depth_to_vbo_u16.argtypes = [PI, PI, PF, C.c_float]
        L.kfo_normals_from_vbo.argtypes = [PI, PI]
        L.kfo_sdf_reset.argtypes = [PV, C.c_float]
        L.kfo_sdf_sphere.argtypes = [PV, PF, C.c_float]
        L.kfo_sdf_fuse.argtypes = [PV, PI, PI, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int]
        L.kfo_sdf_fuse.restype = C.c_uint64
        L.kfo_raycast_sdf.argtypes = [PI, PI, PI, PV, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int,
                                      C.c_int, C.POINTER(KfoRaycastStats)]
        L.kfo_raycast_sdf_touch.argtypes = [PI, PI, PI, PV, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int,
                                            C.c_void_p, C.POINTER(KfoRaycastStats)]
        L.kfo_raycast_box.argtypes = [PI, PF, PF, PF, PF]
        L.kfo_raycast_sphere.argtypes = [PI, PI, PF, PF, PF, C.c_float]
        L.kfo_raycast_plane.argtypes = [PI, PI, PF, PF, PF]
        L.kfo_render_scene.argtypes = [PI, C.c_int, PF, PF]
        L.kfo_fit_to_frustum.argtypes = [PF, PF, PF, C.c_float, C.c_float, PF, C.c_float, C.c_float]
        L.kfo_sub_bounding_volume.argtypes = [PV, PV, PF, PF]
        L.kfo_se3_inverse.argtypes = [PF, PF]
        L.kfo_sdf_fuse_slab.argtypes = [PV, C.POINTER(KfoSlab), PI, PI, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int]
        L.kfo_sdf_fuse_slab.restype = C.c_uint64
        L.kfo_raycast_sdf_slab.argtypes = [C.c_void_p, C.c_int, PV, C.POINTER(KfoSlab), C.c_int, C.c_int, C.c_int, C.c_int, PF, PF,
                                           C.c_float, C.c_float, C.c_float, C.c_int]
        L.kfo_raycast_sdf_slab.restype = None
        L.kfo_color_reset.argtypes = [PV]
        L.kfo_color_reset.restype = None
        L.kfo_sdf_fuse_color.argtypes = [PV, PV, PI, PI, PF, PF, PI, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int]
        L.kfo_sdf_fuse_color.restype = C.c_uint64
        L.kfo_raycast_sdf_color.argtypes = [PI, PI, PI, PV, PV, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int]
        L.kfo_raycast_sdf_color.restype = None
        L.kfo_texture_depth.argtypes = [PI, C.POINTER(KfoKeyframe), C.c_int, PI, PI, PI, PF, PF]
        L.kfo_texture_depth.restype = None
        L.kfo_bilateral_guided.argtypes = [PI, PI, PI, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int]
        L.kfo_bilateral_guided.restype = None
        L.kfo_disp2depth.argtypes = [PI, PI, C.c_float, C.c_float, C.c_float]
        L.kfo_disp2depth.restype = None
        L.kfo_filter_bad_kinect.argtypes = [PI, PI, C.c_int]
        L.kfo_filter_bad_kinect.restype = None
        L.kfo_colour_vbo.argtypes = [PI, PI, PI, PF]
        L.kfo_colour_vbo.restype = None
        L.kfo_sdf_distance.argtypes = [PI, PI, PV, PF, PF]
        L.kfo_sdf_distance.restype = None
        L.kfo_marching_cubes.argtypes = [PV, PV, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.kfo_marching_cubes.restype = C.c_uint64
        L.kfo_icp_point_plane.argtypes = [PI, PI, PI, PF, PF, C.c_float, PI, C.POINTER(KfoLss6), C.c_void_p]
        L.kfo_icp_point_plane.restype = None
        L.kfo_icp_block_dims.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_uint)]
        L.kfo_icp_block_dims.restype = None
        L.kfo_sdf_fuse_h.argtypes = L.kfo_sdf_fuse.argtypes
        L.kfo_sdf_fuse_h.restype = C.c_uint64
        L.kfo_raycast_sdf_h.argtypes = L.kfo_raycast_sdf.argtypes
        L.kfo_raycast_sdf_h.restype = None
        L.kfo_sdf_reset_h.argtypes = [PV, C.c_float]
        L.kfo_sdf_reset_h.restype = None
        L.kfo_sdf_sphere_h.argtypes = [PV, PF, C.c_float]
        L.kfo_sdf_sphere_h.restype = None
        L.kfo_elementwise_scale_bias_f32.argtypes = [PI, PI, C.c_float, C.c_float]
        L.kfo_elementwise_scale_bias_f32.restype = None
        L.kfo_box_half_ignore_invalid_f32.argtypes = [PI, PI]
        L.kfo_box_half_ignore_invalid_f32.restype = None
        L.kfo_max_threads.restype = C.c_int
        L.kfo_trilinear.argtypes = [PV, PF]
        L.kfo_trilinear.restype = C.c_float
        L.kfo_gradient.argtypes = [PV, PF, PF]
        L.kfo_gradient.restype = None
        L.kfo_sdf_accumulate.argtypes = [C.c_float] * 5 + [PF]
        L.kfo_sdf_accumulate.restype = None
        L.kfo_phong_shade.argtypes = [PF, PF]
        L.kfo_phong_shade.restype = C.c_float
        L.kfo_intrinsics_level.argtypes = [PF, PF, C.c_int]
        L.kfo_intrinsics_level.restype = None
        L.kfo_voxel_position.argtypes = [PV, C.c_int, C.c_int, C.c_int, PF]
        L.kfo_voxel_position.restype = None
        for f in ("kfo_bilateral_f32", "kfo_bilateral_u16", "kfo_bilateral_u8", "kfo_depth_to_vbo_f32",
                  "kfo_depth_to_vbo_u16", "kfo_normals_from_vbo", "kfo_sdf_reset", "kfo_sdf_sphere",
                  "kfo_raycast_sdf", "kfo_raycast_sdf_touch", "kfo_raycast_box", "kfo_raycast_sphere",
                  "kfo_raycast_plane", "kfo_render_scene", "kfo_fit_to_frustum",
                  "kfo_sub_bounding_volume", "kfo_se3_inverse"):
            getattr(L, f).restype = None
        _LIB = L
    return _LIB


def _fp(a):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


class Image:
    """Pitched host image backed by a numpy array.  `data` has shape (h, w[, ch])."""

    def __init__(self, w, h, dtype=np.float32, channels=1, pitch_bytes=None):
        self.w, self.h, self.channels = int(w), int(h), int(channels)
        self.dtype = np.dtype(dtype)
        elem = self.dtype.itemsize * self.channels
        self.pitch = int(pitch_bytes) if pitch_bytes else self.w * elem
        assert self.pitch >= self.w * elem and self.pitch % self.dtype.itemsize == 0
        self.raw = np.zeros((self.h, self.pitch), dtype=np.uint8)
        row = self.raw[:, : self.w * elem].view(self.dtype)
        # view into the pitched storage
        self.data = np.lib.stride_tricks.as_strided(
            self.raw.view(self.dtype).reshape(-1),
            shape=(self.h, self.w, self.channels) if channels > 1 else (self.h, self.w),
            strides=(self.pitch, elem, self.dtype.itemsize) if channels > 1 else (self.pitch, elem),
        )
        del row

    @staticmethod
    def from_numpy(a, pitch_bytes=None):
        a = np.asarray(a)
        ch = a.shape[2] if a.ndim == 3 else 1
        im = Image(a.shape[1], a.shape[0], a.dtype, ch, pitch_bytes)
        im.data[...] = a
        return im

    def struct(self):
        return KfoImage(self.pitch, self.raw.ctypes.data, self.w, self.h)

    def ref(self):
        self._s = self.struct()
        return C.byref(self._s)


class Volume:
    """Pitched host BoundedVolume<SDF_t>.  `data` has shape (d, h, w, 2) = (val, w)."""

    def __init__(self, w, h, d, boxmin=(-1, -1, -1), boxmax=(1, 1, 1), pitch_bytes=None, elem_floats=2):
        self.w, self.h, self.d = int(w), int(h), int(d)
        self.ef = elem_floats
        elem = 4 * elem_floats
        self.pitch = int(pitch_bytes) if pitch_bytes else self.w * elem
        self.img_pitch = self.pitch * self.h
        self.raw = np.zeros(self.img_pitch * self.d, dtype=np.uint8)
        self.boxmin = np.asarray(boxmin, np.float32)
        self.boxmax = np.asarray(boxmax, np.float32)
        self.data = np.lib.stride_tricks.as_strided(
            self.raw.view(np.float32), shape=(self.d, self.h, self.w, elem_floats),
            strides=(self.img_pitch, self.pitch, elem, 4))

    def struct(self):
        s = KfoVolume(self.pitch, self.raw.ctypes.data, self.w, self.h, self.img_pitch, self.d)
        for i in range(3):
            s.boxmin[i] = float(self.boxmin[i])
            s.boxmax[i] = float(self.boxmax[i])
        return s

    def ref(self):
        self._s = self.struct()
        return C.byref(self._s)

    def voxel_size(self):
        sz = (self.boxmax - self.boxmin).astype(np.float32)
        return sz / np.array([self.w - 1, self.h - 1, self.d - 1], np.float32)


class VolumeH:
    """Pitched host BoundedVolume<SDF_h>: cells {half val; half w;}.  `data` is float16 (d, h, w, 2)."""
    half = True

    def __init__(self, w, h, d, boxmin=(-1, -1, -1), boxmax=(1, 1, 1), pitch_bytes=None):
        self.w, self.h, self.d = int(w), int(h), int(d)
        self.pitch = int(pitch_bytes) if pitch_bytes else self.w * 4
        self.img_pitch = self.pitch * self.h
        self.raw = np.zeros(self.img_pitch * self.d, dtype=np.uint8)
        self.boxmin = np.asarray(boxmin, np.float32)
        self.boxmax = np.asarray(boxmax, np.float32)
        self.data = np.lib.stride_tricks.as_strided(
            self.raw.view(np.float16), shape=(self.d, self.h, self.w, 2), strides=(self.img_pitch, self.pitch, 4, 2))

    struct = Volume.struct
    ref = Volume.ref
    voxel_size = Volume.voxel_size


class SubVolume:
    """Non-owning view produced by sub_bounding_volume (keeps the parent alive)."""

    def __init__(self, parent, s):
        self.parent, self._s = parent, s
        self.w, self.h, self.d = s.w, s.h, s.d
        self.pitch, self.img_pitch = s.pitch, s.img_pitch
        self.boxmin = np.array(list(s.boxmin), np.float32)
        self.boxmax = np.array(list(s.boxmax), np.float32)
        off = s.ptr - parent.raw.ctypes.data
        self.offset_bytes = off
        z0, rem = divmod(off, s.img_pitch)
        y0, rem = divmod(rem, s.pitch)
        x0 = rem // 8
        self.origin = (x0, y0, z0)
        self.data = parent.data[z0:z0 + s.d, y0:y0 + s.h, x0:x0 + s.w]

    def struct(self):
        return self._s

    def ref(self):
        return C.byref(self._s)


# ---- thin functional API -----------------------------------------------------
def bilateral(out, inp, gs, gr, size, minval=None, nthreads=1):
    L = lib()
    if inp.dtype == np.float32:
        L.kfo_bilateral_f32(out.ref(), inp.ref(), gs, gr, size, 0.0 if minval is None else minval,
                            0 if minval is None else 1, nthreads)
    elif inp.dtype == np.uint16:
        L.kfo_bilateral_u16(out.ref(), inp.ref(), gs, gr, size, int(minval), nthreads)
    elif inp.dtype == np.uint8:
        L.kfo_bilateral_u8(out.ref(), inp.ref(), gs, gr, size, nthreads)
    else:
        raise TypeError(inp.dtype)


def depth_to_vbo(vbo, depth, K, scale=1.0):
    _, k = _fp(K)
    if depth.dtype == np.uint16:
        lib().kfo_depth_to_vbo_u16(vbo.ref(), depth.ref(), k, scale)
    else:
        lib().kfo_depth_to_vbo_f32(vbo.ref(), depth.ref(), k, scale)


def normals_from_vbo(nrm, vbo):
    lib().kfo_normals_from_vbo(nrm.ref(), vbo.ref())


def _is_half(vol):
    return bool(getattr(vol, "half", False))


def sdf_reset(vol, trunc):
    (lib().kfo_sdf_reset_h if _is_half(vol) else lib().kfo_sdf_reset)(vol.ref(), trunc)


def sdf_sphere(vol, center, r):
    _, c = _fp(center)
    (lib().kfo_sdf_sphere_h if _is_half(vol) else lib().kfo_sdf_sphere)(vol.ref(), c, r)


def sdf_fuse(vol, depth, norm, T_cw, K, trunc, max_w, mincostheta, full_extent=False, nthreads=1, slab=None):
    """slab = (full_d, z_offset, full_zmin, full_zmax): integrate `vol` as a Z-slab of a larger volume."""
    _, t = _fp(T_cw)
    _, k = _fp(K)
    if slab is not None:
        sl = KfoSlab(int(slab[0]), int(slab[1]), float(slab[2]), float(slab[3]))
        return int(lib().kfo_sdf_fuse_slab(vol.ref(), C.byref(sl), depth.ref(), norm.ref(), t, k, trunc, max_w, mincostheta,
                                           2 if full_extent == "slab" else (1 if full_extent else 0), nthreads))
    fn = lib().kfo_sdf_fuse_h if _is_half(vol) else lib().kfo_sdf_fuse
    return int(fn(vol.ref(), depth.ref(), norm.ref(), t, k, trunc, max_w, mincostheta, 1 if full_extent else 0, nthreads))


def raycast_sdf(depth, norm, img, vol, T_wc, K, near, far, trunc, subpix=True, nthreads=1):
    _, t = _fp(T_wc)
    _, k = _fp(K)
    st = KfoRaycastStats()
    fn = lib().kfo_raycast_sdf_h if _is_half(vol) else lib().kfo_raycast_sdf
    fn(depth.ref(), norm.ref(), img.ref(), vol.ref(), t, k, near, far, trunc, 1 if subpix else 0, nthreads, C.byref(st))
    return {"rays": st.rays, "steps": st.steps, "hits": st.hits}


def raycast_sdf_touch(depth, norm, img, vol, T_wc, K, near, far, trunc, subpix=True):
    """Returns (stats, number of distinct voxels touched)."""
    _, t = _fp(T_wc)
    _, k = _fp(K)
    st = KfoRaycastStats()
    bm = np.zeros((vol.w * vol.h * vol.d + 7) // 8, np.uint8)
    lib().kfo_raycast_sdf_touch(depth.ref(), norm.ref(), img.ref(), vol.ref(), t, k, near, far, trunc,
                                1 if subpix else 0, bm.ctypes.data, C.byref(st))
    touched = int(np.unpackbits(bm).sum())
    return {"rays": st.rays, "steps": st.steps, "hits": st.hits}, touched


def raycast_box(depth, T_wc, K, boxmin, boxmax):
    _, t = _fp(T_wc)
    _, k = _fp(K)
    _, a = _fp(boxmin)
    _, b = _fp(boxmax)
    lib().kfo_raycast_box(depth.ref(), t, k, a, b)


def raycast_sphere(depth, img, T_wc, K, center, r):
    _, t = _fp(T_wc)
    _, k = _fp(K)
    _, c = _fp(center)
    null = KfoImage(0, None, 0, 0)
    lib().kfo_raycast_sphere(depth.ref(), img.ref() if img is not None else C.byref(null), t, k, c, r)


def raycast_plane(depth, img, T_wc, K, n_w):
    _, t = _fp(T_wc)
    _, k = _fp(K)
    _, n = _fp(n_w)
    null = KfoImage(0, None, 0, 0)
    lib().kfo_raycast_plane(depth.ref(), img.ref() if img is not None else C.byref(null), t, k, n)


def render_scene(depth, scene, T_wc, K):
    _, t = _fp(T_wc)
    _, k = _fp(K)
    lib().kfo_render_scene(depth.ref(), {"room": 0, "full": 1}[scene], t, k)


def fit_to_frustum(T_wc, w, h, K, near, far):
    _, t = _fp(T_wc)
    _, k = _fp(K)
    lo = (C.c_float * 3)()
    hi = (C.c_float * 3)()
    lib().kfo_fit_to_frustum(lo, hi, t, float(w), float(h), k, near, far)
    return np.array(list(lo), np.float32), np.array(list(hi), np.float32)


def sub_bounding_volume(vol, rmin, rmax):
    _, a = _fp(rmin)
    _, b = _fp(rmax)
    out = KfoVolume()
    lib().kfo_sub_bounding_volume(C.byref(out), vol.ref(), a, b)
    return SubVolume(vol, out)


def se3_inverse(T):
    _, t = _fp(T)
    o = (C.c_float * 12)()
    lib().kfo_se3_inverse(o, t)
    return np.array(list(o), np.float32).reshape(3, 4)


def max_threads():
    return int(lib().kfo_max_threads())


def trilinear(vol, pos):
    _, q = _fp(pos)
    return float(lib().kfo_trilinear(vol.ref(), q))


def gradient(vol, pos):
    _, q = _fp(pos)
    o = (C.c_float * 3)()
    lib().kfo_gradient(vol.ref(), q, o)
    return np.array(list(o), np.float32)


def sdf_accumulate(val, w, old_val, old_w, max_w):
    o = (C.c_float * 2)()
    lib().kfo_sdf_accumulate(val, w, old_val, old_w, max_w, o)
    return np.array(list(o), np.float32)


def phong_shade(p_c, n_c):
    p, n = _fp(p_c), _fp(n_c)
    return np.float32(lib().kfo_phong_shade(p[1], n[1]))


def intrinsics_level(K, level):
    _, k = _fp(K)
    o = (C.c_float * 4)()
    lib().kfo_intrinsics_level(o, k, level)
    return np.array(list(o), np.float32)


def voxel_position(vol, x, y, z):
    o = (C.c_float * 3)()
    lib().kfo_voxel_position(vol.ref(), x, y, z, o)
    return np.array(list(o), np.float32)


def elementwise_scale_bias(b, a, s, offset=0.0):
    lib().kfo_elementwise_scale_bias_f32(b.ref(), a.ref(), s, offset)


def box_half_ignore_invalid(out, inp):
    lib().kfo_box_half_ignore_invalid_f32(out.ref(), inp.ref())


def raycast_sdf_slab(state, init, vol, slab, own_lo, own_hi, w, h, T_wc, K, near, far, trunc, subpix=True):
    """One round of the exact multi-GPU march; `state` is a C-contiguous float32 array (9, h, w)."""
    assert state.dtype == np.float32 and state.flags["C_CONTIGUOUS"] and state.shape == (9, h, w)
    _, t = _fp(T_wc)
    _, k = _fp(K)
    sl = KfoSlab(int(slab[0]), int(slab[1]), float(slab[2]), float(slab[3]))
    lib().kfo_raycast_sdf_slab(state.ctypes.data, 1 if init else 0, vol.ref(), C.byref(sl), own_lo, own_hi, w, h, t, k,
                               near, far, trunc, 1 if subpix else 0)


LSS_DTYPE = np.dtype([("JTy", np.float32, 6), ("JTJ", np.float32, 21), ("sqErr", np.float32), ("obs", np.uint32)])


def icp_block_dims(w, h):
    out = (C.c_uint * 4)()
    lib().kfo_icp_block_dims(w, h, out)
    return tuple(out)


def icp_point_plane(Pl, Pr, Nr, KT_lr, T_rl, c, debug=None, want_blocks=False, fn=None):
    """PoseRefinementProjectiveIcpPointPlane: returns the summed system as a LSS_DTYPE scalar (and the
    per-block systems when want_blocks).  `fn` swaps in another implementation with the same C signature
    (the reference-header harness)."""
    _, kt = _fp(KT_lr)
    _, t = _fp(T_rl)
    out = KfoLss6()
    bx, by, gx, gy = icp_block_dims(Pl.w, Pl.h)
    blocks = np.zeros(gx * gy, LSS_DTYPE) if want_blocks else None
    f = lib().kfo_icp_point_plane if fn is None else fn
    f(Pl.ref(), Pr.ref(), Nr.ref(), kt, t, C.c_float(c), debug.ref() if debug is not None else None, C.byref(out),
      blocks.ctypes.data if blocks is not None else None)
    res = np.frombuffer(bytes(out), LSS_DTYPE)[0]
    return (res, blocks) if want_blocks else res


def ColorVolume(w, h, d, boxmin=(-1, -1, -1), boxmax=(1, 1, 1), pitch_bytes=None):
    """BoundedVolume<float>: grey-level colour volume, `data` shape (d, h, w, 1)."""
    return Volume(w, h, d, boxmin, boxmax, pitch_bytes=pitch_bytes, elem_floats=1)


def color_reset(cvol):
    lib().kfo_color_reset(cvol.ref())


def sdf_fuse_color(vol, cvol, depth, norm, T_cw, K, img, T_iw, Kimg, trunc, max_w, mincostheta, full_extent=False, nthreads=1):
    _, t = _fp(T_cw)
    _, k = _fp(K)
    _, ti = _fp(T_iw)
    _, ki = _fp(Kimg)
    return int(lib().kfo_sdf_fuse_color(vol.ref(), cvol.ref(), depth.ref(), norm.ref(), t, k, img.ref(), ti, ki, trunc, max_w,
                                        mincostheta, 1 if full_extent else 0, nthreads))


def raycast_sdf_color(depth, norm, img, vol, cvol, T_wc, K, near, far, trunc, subpix=True, nthreads=1):
    _, t = _fp(T_wc)
    _, k = _fp(K)
    lib().kfo_raycast_sdf_color(depth.ref(), norm.ref(), img.ref(), vol.ref(), cvol.ref(), t, k, near, far, trunc,
                                1 if subpix else 0, nthreads)


def marching_cubes(vol, cvol, ntris, emask, tris):
    """SaveMesh's extraction with the given case tables (ntris[256] u8, emask[256] u16, tris[256, k] i8).
    Returns (verts (3T,3), norms (3T,3), colors (3T,4) or None)."""
    ntris = np.ascontiguousarray(ntris, np.uint8)
    emask = np.ascontiguousarray(emask, np.uint16)
    tris = np.ascontiguousarray(tris, np.int8)
    L = lib()
    cref = cvol.ref() if cvol is not None else None
    nt = int(L.kfo_marching_cubes(vol.ref(), cref, ntris.ctypes.data, emask.ctypes.data, tris.ctypes.data, tris.shape[1], None, None, None))
    verts, norms = np.zeros((3 * nt, 3), np.float32), np.zeros((3 * nt, 3), np.float32)
    colors = np.zeros((3 * nt, 4), np.float32) if cvol is not None else None
    L.kfo_marching_cubes(vol.ref(), cref, ntris.ctypes.data, emask.ctypes.data, tris.ctypes.data, tris.shape[1], verts.ctypes.data,
                         norms.ctypes.data, colors.ctypes.data if colors is not None else None)
    return verts, norms, colors


def sdf_distance(dist, depth, vol, T_wc, K):
    _, t = _fp(T_wc)
    _, k = _fp(K)
    lib().kfo_sdf_distance(dist.ref(), depth.ref(), vol.ref(), t, k)


def disp2depth(inp, out, fu, baseline, min_disp=0.0):
    lib().kfo_disp2depth(inp.ref(), out.ref(), fu, baseline, min_disp)


def filter_bad_kinect(out, inp):
    lib().kfo_filter_bad_kinect(out.ref(), inp.ref(), 1 if inp.dtype == np.uint16 else 0)


def colour_vbo(idimg, vbo, rgb, KT_cd):
    _, t = _fp(KT_cd)
    lib().kfo_colour_vbo(idimg.ref(), vbo.ref(), rgb.ref(), t)


def bilateral_guided(out, inp, guide, gs, gr, gc, size):
    lib().kfo_bilateral_guided(out.ref(), inp.ref(), guide.ref(), 1 if guide.dtype == np.uint8 else 0, gs, gr, gc, size)


def texture_depth(out, keyframes, depth, norm, T_wd, Kdepth, phong=None, fn=None):
    """keyframes: list of (Image uint8 x3 or None, T_iw, K)."""
    n = len(keyframes)
    arr = (KfoKeyframe * max(n, 1))()
    for i, (kimg, T_iw, K) in enumerate(keyframes):
        for j, v in enumerate(np.asarray(K, np.float32).reshape(-1)):
            arr[i].K[j] = float(v)
        for j, v in enumerate(np.asarray(T_iw, np.float32).reshape(-1)):
            arr[i].T_iw[j] = float(v)
        arr[i].img = kimg.struct() if kimg is not None else KfoImage(0, None, 0, 0)
    _, t = _fp(T_wd)
    _, k = _fp(Kdepth)
    f = lib().kfo_texture_depth if fn is None else fn
    f(out.ref(), arr, n, depth.ref(), norm.ref(), phong.ref() if phong is not None else None, t, k)
