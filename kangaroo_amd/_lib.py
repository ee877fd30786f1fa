"""ctypes loader for libkfx.so (the gfx950 HIP kernels behind the C ABI of include/kfx.h).

There is no CPU fallback: if the shared library is missing or fails to load, importing the
ops raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C kangaroo_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KFX_LIB_PATH: load another build of the same library (A/B of compiler flags in scripts/); default = the in-tree build
LIB_PATH = os.environ.get("KFX_LIB_PATH") or os.path.join(_HERE, "libkfx.so")


class KfxImage(C.Structure):
    """kfx_image == roo::Image<T> (include/kfx.h)."""
    _fields_ = [("pitch", C.c_size_t), ("ptr", C.c_void_p), ("w", C.c_size_t), ("h", C.c_size_t)]


class KfxVolume(C.Structure):
    """kfx_volume == roo::BoundedVolume<T> (include/kfx.h)."""
    _fields_ = [("pitch", C.c_size_t), ("ptr", C.c_void_p), ("w", C.c_size_t), ("h", C.c_size_t),
                ("img_pitch", C.c_size_t), ("d", C.c_size_t),
                ("boxmin", C.c_float * 3), ("boxmax", C.c_float * 3)]


class KfxLss6(C.Structure):
    """roo::LeastSquaresSystem<float,6> (Mat.h:483-520)."""
    _fields_ = [("JTy", C.c_float * 6), ("JTJ", C.c_float * 21), ("sqErr", C.c_float), ("obs", C.c_uint)]


class KfxIcpLevel(C.Structure):
    """kfx_icp_level (include/kfx.h): one pyramid level of kfx_icp_refine."""
    _fields_ = [("Pl", KfxImage), ("Pr", KfxImage), ("Nr", KfxImage), ("K", C.c_float * 4), ("iterations", C.c_int), ("rotation_only", C.c_int)]


class KfxKeyframe(C.Structure):
    """kfx_keyframe (include/kfx.h) = roo::ImageKeyframe<uchar3>."""
    _fields_ = [("K", C.c_float * 4), ("T_iw", C.c_float * 12), ("img", KfxImage)]


class KfxSlab(C.Structure):
    """kfx_slab (include/kfx.h): Z-slab of a larger volume."""
    _fields_ = [("full_d", C.c_size_t), ("z_offset", C.c_size_t), ("full_zmin", C.c_float), ("full_zmax", C.c_float)]


class KfxFrameConfig(C.Structure):
    """kfx_frame_config (include/kfx.h): the views and parameters of one frame of the application's loop."""
    _fields_ = [("vol", KfxVolume), ("raw", KfxImage), ("filtered", KfxImage), ("vbo", KfxImage), ("normals", KfxImage),
                ("ray_depth", KfxImage), ("ray_norm", KfxImage), ("ray_img", KfxImage), ("K", C.c_float * 4),
                ("bilateral_gs", C.c_float), ("bilateral_gr", C.c_float), ("bilateral_minval", C.c_float), ("bilateral_size", C.c_uint),
                ("near", C.c_float), ("far", C.c_float), ("trunc_dist", C.c_float), ("max_w", C.c_float), ("mincostheta", C.c_float),
                ("fuse_flags", C.c_uint), ("timing_slots", C.c_int)]


class KfxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libkfx error %d: %s" % (code, msg))
        self.code = code


PI, PV, PF = C.POINTER(KfxImage), C.POINTER(KfxVolume), C.POINTER(C.c_float)

# name -> (restype, argtypes); every symbol include/kfx.h declares
SIGNATURES = {
    "kfx_sdf_fuse": (C.c_int, [PV, PI, PI, PF, PF, C.c_float, C.c_float, C.c_float, C.c_uint, C.c_void_p]),
    "kfx_sdf_fuse_slab": (C.c_int, [PV, C.POINTER(KfxSlab), PI, PI, PF, PF, C.c_float, C.c_float, C.c_float, C.c_uint, C.c_void_p]),
    "kfx_sdf_fuse_slab_h": (C.c_int, [PV, C.POINTER(KfxSlab), PI, PI, PF, PF, C.c_float, C.c_float, C.c_float, C.c_uint, C.c_void_p]),
    "kfx_sdf_fuse_count": (C.c_int, [PV, PI, PI, PF, PF, C.c_float, C.c_float, C.c_uint, C.c_void_p, C.c_void_p]),
    "kfx_raycast_sdf_count": (C.c_int, [PV, C.c_uint, C.c_uint, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kfx_raycast_sdf_count_h": (C.c_int, [PV, C.c_uint, C.c_uint, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kfx_sdf_fuse_h": (C.c_int, [PV, PI, PI, PF, PF, C.c_float, C.c_float, C.c_float, C.c_uint, C.c_void_p]),
    "kfx_raycast_sdf_h": (C.c_int, [PI, PI, PI, PV, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "kfx_raycast_sdf_levels": (C.c_int, [C.c_int, C.POINTER(PI), C.POINTER(PI), C.POINTER(PI), C.POINTER(PI), PV, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "kfx_raycast_sdf_levels_h": (C.c_int, [C.c_int, C.POINTER(PI), C.POINTER(PI), C.POINTER(PI), C.POINTER(PI), PV, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "kfx_sdf_reset_h": (C.c_int, [PV, C.c_float, C.c_void_p]),
    "kfx_sdf_sphere_h": (C.c_int, [PV, PF, C.c_float, C.c_void_p]),
    "kfx_raycast_sdf": (C.c_int, [PI, PI, PI, PV, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "kfx_bilateral_f32": (C.c_int, [PI, PI, C.c_float, C.c_float, C.c_uint, C.c_float, C.c_int, C.c_void_p]),
    "kfx_bilateral_u16": (C.c_int, [PI, PI, C.c_float, C.c_float, C.c_uint, C.c_ushort, C.c_void_p]),
    "kfx_bilateral_u8": (C.c_int, [PI, PI, C.c_float, C.c_float, C.c_uint, C.c_void_p]),
    "kfx_depth_to_vbo_f32": (C.c_int, [PI, PI, PF, C.c_float, C.c_void_p]),
    "kfx_depth_to_vbo_u16": (C.c_int, [PI, PI, PF, C.c_float, C.c_void_p]),
    "kfx_normals_from_vbo": (C.c_int, [PI, PI, C.c_void_p]),
    "kfx_elementwise_scale_bias_f32": (C.c_int, [PI, PI, C.c_float, C.c_float, C.c_void_p]),
    "kfx_box_half_ignore_invalid_f32": (C.c_int, [PI, PI, C.c_void_p]),
    "kfx_sdf_reset": (C.c_int, [PV, C.c_float, C.c_void_p]),
    "kfx_sdf_sphere": (C.c_int, [PV, PF, C.c_float, C.c_void_p]),
    "kfx_raycast_sdf_slab": (C.c_int, [C.c_void_p, C.c_int, PV, C.POINTER(KfxSlab), C.c_int, C.c_int, C.c_int, C.c_int, PF, PF,
                                       C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "kfx_raycast_sdf_slab_h": (C.c_int, [C.c_void_p, C.c_int, PV, C.POINTER(KfxSlab), C.c_int, C.c_int, C.c_int, C.c_int, PF, PF,
                                       C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "kfx_raycast_sdf_slab_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                             PV, C.POINTER(KfxSlab), C.c_int, C.c_int, C.c_int, C.c_int, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "kfx_raycast_state_to_images": (C.c_int, [PI, PI, PI, C.c_void_p, C.c_void_p]),
    "kfx_sdf_fuse_color": (C.c_int, [PV, PV, PI, PI, PF, PF, PI, PF, PF, C.c_float, C.c_float, C.c_float, C.c_uint, C.c_void_p]),
    "kfx_raycast_sdf_color": (C.c_int, [PI, PI, PI, PV, PV, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "kfx_color_reset": (C.c_int, [PV, C.c_void_p]),
    "kfx_depth_to_vbo_normals_f32": (C.c_int, [PI, PI, PI, PF, C.c_float, C.c_void_p]),
    "kfx_depth_pyramid_vbo_normals_f32": (C.c_int, [PI, PI, PI, PF, C.c_int, C.c_float, C.c_void_p]),
    "kfx_bilateral_guided_f32": (C.c_int, [PI, PI, PI, C.c_float, C.c_float, C.c_float, C.c_uint, C.c_void_p]),
    "kfx_bilateral_guided_u8": (C.c_int, [PI, PI, PI, C.c_float, C.c_float, C.c_float, C.c_uint, C.c_void_p]),
    "kfx_texture_depth": (C.c_int, [PI, C.POINTER(KfxKeyframe), C.c_int, PI, PI, PI, PF, PF, C.c_void_p]),
    "kfx_disp2depth": (C.c_int, [PI, PI, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "kfx_filter_bad_kinect_f32": (C.c_int, [PI, PI, C.c_void_p]),
    "kfx_filter_bad_kinect_u16": (C.c_int, [PI, PI, C.c_void_p]),
    "kfx_colour_vbo": (C.c_int, [PI, PI, PI, PF, C.c_void_p]),
    "kfx_raycast_box": (C.c_int, [PI, PF, PF, PF, PF, C.c_void_p]),
    "kfx_raycast_sphere": (C.c_int, [PI, PI, PF, PF, PF, C.c_float, C.c_void_p]),
    "kfx_raycast_plane": (C.c_int, [PI, PI, PF, PF, PF, C.c_void_p]),
    "kfx_sdf_distance": (C.c_int, [PI, PI, PV, PF, PF, C.c_float, C.c_void_p]),
    "kfx_mc_count": (C.c_int, [PV, C.c_void_p, C.c_void_p]),
    "kfx_mc_emit": (C.c_int, [PV, PV, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kfx_icp_refine": (C.c_int, [C.POINTER(KfxIcpLevel), C.c_int, C.c_float, C.c_float, PI, PI, C.POINTER(C.c_double), PF,
                                 C.POINTER(C.c_uint), C.POINTER(C.c_int), C.c_void_p]),
    "kfx_pose_step": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), PF]),
    "kfx_icp_refine_then": (C.c_int, [C.POINTER(KfxIcpLevel), C.c_int, C.c_float, C.c_float, PI, PI, C.POINTER(C.c_double), PF,
                                      C.POINTER(C.c_uint), C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p]),   # (the hook: a CFUNCTYPE(None, c_void_p) cast to void*)
    "kfx_icp_point_plane": (C.c_int, [PI, PI, PI, PF, PF, C.c_float, PI, PI, C.POINTER(KfxLss6), C.c_void_p]),
    "kfx_composite_pack": (C.c_int, [PI, PI, PI, C.c_void_p, C.c_int, C.c_void_p]),
    "kfx_composite_select": (C.c_int, [PI, PI, PI, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "kfx_composite_unpack": (C.c_int, [PI, PI, PI, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kfx_composite_strip_pixels": (C.c_size_t, [C.c_size_t, C.c_size_t, C.c_int]),
    "kfx_composite_strips_pack": (C.c_int, [PI, PI, PI, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "kfx_composite_strips_merge": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]),
    "kfx_composite_strips_unpack": (C.c_int, [PI, PI, PI, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "kfx_alloc_pitched": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t]),
    "kfx_free": (C.c_int, [C.c_void_p]),
    "kfx_alloc_host": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "kfx_free_host": (C.c_int, [C.c_void_p]),
    "kfx_memcpy_2d": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]),
    "kfx_stream_synchronize": (C.c_int, [C.c_void_p]),
    "kfx_last_error_string": (C.c_char_p, []),
    "kfx_error_name": (C.c_char_p, [C.c_int]),
    "kfx_set_math_mode": (C.c_int, [C.c_int]),
    "kfx_get_math_mode": (C.c_int, []),
    "kfx_version": (C.c_int, []),
    "kfx_kernel_source_id": (C.c_char_p, [C.c_char_p]),
    "kfx_device_count": (C.c_int, []),
    "kfx_set_device": (C.c_int, [C.c_int]),
    "kfx_sdf_summary_create": (C.c_int, [C.POINTER(C.c_void_p), PV]),
    "kfx_sdf_summary_destroy": (C.c_int, [C.c_void_p]),
    "kfx_sdf_summary_invalidate": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kfx_sdf_reset_tracked": (C.c_int, [PV, C.c_void_p, C.c_float, C.c_void_p]),
    "kfx_sdf_fuse_tracked": (C.c_int, [PV, C.c_void_p, PI, PI, PF, PF, C.c_float, C.c_float, C.c_float, C.c_uint, C.c_void_p]),
    "kfx_raycast_sdf_tracked": (C.c_int, [PI, PI, PI, PV, C.c_void_p, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "kfx_sdf_summary_rebuild": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kfx_raycast_sdf_count_tracked": (C.c_int, [PV, C.c_void_p, C.c_uint, C.c_uint, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kfx_frame_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(KfxFrameConfig)]),
    "kfx_frame_destroy": (C.c_int, [C.c_void_p]),
    "kfx_frame_reset": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kfx_frame_set_track": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "kfx_frame_get_track": (C.c_int, [C.c_void_p]),
    "kfx_frame_summary": (C.c_void_p, [C.c_void_p]),
    "kfx_frame_count": (C.c_longlong, [C.c_void_p]),
    "kfx_frame_step": (C.c_int, [C.c_void_p, PI, PF, PF, C.c_uint, C.c_void_p]),
    "kfx_frame_timings": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, PF]),
    "kfx_frame_set_timing": (C.c_int, [C.c_void_p, C.c_uint]),
    "kfx_raycast_sdf_levels_tracked": (C.c_int, [C.c_int, C.POINTER(PI), C.POINTER(PI), C.POINTER(PI), C.POINTER(PI), PV, C.c_void_p, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]),
}

_lib = None


def load():
    """Load libkfx.so and bind every entry point; raises if anything is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "kangaroo_amd: %s not found -- the HIP extension is required (no CPU fallback). "
                "Run `make -C kangaroo_amd/csrc` or __graft_entry__.build()." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if the symbol is not exported
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


_dbg = None


def load_debug():
    """libkfx_debug.so: the measurement aids and arithmetic self-checks of include/kfx_debug.h (tests / scripts only; the
    product library libkfx.so does not contain them)."""
    global _dbg
    if _dbg is None:
        load()
        path = os.path.join(os.path.dirname(LIB_PATH), "libkfx_debug.so")
        if not os.path.exists(path):
            raise ImportError("kangaroo_amd: %s not found -- run `make -C kangaroo_amd/csrc`" % path)
        _dbg = C.CDLL(path)
    return _dbg


def check(code):
    if code != 0:
        raise KfxError(code, load().kfx_last_error_string().decode())
