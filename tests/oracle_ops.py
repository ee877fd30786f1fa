"""Oracle-backed stand-in for `kangaroo_amd.roo`, for CPU tests of the host-side pipeline logic
(slab partitioning, compositing) under gloo.  Lives in tests/: the product never imports it."""
import numpy as np
import torch

import oracle

_KIND = {"f32": (np.float32, 1), "f32x4": (np.float32, 4), "u16": (np.uint16, 1), "u8": (np.uint8, 1)}


class Image(oracle.Image):
    def __init__(self, w, h, kind="f32", pitch=None, device=None):
        dt, ch = _KIND[kind]
        super().__init__(w, h, dt, ch, pitch_bytes=pitch)
        self.kind = kind

    def tensor(self):
        return torch.from_numpy(self.data)

    def MemcpyFromHost(self, arr):
        self.data[...] = arr
        return self

    def MemcpyToHost(self):
        return self.data.copy()


class BoundedVolume(oracle.Volume):
    def __init__(self, w, h, d, boxmin=(-1, -1, -1), boxmax=(1, 1, 1), device=None, pitch=None):
        super().__init__(w, h, d, boxmin, boxmax, pitch_bytes=pitch)

    def MemcpyToHost(self):
        return self.data.copy()


def SdfReset(vol, trunc):
    oracle.sdf_reset(vol, trunc)


def BilateralFilter(out, inp, gs, gr, size, minval=None):
    oracle.bilateral(out, inp, gs, gr, size, minval)


def DepthToVbo(vbo, depth, K, scale=1.0):
    oracle.depth_to_vbo(vbo, depth, K, scale)


def NormalsFromVbo(n, v):
    oracle.normals_from_vbo(n, v)


def SdfFuse(vol, depth, norm, T_cw, K, trunc, maxw, mincostheta, full_extent=False):
    oracle.sdf_fuse(vol, depth, norm, T_cw, K, trunc, maxw, mincostheta, full_extent=full_extent)


def RaycastSdf(depth, norm, img, vol, T_wc, K, near, far, trunc, subpix=True):
    oracle.raycast_sdf(depth, norm, img, vol, T_wc, K, near, far, trunc, subpix)
