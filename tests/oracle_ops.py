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

    def SubImage(self, x, y, w, h):
        """Top-left sub-rectangle sharing the parent's storage (only x = y = 0 is needed by the pipelines)."""
        assert x == 0 and y == 0 and w <= self.w and h <= self.h
        sub = Image.__new__(Image)
        sub.__dict__.update(self.__dict__)
        sub.w, sub.h = int(w), int(h)
        sub.data = self.data[:h, :w]
        return sub

    def MemcpyToHost(self):
        return self.data.copy()


class BoundedVolume(oracle.Volume):
    def __init__(self, w, h, d, boxmin=(-1, -1, -1), boxmax=(1, 1, 1), device=None, pitch=None):
        super().__init__(w, h, d, boxmin, boxmax, pitch_bytes=pitch)

    def MemcpyToHost(self):
        return self.data.copy()

    def planes(self, z0, z1):
        return torch.from_numpy(self.raw)[z0 * self.img_pitch: z1 * self.img_pitch]

    def ZSlab(self, z0, z1):
        """Same semantics as kangaroo_amd.roo.BoundedVolume.ZSlab (view + bbox of first/last plane)."""
        f = np.float32
        s = (self.boxmax - self.boxmin).astype(f)

        def pos(x, y, z):
            return np.array([self.boxmin[0] + s[0] * f(x) / f(self.w - 1), self.boxmin[1] + s[1] * f(y) / f(self.h - 1),
                             self.boxmin[2] + s[2] * f(z) / f(self.d - 1)], f)
        return _SlabView(self, z0, z1, pos(0, 0, z0), pos(self.w - 1, self.h - 1, z1 - 1))


class _SlabView:
    def __init__(self, parent, z0, z1, lo, hi):
        self.parent, self.w, self.h, self.d = parent, parent.w, parent.h, z1 - z0
        self.boxmin, self.boxmax = lo, hi
        st = oracle.KfoVolume(parent.pitch, parent.raw.ctypes.data + z0 * parent.img_pitch, parent.w, parent.h,
                              parent.img_pitch, z1 - z0)
        for i in range(3):
            st.boxmin[i] = float(lo[i])
            st.boxmax[i] = float(hi[i])
        self._s = st

    def ref(self):
        import ctypes
        return ctypes.byref(self._s)


def SdfReset(vol, trunc):
    oracle.sdf_reset(vol, trunc)


def BilateralFilter(out, inp, gs, gr, size, minval=None):
    oracle.bilateral(out, inp, gs, gr, size, minval)


def DepthToVbo(vbo, depth, K, scale=1.0):
    oracle.depth_to_vbo(vbo, depth, K, scale)


def NormalsFromVbo(n, v):
    oracle.normals_from_vbo(n, v)


def SdfFuse(vol, depth, norm, T_cw, K, trunc, maxw, mincostheta, full_extent=False, slab=None):
    oracle.sdf_fuse(vol, depth, norm, T_cw, K, trunc, maxw, mincostheta, full_extent=full_extent, slab=slab)


def RaycastSdf(depth, norm, img, vol, T_wc, K, near, far, trunc, subpix=True):
    oracle.raycast_sdf(depth, norm, img, vol, T_wc, K, near, far, trunc, subpix)


def RaycastSdfSlab(state, init, vol, slab, own_lo, own_hi, w, h, T_wc, K, near, far, trunc, subpix=True):
    oracle.raycast_sdf_slab(state.numpy(), init, vol, slab, own_lo, own_hi, w, h, T_wc, K, near, far, trunc, subpix)


def RaycastStateToImages(depth, norm, img, state):
    st = state.numpy()
    hit = (st[3] == 1.0) & (st[0] > 0)
    depth.data[...] = np.where(hit, st[0], np.float32("nan"))
    img.data[...] = np.where(hit, st[8], np.float32(0))
    n = np.stack([st[5], st[6], st[7], np.ones_like(st[0])], axis=-1)
    norm.data[...] = np.where(hit[..., None], n, np.float32(0))


class LeastSquaresSystem:
    def __init__(self, rec):
        self.JTy = np.array(rec["JTy"], np.float32)
        self.raw = np.array(rec["JTJ"], np.float32)
        self.JTJ = np.zeros((6, 6), np.float32)
        i = 0
        for r in range(6):
            for c in range(r + 1):
                self.JTJ[r, c] = self.JTJ[c, r] = self.raw[i]
                i += 1
        self.sqErr = np.float32(rec["sqErr"])
        self.obs = int(rec["obs"])


def PoseRefinementProjectiveIcpPointPlane(dPl, dPr, dNr, KT_lr, T_rl, c, dWorkspace=None, dDebug=None):
    return LeastSquaresSystem(oracle.icp_point_plane(dPl, dPr, dNr, KT_lr, T_rl, c, dDebug))


class Pyramid:
    def __init__(self, w, h, levels, kind="f32"):
        self.imgs = [Image(w >> l, h >> l, kind) for l in range(levels) if (w >> l) > 0 and (h >> l) > 0]

    def __getitem__(self, l):
        return self.imgs[l]

    def __len__(self):
        return len(self.imgs)


def BoxHalfIgnoreInvalid(out, inp):
    oracle.box_half_ignore_invalid(out, inp)


def BoxReduceIgnoreInvalid(pyramid):
    for l in range(1, len(pyramid)):
        BoxHalfIgnoreInvalid(pyramid[l], pyramid[l - 1])
