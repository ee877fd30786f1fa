"""ctypes front end of include/kfx_slab.h: the transports behind kfx_comm and kfx_slab_frame, a slab rank's frame as ONE
library call (the N > 1 counterpart of roo.Frame / kfx_frame_step).  torch only supplies device memory and the stream.

    comm  = slab.Comm.rccl(rank, world, "/tmp/kfx.<launch>.id")      # one process per GPU: libkfx_rccl.so (RCCL over xGMI)
    comms = slab.Comm.threads(world)                                  # ranks = host threads sharing one GPU (tests)
    comm  = slab.Comm.torch(dist)                                     # collectives through torch.distributed (gloo tests)
    comm  = slab.Comm.loopback(rank, world)                           # one rank measured by itself (host-overhead floor)
    frame = slab.SlabFrame(comm, vol_local, layout, images..., K, ...)
    frame.step(T_wc, T_cw)
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import KfxImage, KfxVolume

V, Z = C.c_void_p, C.c_size_t


class KfxComm(C.Structure):
    """kfx_comm (include/kfx_slab.h): a table of transport functions."""


_P = C.POINTER(KfxComm)
KfxComm._fields_ = [("rank", C.c_int), ("world", C.c_int), ("impl", V),
                    ("all_reduce", C.CFUNCTYPE(C.c_int, _P, V, Z, C.c_int, V)),
                    ("exchange", C.CFUNCTYPE(C.c_int, _P, V, V, Z, V, V, Z, V)),
                    ("barrier", C.CFUNCTYPE(C.c_int, _P)),
                    ("destroy", C.CFUNCTYPE(None, _P)),
                    ("broadcast", C.CFUNCTYPE(C.c_int, _P, V, Z, C.c_int, V)),
                    ("all_to_all", C.CFUNCTYPE(C.c_int, _P, V, V, Z, V)),
                    ("all_gather", C.CFUNCTYPE(C.c_int, _P, V, V, Z, V)),
                    ("exchange_v", C.CFUNCTYPE(C.c_int, _P, V, Z, V, Z, V, Z, V, Z, V)),
                    ("flags", C.c_int),
                    ("dup", C.CFUNCTYPE(C.c_int, _P, _P))]
HOST_BLOCKING = 1   # KFX_COMM_HOST_BLOCKING
PIPE_MAX = 4        # KFX_SLAB_PIPE_MAX


class KfxSlabLayout(C.Structure):
    """kfx_slab_layout (include/kfx_slab.h)."""
    _fields_ = [("full_d", C.c_size_t), ("full_zmin", C.c_float), ("full_zmax", C.c_float), ("rank", C.c_int), ("world", C.c_int),
                ("ghost", C.c_int), ("z0", C.c_size_t), ("z1", C.c_size_t), ("s0", C.c_size_t), ("s1", C.c_size_t),
                ("local_zmin", C.c_float), ("local_zmax", C.c_float)]


class KfxSlabFrameConfig(C.Structure):
    """kfx_slab_frame_config (include/kfx_slab.h)."""
    _fields_ = [("local", KfxVolume), ("layout", KfxSlabLayout), ("raw", KfxImage), ("filtered", KfxImage), ("vbo", KfxImage), ("normals", KfxImage),
                ("ray_depth", KfxImage), ("ray_norm", KfxImage), ("ray_img", KfxImage), ("K", C.c_float * 4),
                ("bilateral_gs", C.c_float), ("bilateral_gr", C.c_float), ("bilateral_minval", C.c_float), ("bilateral_size", C.c_uint),
                ("near", C.c_float), ("far", C.c_float), ("trunc_dist", C.c_float), ("max_w", C.c_float), ("mincostheta", C.c_float),
                ("halo", C.c_int), ("raycast", C.c_int), ("merge", C.c_int), ("inputs", C.c_int), ("overlap", C.c_int), ("tiles", C.c_int),
                ("unchecked", C.c_int), ("timing_slots", C.c_int), ("pipe_depth", C.c_int), ("pipe_images", KfxImage * (3 * (PIPE_MAX - 1)))]


HALO = {"recompute": 0, "exchange": 1}
RAYCAST = {"exact": 0, "composite": 1}
MERGE = {"direct": 0, "allreduce": 1}
INPUTS = {"replicate": 0, "broadcast": 1}
TIMING_FIELDS = 6   # KFX_SLAB_FRAME_TIMING_FIELDS: preprocess, sdf_fuse (+ ghost planes), raycast, merge, frame, period

_bound = False


def _L():
    """libkfx.so with the kfx_slab.h entry points bound."""
    global _bound
    L = _lib.load()
    if not _bound:
        PF, PI, PV = _lib.PF, _lib.PI, _lib.PV
        PL = C.POINTER(KfxSlabLayout)
        sig = {
            "kfx_comm_create_threads": (C.c_int, [_P, C.c_int]),
            "kfx_comm_create_threads_p2p": (C.c_int, [_P, C.c_int, C.c_int]),
            "kfx_slab_frame_images": (C.c_int, [V, C.c_longlong, _lib.PI, _lib.PI, _lib.PI]),
            "kfx_comm_create_loopback": (C.c_int, [_P, C.c_int, C.c_int]),
            "kfx_slab_layout_init": (C.c_int, [PL, C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]),
            "kfx_slab_broadcast_inputs": (C.c_int, [PI, PI, V, C.c_int, _P, V]),
            "kfx_slab_exchange_halos": (C.c_int, [PV, PL, _P, V]),
            "kfx_slab_composite": (C.c_int, [PI, PI, PI, V, V, _P, V]),
            "kfx_slab_composite_direct_scratch_bytes": (C.c_size_t, [C.c_size_t, C.c_size_t, C.c_int]),
            "kfx_slab_composite_direct": (C.c_int, [PI, PI, PI, V, _P, V]),
            "kfx_slab_exact_scratch_bytes": (C.c_size_t, [C.c_size_t, C.c_size_t]),
            "kfx_slab_raycast_exact": (C.c_int, [PI, PI, PI, V, V, PV, PL, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, _P, V, C.POINTER(C.c_int)]),
            "kfx_slab_raycast_exact_allreduce": (C.c_int, [PI, PI, PI, V, V, PV, PL, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, _P, V, C.POINTER(C.c_int)]),
            "kfx_slab_exact_tiled_scratch_bytes": (C.c_size_t, [C.c_size_t, C.c_size_t, C.c_int, C.c_int]),
            "kfx_slab_exact_ghost": (C.c_int, [C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_size_t, C.c_float, PF, C.c_int, C.c_int]),
            "kfx_slab_set_normals_stage": (C.c_int, [C.c_int]),
            "kfx_slab_raycast_exact_tiled": (C.c_int, [PI, PI, PI, V, PV, PL, PF, PF, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, _P, V,
                                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]),
            "kfx_slab_frame_create": (C.c_int, [C.POINTER(V), C.POINTER(KfxSlabFrameConfig), _P]),
            "kfx_slab_frame_destroy": (C.c_int, [V]),
            "kfx_slab_frame_configure": (C.c_int, [V, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
            "kfx_slab_frame_reset": (C.c_int, [V, V]),
            "kfx_slab_frame_step": (C.c_int, [V, PI, PF, PF, C.c_uint, V]),
            "kfx_slab_frame_wait": (C.c_int, [V, V]),
            "kfx_slab_frame_sync": (C.c_int, [V, V]),
            "kfx_slab_frame_count": (C.c_longlong, [V]),
            "kfx_slab_frame_set_timing": (C.c_int, [V, C.c_uint]),
            "kfx_slab_frame_timings": (C.c_int, [V, C.c_longlong, C.c_int, PF]),
            "kfx_slab_frame_last_steps": (C.c_int, [V]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _bound = True
    return L


def _stream(stream=None):
    return V(stream if stream is not None else torch.cuda.current_stream().cuda_stream)


class _DevMem:
    """A device range as something torch.as_tensor understands (the callbacks of Comm.torch see raw pointers)."""

    def __init__(self, ptr, nbytes, typestr="|u1", itemsize=1):
        self.__cuda_array_interface__ = {"shape": (nbytes // itemsize,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _dev(ptr, nbytes, dtype=torch.uint8):
    ts, sz = {torch.uint8: ("|u1", 1), torch.int32: ("<i4", 4), torch.float32: ("<f4", 4), torch.int64: ("<i8", 8)}[dtype]
    return torch.as_tensor(_DevMem(ptr, nbytes, ts, sz), device="cuda")


class Comm:
    """One rank's kfx_comm."""

    def __init__(self, struct, keep=None, owner=True):
        self.c = struct
        self._keep = keep       # whatever the table points into (callbacks, the thread group's array)
        self._owner = owner

    @property
    def rank(self):
        return self.c.rank

    @property
    def world(self):
        return self.c.world

    def ref(self):
        return C.byref(self.c)

    def barrier(self):
        _lib.check(self.c.barrier(C.byref(self.c)))

    def destroy(self):
        if self._owner and self.c.destroy:
            self.c.destroy(C.byref(self.c))
            self._owner = False

    @staticmethod
    def rccl(rank, world, rendezvous_file, timeout_s=120):
        """libkfx_rccl.so: one process per GPU, RCCL over xGMI.  The caller has selected its device."""
        path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libkfx_rccl.so")
        R = C.CDLL(path)
        R.kfx_comm_create_rccl.argtypes = [_P, C.c_int, C.c_int, C.c_char_p, C.c_int]
        R.kfx_comm_create_rccl.restype = C.c_int
        c = KfxComm()
        st = R.kfx_comm_create_rccl(C.byref(c), rank, world, str(rendezvous_file).encode(), int(timeout_s))
        if st != 0:
            raise RuntimeError("kfx_comm_create_rccl failed: %d (1000 + ncclResult_t, or KFX_E_*)" % st)
        return Comm(c, keep=R)

    @staticmethod
    def threads(world, p2p=False, timeout_ms=10000):
        """world comms for host threads of this process that share the current device (every collective must be called by all).
        p2p: the neighbour exchanges are matched pairwise and in order per directed link, the way RCCL matches send / recv, instead
        of being barriers of all ranks; a leg without a matching partner blocks until the timeout, then fails (KFX_E_TIMEOUT)."""
        arr = (KfxComm * world)()
        if p2p:
            _lib.check(_L().kfx_comm_create_threads_p2p(arr, world, int(timeout_ms)))
        else:
            _lib.check(_L().kfx_comm_create_threads(arr, world))
        return [Comm(arr[r], keep=arr, owner=(r == 0)) for r in range(world)]

    @staticmethod
    def loopback(rank, world):
        c = KfxComm()
        _lib.check(_L().kfx_comm_create_loopback(C.byref(c), rank, world))
        return Comm(c)

    @staticmethod
    def torch(dist, group=None, _into=None, _keep=None):
        """Collectives through torch.distributed on tensors that alias the C side's device buffers: the transport of the tests'
        gloo ranks sharing one GPU (gloo moves device tensors through host copies, so every call synchronises) -- and a way to run
        kfx_slab_frame over whatever backend a launcher has already set up.  group: the process group (default: the world);
        kfx_comm::dup makes a new group over the same ranks (dist.new_group: every rank calls)."""
        rank, world = dist.get_rank(), dist.get_world_size()
        P2P = lambda op, t, peer: dist.P2POp(op, t, peer, group=group)   # noqa: E731
        nccl = dist.get_backend() == "nccl"

        def guard(fn):
            def call(*a):
                try:
                    # everything torch issues here -- copies, RCCL collectives -- goes to the stream the library is enqueueing the
                    # frame on (the last argument of every collective: the caller's stream, or the frame's side stream for an
                    # overlapped merge), not to whatever torch's current stream happens to be
                    st = a[-1] if len(a) > 1 else None
                    if st:
                        with torch.cuda.stream(torch.cuda.ExternalStream(int(st))):
                            fn(*a)
                    else:
                        fn(*a)
                    return 0
                except Exception as e:   # noqa: BLE001  (a Python exception must not unwind through C)
                    import sys
                    print("kangaroo_amd.slab.Comm.torch: %r" % (e,), file=sys.stderr)
                    return -4
            return call

        def sync():
            if not nccl:
                torch.cuda.synchronize()

        def p2p(ops):
            if not ops:
                return
            if nccl:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
                return
            torch.cuda.synchronize()   # gloo: stage device tensors through host copies (its own device path takes ~0.1 s per message)
            host = [(op, op.tensor.cpu() if op.op is dist.isend else torch.empty(op.tensor.shape, dtype=op.tensor.dtype)) for op in ops]
            for req in dist.batch_isend_irecv([P2P(op.op, t, op.peer) for op, t in host]):
                req.wait()
            for op, t in host:
                if op.op is not dist.isend:
                    op.tensor.copy_(t)
            torch.cuda.synchronize()

        def all_reduce(c, buf, count, op, stream):
            if count == 0:
                return
            dt = {0: torch.int64, 1: torch.float32, 2: torch.int32}[op]
            t = _dev(buf, count * (8 if op == 0 else 4), dt)
            sync()
            dist.all_reduce(t, op=dist.ReduceOp.MIN if op == 0 else dist.ReduceOp.SUM, group=group)
            sync()

        def exchange_v(c, send_lo, bsl, recv_lo, brl, send_hi, bsh, recv_hi, brh, stream):
            ops = []
            if rank > 0 and bsl:
                ops.append(P2P(dist.isend, _dev(send_lo, bsl), rank - 1))
            if rank > 0 and brl:
                ops.append(P2P(dist.irecv, _dev(recv_lo, brl), rank - 1))
            if rank + 1 < world and bsh:
                ops.append(P2P(dist.isend, _dev(send_hi, bsh), rank + 1))
            if rank + 1 < world and brh:
                ops.append(P2P(dist.irecv, _dev(recv_hi, brh), rank + 1))
            p2p(ops)

        def exchange(c, send_lo, recv_lo, blo, send_hi, recv_hi, bhi, stream):
            exchange_v(c, send_lo, blo, recv_lo, blo, send_hi, bhi, recv_hi, bhi, stream)

        def barrier(c):
            torch.cuda.synchronize()
            dist.barrier(group=group)

        def broadcast(c, buf, nbytes, root, stream):
            if nbytes == 0:
                return
            sync()
            dist.broadcast(_dev(buf, nbytes), src=root, group=group)
            sync()

        def all_to_all(c, send, recv, nbytes, stream):
            if nbytes == 0:
                return
            s_, r_ = _dev(send, nbytes * world).view(world, nbytes), _dev(recv, nbytes * world).view(world, nbytes)
            if nccl:
                dist.all_to_all_single(r_, s_, group=group)
                return
            sync()   # (the producer may have run on another stream than the one this copy is issued on)
            r_[rank].copy_(s_[rank])
            ops = []
            for k in range(1, world):
                to, frm = (rank + k) % world, (rank - k) % world
                ops += [P2P(dist.isend, s_[to], to), P2P(dist.irecv, r_[frm], frm)]
            p2p(ops)

        def all_gather(c, send, recv, nbytes, stream):
            if nbytes == 0:
                return
            s_, r_ = _dev(send, nbytes), _dev(recv, nbytes * world)
            if nccl:
                dist.all_gather_into_tensor(r_, s_, group=group)
                return
            sync()
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, s_.cpu(), group=group)
            r_.copy_(torch.cat(parts))
            sync()

        c = KfxComm() if _into is None else _into
        c.rank, c.world, c.impl = rank, world, None
        fields = dict(KfxComm._fields_)
        cbs = {} if _keep is None else _keep
        tag = "" if _into is None else "dup%d." % len(cbs)
        for name, fn in (("all_reduce", all_reduce), ("exchange", exchange), ("barrier", barrier), ("broadcast", broadcast),
                         ("all_to_all", all_to_all), ("all_gather", all_gather), ("exchange_v", exchange_v)):
            cbs[tag + name] = fields[name](guard(fn))
            setattr(c, name, cbs[tag + name])
        cbs[tag + "destroy"] = fields["destroy"](lambda c_: None)
        c.destroy = cbs[tag + "destroy"]
        c.flags = 0 if nccl else HOST_BLOCKING   # gloo: every call synchronises with the peers

        def dup(c_, out):
            try:
                g2 = dist.new_group(ranks=list(range(world)), backend=dist.get_backend())   # (collective: every rank duplicates alike)
                Comm.torch(dist, group=g2, _into=out.contents, _keep=cbs)
                return 0
            except Exception as e:   # noqa: BLE001
                import sys
                print("kangaroo_amd.slab.Comm.torch dup: %r" % (e,), file=sys.stderr)
                return -4
        cbs[tag + "dup"] = fields["dup"](dup)
        c.dup = cbs[tag + "dup"]
        return Comm(c, keep=cbs, owner=False)


def exact_ghost(dims, boxmin, boxmax, trunc, K, w, h):
    """kfx_slab_exact_ghost: the ghost planes per side with which the exact hand-over needs no last stage for the normals."""
    Kc = (C.c_float * 4)(*[float(x) for x in K])
    return int(_L().kfx_slab_exact_ghost(int(dims[2]), float(boxmin[2]), float(boxmax[2]), float(boxmax[0]) - float(boxmin[0]), int(dims[0]),
                                         float(trunc), Kc, int(w), int(h)))


def set_normals_stage(keep):
    """kfx_slab_set_normals_stage: keep the hand-over's last stage whatever the ghost width (every rank alike); returns the old setting."""
    return int(_L().kfx_slab_set_normals_stage(1 if keep else 0))


def layout(full_d, full_zmin, full_zmax, rank, world, ghost=2):
    lay = KfxSlabLayout()
    _lib.check(_L().kfx_slab_layout_init(C.byref(lay), full_d, float(full_zmin), float(full_zmax), rank, world, ghost))
    return lay


class SlabFrame:
    """kfx_slab_frame: the frame of one slab rank -- preprocess -> SdfFuse on this rank's planes (+ ghost planes) -> the slab
    raycast with its collectives -- enqueued by ONE library call per frame (include/kfx_slab.h)."""

    def __init__(self, comm, vol, lay, raw, filtered, vbo, normals, ray_d, ray_n, ray_i, K, bilateral, near, far, trunc_dist, max_w, mincostheta,
                 halo="recompute", raycast="exact", merge="direct", inputs="replicate", overlap=False, tiles=0, unchecked=False, timing_slots=0,
                 pipe_images=()):
        """pipe_images: [(depth, norm, img), ...] -- the image sets 1 .. of the pipelined exact raycast (raycast="exact" with overlap: frame
        k renders into set k % (1 + len(pipe_images)), set 0 being ray_d / ray_n / ray_i); at most PIPE_MAX - 1 of them."""
        self.comm = comm
        self._keep = (vol, raw, filtered, vbo, normals, ray_d, ray_n, ray_i, tuple(pipe_images))   # the frame holds raw pointers into these
        self.image_sets = [(ray_d, ray_n, ray_i)] + [tuple(t) for t in pipe_images]
        cfg = KfxSlabFrameConfig()
        cfg.local, cfg.layout = vol.view(), lay
        cfg.raw, cfg.filtered, cfg.vbo, cfg.normals = raw.view(), filtered.view(), vbo.view(), normals.view()
        cfg.ray_depth, cfg.ray_norm, cfg.ray_img = ray_d.view(), ray_n.view(), ray_i.view()
        for i in range(4):
            cfg.K[i] = float(K[i])
        cfg.bilateral_gs, cfg.bilateral_gr = float(bilateral["gs"]), float(bilateral["gr"])
        cfg.bilateral_size, cfg.bilateral_minval = int(bilateral["size"]), float(bilateral["minval"])
        cfg.near, cfg.far, cfg.trunc_dist, cfg.max_w, cfg.mincostheta = float(near), float(far), float(trunc_dist), float(max_w), float(mincostheta)
        cfg.halo, cfg.raycast, cfg.merge, cfg.inputs = HALO[halo], RAYCAST[raycast], MERGE[merge], INPUTS[inputs]
        cfg.overlap, cfg.tiles, cfg.unchecked, cfg.timing_slots = int(bool(overlap)), int(tiles), int(bool(unchecked)), int(timing_slots)
        assert len(pipe_images) <= PIPE_MAX - 1
        cfg.pipe_depth = 1 + len(pipe_images) if pipe_images else 0
        for k, (d_, n_, i_) in enumerate(pipe_images):
            cfg.pipe_images[3 * k], cfg.pipe_images[3 * k + 1], cfg.pipe_images[3 * k + 2] = d_.view(), n_.view(), i_.view()
        self.handle = V()
        _lib.check(_L().kfx_slab_frame_create(C.byref(self.handle), C.byref(cfg), comm.ref()))
        self.timing_slots = int(timing_slots)

    def __del__(self):
        try:
            if self.handle is not None and self.handle.value:
                _L().kfx_slab_frame_destroy(self.handle)
                self.handle = None
        except Exception:   # interpreter shutdown
            pass

    def configure(self, halo=None, raycast=None, merge=None, inputs=None, overlap=None, tiles=None):
        a = [-1 if v is None else m[v] for v, m in ((halo, HALO), (raycast, RAYCAST), (merge, MERGE), (inputs, INPUTS))]
        _lib.check(_L().kfx_slab_frame_configure(self.handle, a[0], a[1], a[2], a[3], -1 if overlap is None else int(bool(overlap)),
                                                 -1 if tiles is None else int(tiles)))

    def reset(self, stream=None):
        _lib.check(_L().kfx_slab_frame_reset(self.handle, _stream(stream)))

    @property
    def count(self):
        return int(_L().kfx_slab_frame_count(self.handle))

    @property
    def last_steps(self):
        return int(_L().kfx_slab_frame_last_steps(self.handle))

    EVENTS_NONE, EVENTS_FUSE, EVENTS_ALL = 0, 6, 31

    def set_timing(self, mask):
        """Which device events the following steps record (EVENTS_*; True = all, False = none): each costs the stream ~3 us."""
        mask = self.EVENTS_ALL if mask is True else (self.EVENTS_NONE if mask is False else int(mask))
        _lib.check(_L().kfx_slab_frame_set_timing(self.handle, mask))

    def step(self, T_wc, T_cw=None, raw=None, parts=0, stream=None):
        twc = np.ascontiguousarray(np.asarray(T_wc, np.float32)[:3].reshape(-1))
        tcw = None if T_cw is None else np.ascontiguousarray(np.asarray(T_cw, np.float32)[:3].reshape(-1))
        PF = _lib.PF
        _lib.check(_L().kfx_slab_frame_step(self.handle, None if raw is None else raw.ref(), twc.ctypes.data_as(PF),
                                            None if tcw is None else tcw.ctypes.data_as(PF), int(parts), _stream(stream)))

    def wait(self, stream=None):
        _lib.check(_L().kfx_slab_frame_wait(self.handle, _stream(stream)))

    def images(self, frame=-1):
        """(depth, norm, img) holding the rendering of `frame` (default: the last one stepped): set frame % depth of the pipelined
        exact raycast, else the frame's own ray images.  Valid once wait() / sync() has passed."""
        d, n, i = KfxImage(), KfxImage(), KfxImage()
        _lib.check(_L().kfx_slab_frame_images(self.handle, int(frame), C.byref(d), C.byref(n), C.byref(i)))
        for s_ in self.image_sets:
            if s_[0].view().ptr == d.ptr:
                return s_
        raise RuntimeError("kfx_slab_frame_images returned images this object does not know")

    def sync(self, stream=None):
        _lib.check(_L().kfx_slab_frame_sync(self.handle, _stream(stream)))

    def timings(self, first, n):
        """(n, 6) float32 milliseconds: preprocess, sdf_fuse (+ ghost planes), raycast, merge, frame, period (NaN: not recorded)."""
        out = np.full((n, TIMING_FIELDS), np.nan, np.float32)
        if n > 0:
            _lib.check(_L().kfx_slab_frame_timings(self.handle, int(first), int(n), out.ctypes.data_as(_lib.PF)))
        return out
