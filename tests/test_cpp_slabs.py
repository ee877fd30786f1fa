"""The Z-slab partition driven from C++ only (apps/kinectfusion_slabs.cpp -> include/kangaroo/SlabVolume.h -> include/kfx_slab.h):
no Python in the product path.  The GPU box has one GPU, so the multi-rank runs use the in-process transport (ranks = host
threads sharing the device): same slab code, same kernels, collectives by the thread group instead of RCCL.  The RCCL
transport itself runs with a single rank (RCCL refuses two ranks on one device); with more GPUs scripts/launch_ranks.sh
starts one process per GPU."""
import os
import re
import subprocess

import pytest

import kfx_testlib as T

APP = os.path.join(T.ROOT, "apps", "kinectfusion_slabs")
pytestmark = pytest.mark.gpu


def run(*args, env=None):
    out = subprocess.run([APP] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    m = re.search(r"checksums depth=(\w+) norm=(\w+) img=(\w+) volume=(\w+) history=(\w+) hits=(\d+) ranks_agree=(\d)", out.stdout)
    assert m, out.stdout
    ms = float(re.search(r"([\d.]+) ms/frame", out.stdout).group(1))
    return dict(depth=m.group(1), norm=m.group(2), img=m.group(3), volume=m.group(4), history=m.group(5), hits=int(m.group(6)), agree=int(m.group(7)), ms=ms,
                text=out.stdout)


COMMON = ("--res", 128, "--frames", 4, "--width", 320, "--height", 240)


def test_cpp_slabs_exact_march_equals_single_volume():
    """4 and 3 slabs, ghost planes exchanged or recomputed, march state handed over: volume and all three images are
    bit-identical to the one-slab run (which is the single-volume pipeline)."""
    ref = run(*COMMON, "--ranks", 1, "--raycast", "exact")
    assert ref["hits"] > 320 * 240 // 3 and ref["agree"] == 1
    for ranks, halo, mode in ((4, "exchange", "exact"), (3, "recompute", "exact"), (8, "exchange", "exact"), (2, "exchange", "exact"),
                              (4, "exchange", "exact-allreduce")):   # the last: the all-reduce cross-check of the hand-over
        got = run(*COMMON, "--ranks", ranks, "--raycast", mode, "--halo", halo)
        assert got["agree"] == 1
        for k in ("depth", "norm", "img", "volume", "hits"):
            assert got[k] == ref[k], (ranks, halo, mode, k, got["text"], ref["text"])
        rounds = int(re.search(r"\((\d+) rounds\)", got["text"]).group(1))
        assert rounds == ranks + 1 if mode == "exact" else 1 < rounds <= ranks + 3   # hand-over: a fixed number of stages


def test_cpp_slabs_tiled_handover_equals_single_volume():
    """The hand-over pipelined over image row-tiles (kfx_slab_raycast_exact_tiled: the march state of a tile travels as a token
    up and down the rank order, one tile-sized message per link and step): volume and all three images bit-identical to the
    one-slab run for 2 / 3 / 4 / 8 rank threads and 1 / 4 / 8 tiles (240 rows: 8 tiles of 30; 7 tiles: the last one short)."""
    ref = run(*COMMON, "--ranks", 1, "--raycast", "exact")
    one = run(*COMMON, "--ranks", 1, "--raycast", "exact", "--tiles", 4)
    for k in ("depth", "norm", "img", "volume", "hits"):
        assert one[k] == ref[k], (k, one["text"], ref["text"])
    for ranks, halo in ((2, "exchange"), (3, "recompute"), (4, "exchange"), (8, "recompute")):
        for tiles in (1, 4, 8) + ((7,) if ranks == 3 else ()):
            got = run(*COMMON, "--ranks", ranks, "--raycast", "exact", "--halo", halo, "--tiles", tiles)
            assert got["agree"] == 1
            for k in ("depth", "norm", "img", "volume", "hits"):
                assert got[k] == ref[k], (ranks, halo, tiles, k, got["text"], ref["text"])
            rounds = int(re.search(r"\((\d+) rounds\)", got["text"]).group(1))
            assert rounds == ranks + tiles - 1 + 1, (ranks, tiles, rounds)   # world + tiles - 1 token steps, the normals' stage
    # the finalised results travel by direct sends (strips: all-to-all, sum at the owner, all-gather); the all-reduce they replace
    # gives the same images (3 and 5 ranks: strips that do not divide the image evenly)
    for ranks in (3, 5):
        got = run(*COMMON, "--ranks", ranks, "--raycast", "exact", "--tiles", 4, env=dict(os.environ, KFX_SLAB_FINALISE="allreduce"))
        for k in ("depth", "norm", "img", "volume", "hits"):
            assert got[k] == ref[k], (ranks, k, got["text"], ref["text"])
        other = run(*COMMON, "--ranks", ranks, "--raycast", "exact", "--tiles", 4)
        for k in ("depth", "norm", "img", "volume", "hits"):
            assert other[k] == ref[k], (ranks, k, other["text"], ref["text"])


def test_cpp_slabs_frame_driver_equals_the_operator_calls():
    """--driver frame: every rank issues its frame as ONE kfx_slab_frame_step call (preprocess, kfx_sdf_fuse_slab, ghost planes,
    the slab raycast and its collectives enqueued by the library).  Same volume and image bits as the roo:: calls of the default
    driver -- exact (default 4 tiles, and 1 / 8), composite with either merge, the merge overlapped with the next frame, inputs
    broadcast, ghost planes exchanged or recomputed."""
    for ranks in (2, 3, 4, 8):
        ref = run(*COMMON, "--ranks", 1, "--raycast", "exact")
        for extra in (("--halo", "recompute"), ("--halo", "exchange", "--tiles", 8), ("--halo", "exchange", "--inputs", "broadcast", "--tiles", 1)):
            got = run(*COMMON, "--ranks", ranks, "--raycast", "exact", "--driver", "frame", *extra)
            assert got["agree"] == 1 and "one kfx_slab_frame_step per frame" in got["text"]
            for k in ("depth", "norm", "img", "volume", "hits"):
                assert got[k] == ref[k], (ranks, extra, k, got["text"], ref["text"])
    for ranks in (2, 4, 5):
        for merge in ("direct", "allreduce"):
            ops = run(*COMMON, "--ranks", ranks, "--raycast", "composite", "--halo", "recompute", "--merge", merge)
            for extra in ((), ("--overlap",)):
                got = run(*COMMON, "--ranks", ranks, "--raycast", "composite", "--halo", "recompute", "--merge", merge, "--driver", "frame", *extra)
                assert got["agree"] == 1
                for k in ("depth", "norm", "img", "volume", "hits"):
                    assert got[k] == ops[k], (ranks, merge, extra, k, got["text"], ops["text"])
    # the overlapped merge beside the ghost-plane exchange would interleave collectives in rank-dependent order: refused
    bad = subprocess.run([APP, *[str(a) for a in COMMON], "--ranks", "2", "--raycast", "composite", "--halo", "exchange", "--driver", "frame", "--overlap"],
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "overlap needs" in bad.stdout + bad.stderr


def test_cpp_slabs_pipelined_frames_equal_single_volume():
    """Round-5 verdict, item 1: the exact hand-over with the frames PIPELINED across the ranks (kfx_slab_frame, raycast exact +
    overlap: frame k's final exchange on the side stream through a second communicator -- kfx_comm::dup -- into image set k % D while
    the caller's stream carries frame k + 1; the application steps all frames back to back, no synchronisation, no barrier, and picks
    frame k's rendering up when frame k + D is about to take its set).  EVERY frame's rendered depth (`history`), the last frame's
    three images and the volume are bit-identical to the one-slab run: 2 / 3 / 4 / 8 rank threads x 1 / 4 / 8 tiles x 2 / 3 / 4 sets
    in flight, ghost planes exchanged or recomputed, inputs broadcast -- over the barrier transport and over the point-to-point one
    (neighbour exchanges matched pairwise like RCCL's send / recv, no common barrier: the ranks drift apart inside a frame)."""
    common = ("--res", 128, "--frames", 11, "--width", 320, "--height", 240, "--raycast", "exact")
    ref = run(*common, "--ranks", 1)
    assert ref["hits"] > 320 * 240 // 3
    unpiped = run(*common, "--ranks", 4, "--driver", "frame", "--halo", "recompute")
    keys = ("depth", "norm", "img", "volume", "history", "hits")
    for k in keys:
        assert unpiped[k] == ref[k], (k, unpiped["text"], ref["text"])
    n = 0
    for ranks in (2, 3, 4, 8):
        for tiles in (1, 4, 8):
            depth = (2, 3, 4)[n % 3]
            transport = ("threads", "threads-p2p")[(n // 3 + n) % 2]
            extra = (("--halo", "recompute"), ("--halo", "exchange"), ("--halo", "exchange", "--inputs", "broadcast"))[n % 3]
            n += 1
            got = run(*common, "--ranks", ranks, "--driver", "frame", "--tiles", tiles, "--pipeline", depth, "--transport", transport, *extra)
            assert got["agree"] == 1 and "pipeline=%d" % depth in got["text"]
            for k in keys:
                assert got[k] == ref[k], (ranks, tiles, depth, transport, extra, k, got["text"], ref["text"])
    # the other transport / depth for the 8-rank, 1-tile case: the longest token chain against the shortest own part
    for transport, depth in (("threads-p2p", 4), ("threads", 3)):
        got = run(*common, "--ranks", 8, "--driver", "frame", "--tiles", 1, "--pipeline", depth, "--transport", transport, "--halo", "recompute")
        for k in keys:
            assert got[k] == ref[k], (transport, depth, k, got["text"], ref["text"])
    # ... and the point-to-point transport under the unpipelined drivers (every path that exchanges with neighbours)
    for extra in (("--driver", "frame", "--tiles", 4), ("--halo", "exchange"), ("--tiles", 7)):
        got = run(*common, "--ranks", 3, "--transport", "threads-p2p", *extra)
        for k in keys:
            assert got[k] == ref[k], (extra, k, got["text"], ref["text"])


def test_cpp_slabs_wide_ghost_planes_drop_the_last_stage():
    """Ghost planes as wide as a hit can fall back behind the sample that found it (kfx_slab_exact_ghost: one march step along the
    longest ray of the image + the gradient stencil): the rank that finds a hit always holds its stencil, finalises it itself, and
    the hand-over runs WITHOUT its last stage (no whole-image neighbour exchange, no whole-image launch): world + tiles - 1 steps
    instead of world + tiles.  Same bits as the single volume -- every frame's depth, the last frame's images, the volume --
    unpipelined and pipelined, both thread transports, ghost planes recomputed or exchanged; KFX_SLAB_NORMALS_STAGE=1 keeps the stage."""
    common = ("--res", 128, "--frames", 7, "--width", 320, "--height", 240, "--raycast", "exact")
    ref = run(*common, "--ranks", 1)
    keys = ("depth", "norm", "img", "volume", "history", "hits")
    n = 0
    for ranks in (2, 3, 4, 8):
        for tiles in (1, 4):
            transport = ("threads", "threads-p2p")[n % 2]
            extra = (("--halo", "recompute"), ("--halo", "recompute", "--pipeline", 3), ("--halo", "exchange"), ("--halo", "exchange", "--pipeline", 2))[n % 4]
            n += 1
            got = run(*common, "--ranks", ranks, "--driver", "frame", "--tiles", tiles, "--ghost", "auto", "--transport", transport, *extra)
            assert got["agree"] == 1
            for k in keys:
                assert got[k] == ref[k], (ranks, tiles, transport, extra, k, got["text"], ref["text"])
            rounds = int(re.search(r"\((\d+) rounds\)", got["text"]).group(1))
            assert rounds == ranks + tiles - 1, (rounds, ranks, tiles, got["text"])
    # the operator-by-operator driver (roo::SlabVolume) takes the same layout; and the stage stays when asked for
    got = run(*common, "--ranks", 4, "--tiles", 4, "--ghost", "auto", "--halo", "recompute")
    for k in keys:
        assert got[k] == ref[k], (k, got["text"], ref["text"])
    assert int(re.search(r"\((\d+) rounds\)", got["text"]).group(1)) == 4 + 4 - 1
    env = dict(os.environ, KFX_SLAB_NORMALS_STAGE="1")
    kept = run(*common, "--ranks", 4, "--driver", "frame", "--tiles", 4, "--ghost", "auto", "--halo", "recompute", env=env)
    for k in keys:
        assert kept[k] == ref[k], (k, kept["text"], ref["text"])
    assert int(re.search(r"\((\d+) rounds\)", kept["text"]).group(1)) == 4 + 4
    # a narrower ghost than that keeps the stage by itself
    narrow = run(*common, "--ranks", 4, "--driver", "frame", "--tiles", 4, "--ghost", 3, "--halo", "recompute")
    for k in keys:
        assert narrow[k] == ref[k], (k, narrow["text"], ref["text"])
    assert int(re.search(r"\((\d+) rounds\)", narrow["text"]).group(1)) == 4 + 4


def test_threads_transport_point_to_point_mode_blocks_on_a_mismatched_leg():
    """Round-5 verdict, What's weak 8: on RCCL a neighbour exchange whose two sides disagree -- one skips the leg, or names another
    size -- does not fail, it HANGS.  The in-process transport's point-to-point mode (kfx_comm_create_threads_p2p) reproduces that: the
    leg blocks until the timeout and then fails with KFX_E_TIMEOUT on both sides; a matching exchange afterwards goes through; the
    barrier mode reports the same disagreement at once (KFX_E_SHAPE)."""
    import ctypes as C
    import threading
    import time
    import torch
    from kangaroo_amd import slab
    torch.cuda.set_device(0)
    a = [torch.full((256,), float(r + 1), device="cuda") for r in range(2)]
    b = [torch.zeros(256, device="cuda") for r in range(2)]
    torch.cuda.synchronize()
    V = C.c_void_p

    def pair(comms, up_bytes, want_bytes):
        """rank 0 sends up_bytes upwards, rank 1 expects want_bytes from below; (status of rank 0, of rank 1, seconds)"""
        st = [None, None]

        def r0():
            c = comms[0].c
            st[0] = c.exchange_v(C.byref(c), None, 0, None, 0, V(a[0].data_ptr()), up_bytes, None, 0, None)

        def r1():
            c = comms[1].c
            st[1] = c.exchange_v(C.byref(c), None, 0, V(b[1].data_ptr()), want_bytes, None, 0, None, 0, None)
        t0 = time.time()
        th = threading.Thread(target=r1)
        th.start()
        r0()
        th.join()
        return st[0], st[1], time.time() - t0

    comms = slab.Comm.threads(2, p2p=True, timeout_ms=400)
    s0, s1, dt = pair(comms, 64, 128)                    # sizes disagree
    assert s0 == -6 and s1 == -6 and 0.35 < dt < 5.0, (s0, s1, dt)
    s0, s1, dt = pair(comms, 64, 0)                      # the receiver skips the leg: it returns at once, the sender waits in vain
    assert s0 == -6 and s1 == 0 and dt > 0.35, (s0, s1, dt)
    s0, s1, dt = pair(comms, 1024, 1024)                 # a matching pair afterwards: the link is clean again
    torch.cuda.synchronize()
    assert s0 == 0 and s1 == 0 and dt < 0.35 and bool((b[1] == 1.0).all()), (s0, s1, dt)
    comms[0].destroy()
    comms = slab.Comm.threads(2)                         # barrier mode: the same disagreement is an error at once
    s0, s1, dt = pair(comms, 64, 128)
    assert s1 == -2 and dt < 0.35, (s0, s1, dt)          # KFX_E_SHAPE
    comms[0].destroy()


def test_cpp_slabs_inputs_broadcast_from_rank_zero():
    """--inputs broadcast: only rank 0 filters the frame and derives the normal map; kfx_slab_broadcast_inputs (pitched
    images: staged densely) hands both to the other ranks.  Same bits everywhere as with every rank preprocessing."""
    ref = run(*COMMON, "--ranks", 4, "--raycast", "exact", "--halo", "exchange")
    got = run(*COMMON, "--ranks", 4, "--raycast", "exact", "--halo", "exchange", "--inputs", "broadcast")
    assert got["agree"] == 1
    for k in ("depth", "norm", "img", "volume", "hits"):
        assert got[k] == ref[k], (k, got["text"], ref["text"])


def test_cpp_slabs_composite_and_fast_mode():
    """Nearest-hit composite: the volume is still bit-identical, every rank ends with the same images and the hit count
    stays within 1 % of the single-volume march (rays restart at slab entries).  Also in fast numerics."""
    for fast in ((), ("--fast",)):
        ref = run(*COMMON, *fast, "--ranks", 1, "--raycast", "composite")
        got = run(*COMMON, *fast, "--ranks", 4, "--raycast", "composite", "--halo", "exchange")
        assert got["agree"] == 1 and got["volume"] == ref["volume"] and "direct-send merge" in got["text"]
        assert abs(got["hits"] - ref["hits"]) <= 0.01 * ref["hits"]
        # the direct-send merge (default: kfx_slab_composite_direct, all-to-all + all-gather of image strips) picks the winners of
        # the key / payload merge (--merge allreduce: kfx_slab_composite): same depth bits, same hits
        other = run(*COMMON, *fast, "--ranks", 4, "--raycast", "composite", "--halo", "exchange", "--merge", "allreduce")
        assert other["agree"] == 1 and "all-reduce merge" in other["text"]
        assert other["depth"] == got["depth"] and other["hits"] == got["hits"] and other["volume"] == got["volume"]
    for ranks in (3, 5):   # strips that do not divide the image evenly
        a = run(*COMMON, "--ranks", ranks, "--raycast", "composite")
        b = run(*COMMON, "--ranks", ranks, "--raycast", "composite", "--merge", "allreduce")
        assert a["agree"] == 1 and b["agree"] == 1 and a["depth"] == b["depth"] and a["hits"] == b["hits"]


def test_cpp_slabs_rccl_transport_single_rank(tmp_path):
    """The RCCL transport (libkfx_rccl.so: ncclCommInitRank, all-reduce, grouped send / recv) on the one GPU of the box: a
    one-rank communicator, through the same code path N ranks take; results equal the thread transport's."""
    ref = run(*COMMON, "--ranks", 1, "--raycast", "exact")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    for mode in ("exact", "composite"):
        got = run(*COMMON, "--transport", "rccl", "--raycast", mode, "--rendezvous", str(tmp_path / "id"), env=env)
        assert "RCCL" in got["text"] and got["volume"] == ref["volume"] and got["depth"] == ref["depth"] and got["norm"] == ref["norm"]


def test_cpp_slabs_refuse_more_rccl_ranks_than_gpus(tmp_path):
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    out = subprocess.run([APP, "--transport", "rccl", "--rendezvous", str(tmp_path / "id")], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 2 and "need 2 GPUs" in out.stderr


def test_cpp_slabs_c4_1024_cubed_in_eight_slabs():
    """BASELINE config C4's volume (1024^3, 8 GiB, 640x480) cut into eight slabs the way eight GPUs would hold it -- here as
    eight rank threads on the one GPU: halo exchange + exact march give the volume and images of the one-slab run."""
    common = ("--res", 1024, "--frames", 2, "--width", 640, "--height", 480, "--raycast", "exact")
    ref = run(*common, "--ranks", 1)
    assert ref["hits"] > 640 * 480 // 3
    # the stage protocol through the roo:: calls, and the default of bench.py --gpus 8: one kfx_slab_frame_step per frame and rank,
    # ghost planes exchanged with the neighbours, the march handed over in four row-tiles
    for extra in (("--halo", "exchange"), ("--halo", "exchange", "--driver", "frame", "--tiles", 4), ("--halo", "recompute", "--driver", "frame")):
        got = run(*common, "--ranks", 8, *extra)
        assert got["agree"] == 1
        for k in ("depth", "norm", "img", "volume", "hits"):
            assert got[k] == ref[k], (extra, k, got["text"], ref["text"])


def test_rccl_transport_collectives_on_a_one_rank_communicator(tmp_path):
    """Every function of the RCCL transport's table (libkfx_rccl.so: ncclAllReduce, ncclAllGather, ncclBroadcast, grouped
    ncclSend / ncclRecv -- the neighbour exchange and the direct merge's all-to-all) on the one GPU of the box: a one-rank
    communicator is the identity, through the calls N ranks make (kfx_slab_composite_direct returns early for one rank, so the
    application test above never reaches all_to_all / all_gather)."""
    import ctypes as C
    import torch
    from kangaroo_amd import _lib

    from kangaroo_amd.slab import KfxComm as Comm   # kfx_comm (include/kfx_slab.h)
    P, V = C.POINTER(Comm), C.c_void_p
    R = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libkfx_rccl.so"))
    R.kfx_comm_create_rccl.argtypes = [P, C.c_int, C.c_int, C.c_char_p, C.c_int]
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")   # the caller selects and initialises its device
    comm = Comm()
    assert R.kfx_comm_create_rccl(C.byref(comm), 0, 1, str(tmp_path / "id").encode(), 30) == 0
    try:
        assert comm.rank == 0 and comm.world == 1
        st = V(torch.cuda.current_stream().cuda_stream)
        a = torch.arange(1000, dtype=torch.float32, device="cuda") - 7.5
        b = torch.full_like(a, float("nan"))
        want = a.clone()
        assert comm.all_to_all(C.byref(comm), V(a.data_ptr()), V(b.data_ptr()), a.numel() * 4, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(b, want)
        b.fill_(float("nan"))
        assert comm.all_gather(C.byref(comm), V(a.data_ptr()), V(b.data_ptr()), a.numel() * 4, st) == 0
        assert comm.all_reduce(C.byref(comm), V(a.data_ptr()), a.numel(), 1, st) == 0            # KFX_COMM_SUM_F32
        k = torch.arange(1000, dtype=torch.int64, device="cuda") * 3 - 11
        assert comm.all_reduce(C.byref(comm), V(k.data_ptr()), k.numel(), 0, st) == 0            # KFX_COMM_MIN_I64
        assert comm.broadcast(C.byref(comm), V(a.data_ptr()), a.numel() * 4, 0, st) == 0
        assert comm.exchange(C.byref(comm), V(a.data_ptr()), V(b.data_ptr()), 64, V(a.data_ptr()), V(b.data_ptr()), 64, st) == 0   # no neighbours: nothing moves
        assert comm.exchange_v(C.byref(comm), V(a.data_ptr()), 64, V(b.data_ptr()), 128, V(a.data_ptr()), 32, V(b.data_ptr()), 16, st) == 0   # likewise
        assert comm.barrier(C.byref(comm)) == 0
        torch.cuda.synchronize()
        assert torch.equal(b, want) and torch.equal(a, want) and torch.equal(k, torch.arange(1000, dtype=torch.int64, device="cuda") * 3 - 11)
        assert comm.all_reduce(C.byref(comm), V(a.data_ptr()), a.numel(), 99, st) != 0           # unknown operation
        # kfx_comm::dup (ncclCommSplit): a second communicator over the same ranks, what the pipelined frames' side stream uses
        assert comm.flags == 0 and bool(comm.dup)
        side = Comm()
        assert comm.dup(C.byref(comm), C.byref(side)) == 0
        try:
            assert side.rank == 0 and side.world == 1 and side.impl != comm.impl
            s2 = torch.cuda.Stream()
            b.fill_(float("nan"))
            torch.cuda.synchronize()
            assert side.all_to_all(C.byref(side), V(a.data_ptr()), V(b.data_ptr()), a.numel() * 4, V(s2.cuda_stream)) == 0
            assert comm.all_gather(C.byref(comm), V(a.data_ptr()), V(a.data_ptr()), a.numel() * 4, st) == 0   # both communicators in flight
            assert side.barrier(C.byref(side)) == 0
            torch.cuda.synchronize()
            assert torch.equal(b, want)
        finally:
            side.destroy(C.byref(side))
    finally:
        comm.destroy(C.byref(comm))
