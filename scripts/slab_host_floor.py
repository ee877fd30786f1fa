#!/usr/bin/env python3
"""The host-bound floor of one rank of an 8-GPU run, measured on ONE GPU (round-4 verdict, item 1b): rank 3 of 8 of the 512^3
(and 1024^3) volume -- its 64 (128) planes + ghosts, its fuse, its slab march, the merge's / the hand-over's kernels and copies --
stepped through the loop-back transport (kfx_comm_create_loopback: every collective moves the bytes a real one would deliver to
this rank, from this rank's own buffers), once as ONE kfx_slab_frame_step call per frame and once by SlabPipeline's
operator-by-operator Python loop.  The images are not a rendering (nobody marched the other slabs); times and launches are a
real rank's.  Reported per variant: the frame by the host clock, the frame between its first and last device event, their
difference (what the host adds), and the parts.

Usage: python scripts/slab_host_floor.py [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from kangaroo_amd import roo, scenes, slab  # noqa: E402
from kangaroo_amd.pipeline import SlabPipeline  # noqa: E402


class LoopbackDist:
    """torch.distributed's surface as SlabPipeline(driver="python") uses it, for ONE rank of `world`: what a collective would
    deliver comes from this rank's own tensors (kfx_comm_create_loopback's semantics)."""

    class ReduceOp:
        MIN, SUM, MAX = "min", "sum", "max"

    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    class _Done:
        def wait(self):
            return None

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.isend, self.irecv = "isend", "irecv"

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.world

    def get_backend(self):
        return "nccl"

    def all_reduce(self, t, op=None):
        return None

    def reduce(self, t, dst=0, op=None):
        return None

    def broadcast(self, t, src=0):
        return None

    def barrier(self):
        torch.cuda.synchronize()

    def all_to_all_single(self, recv, send):
        recv.copy_(send)

    def all_gather_into_tensor(self, full, part):
        full.view(self.world, -1).copy_(part.reshape(1, -1).expand(self.world, -1))

    def batch_isend_irecv(self, ops):
        sends = {op.peer: op.tensor for op in ops if op.op == "isend"}
        for op in ops:
            if op.op == "irecv":   # what the peer would send is as large as what this rank sends it
                src = sends.get(op.peer)
                if src is not None and src.numel() == op.tensor.numel():
                    op.tensor.copy_(src.reshape(op.tensor.shape))
        return [self._Done() for _ in ops]


def run(N, world, rank, raycast, driver, steps, scene="full", w=640, h=480, events=31, **kw):
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h)
    dist = LoopbackDist(rank, world)
    comm = slab.Comm.loopback(rank, world) if driver == "c" else None
    pipe = SlabPipeline(roo, dist, (N, N, N), bmin, bmax, w, h, halo="recompute", raycast=raycast, K=K, near=near, far=far, driver=driver, comm=comm,
                        timing_slots=steps + 64, unchecked=(driver == "c"), **kw)
    if pipe.sframe is not None:
        pipe.sframe.set_timing(events)
    poses = [scenes.orbit_pose(i, 30) for i in range(30)]
    frames = []
    for T in poses:
        im = roo.Image(w, h, "f32", pitch=pipe.raw.pitch)
        im.MemcpyFromHost(scenes.render_depth(scene, w, h, T, K))
        frames.append(im)
    ev = None
    if pipe.sframe is None:
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(steps)]
        for a, b in ev:
            a.record(); b.record()

    def sync():
        if pipe.sframe is not None:
            try:
                pipe.sframe.sync()
            except Exception:   # noqa: BLE001  (unchecked exact march: nobody finishes the other slabs' rays)
                pass
        pipe.wait_composite()
        torch.cuda.synchronize()
    for i in range(900):
        pipe.step(poses[i % 30], frames[i % 30])
    sync()
    first = pipe.sframe.count if pipe.sframe is not None else 0
    t0 = time.perf_counter()
    for s in range(steps):
        if ev is not None:
            ev[s][0].record()
        pipe.step(poses[s % 30], frames[s % 30])
        if ev is not None:
            ev[s][1].record()
    sync()
    total = (time.perf_counter() - t0) / steps * 1e3
    out = {"frame_host_clock_ms": round(total, 5), "frames_per_sec": round(1e3 / total, 1)}
    if pipe.sframe is not None:
        t = pipe.sframe.timings(first, steps)
        opt = lambda col: None if not np.isfinite(t[:, col]).all() else round(float(np.mean(t[:, col])), 5)   # noqa: E731
        out.update({"events_recorded_per_frame": bin(events).count("1"), "preprocess_ms": opt(0), "sdf_fuse_ms": opt(1), "raycast_ms": opt(2), "merge_ms": opt(3),
                    "frame_events_ms": opt(4), "period_events_ms": round(float(np.nanmean(t[:, 5])), 5)})
        out["host_gap_ms"] = round(total - out["period_events_ms"], 5)
        if raycast == "exact":
            out["handover_steps"] = pipe.sframe.last_steps
    else:
        fe = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        out.update({"frame_events_ms": round(fe, 5), "host_gap_ms": round(total - fe, 5)})
    del pipe, frames
    torch.cuda.empty_cache()
    return out


def main():
    roo.set_math_mode("fast")
    res = {"note": "rank 3 of 8, loop-back transport, halo recompute, 640x480, scene S_full, fast numerics; 300 timed frames after 900 untimed"}
    for N in (512, 1024):
        r = {}
        r["c_exact_tiles4"] = run(N, 8, 3, "exact", "c", 300, tiles=4)
        r["c_exact_tiles1"] = run(N, 8, 3, "exact", "c", 300, tiles=1)
        r["c_composite_direct"] = run(N, 8, 3, "composite", "c", 300, merge="direct")
        r["c_composite_direct_two_events"] = run(N, 8, 3, "composite", "c", 300, merge="direct", events=6)
        r["c_exact_tiles4_two_events"] = run(N, 8, 3, "exact", "c", 300, tiles=4, events=6)
        # round 6: frames pipelined -- the final exchange of frame k on the frame object's side stream (second communicator) under frame k + 1
        r["c_exact_tiles4_pipelined_two_events"] = run(N, 8, 3, "exact", "c", 300, tiles=4, events=6, overlap=True)
        r["c_exact_tiles1_pipelined_two_events"] = run(N, 8, 3, "exact", "c", 300, tiles=1, events=6, overlap=True)
        r["c_exact_tiles1_two_events"] = run(N, 8, 3, "exact", "c", 300, tiles=1, events=6)
        # two ghost planes per side: the hand-over with its last stage (the other exact rows: kfx_slab_exact_ghost's width, without it)
        r["c_exact_tiles4_two_events_ghost2"] = run(N, 8, 3, "exact", "c", 300, tiles=4, events=6, ghost=2)
        r["c_exact_tiles4_pipelined_two_events_ghost2"] = run(N, 8, 3, "exact", "c", 300, tiles=4, events=6, overlap=True, ghost=2)
        r["c_composite_direct_overlapped"] = run(N, 8, 3, "composite", "c", 300, merge="direct", overlap=True)
        r["python_composite_direct"] = run(N, 8, 3, "composite", "python", 300, merge="direct")
        r["python_composite_direct_overlapped"] = run(N, 8, 3, "composite", "python", 300, merge="direct", overlap=True)
        res["%d_cubed_slab_of_8" % N] = r
        for k, v in r.items():
            print(N, k, json.dumps(v), flush=True)
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
