#!/usr/bin/env python3
"""bench.py -- KinectFusion frames/s (640x480 depth -> 512^3 TSDF) on MI355X, plus the SdfFuse
HBM roofline and the CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scene full|room] [--math fast|exact] [--no-cpu-baseline]

A "step" is one frame of the headless KinectFusion loop on synthetic depth that is already
resident in HBM: BilateralFilter -> DepthToVbo -> NormalsFromVbo -> SdfFuse -> RaycastSdf
(BASELINE.json configs[1]: 512^3 TSDF SdfFuse + RaycastSdf, 640x480, 1x MI355X; the cheap
preprocess chain is included so that a step is a whole frame).  Frames follow a 30-pose orbit
with known poses (no ICP).  N > 1: the 512^3 volume is Z-slab partitioned over
the ranks (strong scaling), see kangaroo_amd/pipeline.py.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
N_ORBIT = 30
PMC_TRAFFIC_FILE = "profiles/r03_pmc_traffic.json"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--scene", default="full", choices=["room", "full"],
                    help="full (default) = S_full of SURVEY 8(d), the roofline scene: a wall behind the volume, ~98 %% of the "
                         "voxels are updated every frame (the heaviest SdfFuse traffic); room = S_room, a furnished room "
                         "(61 %% updated)")
    ap.add_argument("--math", default="fast", choices=["fast", "exact"],
                    help="numerics mode of SdfFuse: fast = rcp/rsq/FMA perf build (reference's own -use_fast_math regime, "
                         "tolerance-tested), exact = IEEE, bit-identical to the oracle")
    ap.add_argument("--res", type=int, default=512, help="volume resolution N (N^3 voxels)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--raycast", default="composite", choices=["composite", "exact", "exact_allreduce"],
                    help="multi-GPU raycast: per-slab march + nearest-hit composite; the bit-exact march-state hand-over between "
                         "neighbour ranks (world + 1 stages, no host check in between); or its cross-check with an all-reduce per round")
    ap.add_argument("--halo", default="recompute", choices=["recompute", "exchange"],
                    help="N > 1: how ghost planes are kept current -- recomputed by each rank (no traffic) or "
                         "exchanged with the neighbours over RCCL point-to-point after each SdfFuse")
    ap.add_argument("--inputs", default="replicate", choices=["replicate", "broadcast"],
                    help="N > 1: every rank preprocesses the frame itself, or rank 0 does and broadcasts the filtered depth + normal map")
    ap.add_argument("--images", default="all", choices=["all", "root"],
                    help="N > 1, composite raycast: every rank ends up with the merged images (all-reduce of the winners' payload), or "
                         "only rank 0 does (reduce: half the traffic)")
    ap.add_argument("--overlap", action="store_true",
                    help="N > 1, composite raycast: merge frame k's images on a second stream under frame k+1's preprocess + SdfFuse")
    ap.add_argument("--summary", nargs="?", const="on", default="auto", choices=["auto", "on", "off"],
                    help="fast math, 1 GPU: the brick summary (kfx_sdf_summary: SdfFuse keeps value ranges per 8^3 cells, RaycastSdf "
                         "marches through class tables built from them and crosses free / never-observed space without reading the "
                         "volume).  auto (default): FramePipeline(track='auto') times both marches on frames 8-19 of the stream and "
                         "keeps the faster (the table march in S_full: +19 %% frames/s, the plain one in S_room); on: always; off: "
                         "the plain kernels.  The line reports the other variant beside the headline")
    ap.add_argument("--prime", type=int, default=450,
                    help="untimed frames of the same stream run before the W warm-up steps, so that the timed steps see a volume in "
                         "steady state and settled clocks whatever W is (0 = start from the freshly reset volume).  450 = fifteen "
                         "orbits, 0.2 s: --summary auto decides on frames 8-19, and the clocks take ~250 frames to settle after "
                         "those twelve frames with two marches each (SdfFuse 0.30 -> 0.277 ms in S_room, measured with "
                         "KFX_BENCH_DUMP=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=30, help="upper bound on the timed CPU-baseline frames (the sample also stops after ~12 s)")
    return ap.parse_args()


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, scene, n_frames):
    """The oracle (kind "port") timed on this host's cores: full frames of the same workload
    (same volume size, image size, scene, poses).  Two figures: all cores (OpenMP
    over z-slices / rows) = `value`, and one full frame on a single thread.  The library is
    rebuilt -O3 -march=native for this host (oracle/_native, same -ffp-contract=off arithmetic);
    if that build fails the portable -O2 build of the parity tests is timed instead."""
    import oracle
    from kangaroo_amd import scenes
    build = oracle.use_native_build()
    N, w, h = args.res, args.width, args.height
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h)
    tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
    threads = oracle.max_threads()
    vol = oracle.Volume(N, N, N, bmin, bmax)
    oracle.sdf_reset(vol, float("nan"))
    f, vbo, nrm = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h, channels=4)
    rd, rn, ri = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)

    def frame(i, nt):
        T_wc = scenes.orbit_pose(i, N_ORBIT)
        raw = oracle.Image.from_numpy(scenes.render_depth(scene, w, h, T_wc, K))
        t0 = time.perf_counter()
        oracle.bilateral(f, raw, nthreads=nt, **scenes.BILATERAL)
        oracle.depth_to_vbo(vbo, f, K)
        oracle.normals_from_vbo(nrm, vbo)
        oracle.sdf_fuse(vol, f, nrm, scenes.se3_inverse(T_wc), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, nthreads=nt)
        oracle.raycast_sdf(rd, rn, ri, vol, T_wc, K, near, far, tr, True, nthreads=nt)
        return time.perf_counter() - t0

    times = []
    budget_s = 8.0   # bounded sample: at least 2 timed frames, then as many as fit ~8 s of CPU work (at most n_frames)
    for i in range(n_frames + 1):  # first frame untimed (page faults of the volume)
        if len(times) >= 2 and sum(times) + times[-1] > budget_s:
            break
        dt = frame(i, threads)
        if i > 0:
            times.append(dt)
    fps = len(times) / sum(times)
    single_s = frame(len(times) + 1, 1) if threads > 1 else times[-1]   # one full frame on one thread
    return {"value": round(fps, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "single_thread": {"value": round(1.0 / single_s, 4), "unit": "frames/s", "cores": 1, "sample": "1 full frame = %.1f s" % single_s},
            "cpu_model": cpu_model(), "build": build,
            "sample": "%d full frames = %.1f s (%d^3 volume, %dx%d, scene %s, orbit poses) of the C restatement oracle/kfx_oracle.c, "
                      "OpenMP over z-slices/rows on %d threads; 1 untimed warm frame; then 1 full frame on 1 thread"
                      % (len(times), sum(times), N, w, h, scene, threads)}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU, RCCL) through
    torch.distributed.run as a CHILD of this process -- which has not imported torch or touched a GPU --, relay the
    ranks' output and exit with the child's code.  Never exec: a process that has initialised the GPU must not be replaced."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    sys.exit(proc.returncode)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)   # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL needs one GPU per rank; KFX_BENCH_BACKEND=gloo lets several ranks share a GPU to smoke-test the
        # distributed code path on a 1-GPU box (never used for reported numbers)
        backend = os.environ.get("KFX_BENCH_BACKEND", "nccl")
        if backend == "nccl" and ndev < world:
            sys.exit("bench.py: --gpus %d needs %d GPUs, this node shows %d (KFX_BENCH_BACKEND=gloo lets ranks share a GPU "
                     "for smoke tests only)" % (world, world, ndev))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
    n_gpus = world if distributed else 1

    from kangaroo_amd import roo, scenes
    from kangaroo_amd.pipeline import FramePipeline, SlabPipeline

    roo.set_math_mode(args.math)
    N, w, h = args.res, args.width, args.height
    scene = args.scene
    bmin, bmax, near, far = scenes.SCENES[scene]
    K = scenes.intrinsics(w, h)
    if distributed:
        if args.overlap and args.halo == "exchange":
            sys.exit("bench.py: --overlap needs --halo recompute (collective ordering, kangaroo_amd/pipeline.py)")
        pipe = SlabPipeline(roo, dist, (N, N, N), bmin, bmax, w, h, halo=args.halo, raycast=args.raycast, K=K, near=near, far=far,
                            overlap=args.overlap, inputs=args.inputs, images=args.images)
    else:
        # fast numerics: SdfFuse keeps a brick summary of the volume as a by-product and RaycastSdf takes its steps through
        # uniformly free / never-observed regions from it (same volume bits; depth within the fast-mode tolerance of the plain
        # march, tests/test_gpu_summary.py).  Exact numerics gain nothing from it (averaged +trunc values are not bit-uniform).
        policy = args.summary if args.math == "fast" else "off"
        pipe = FramePipeline(roo, (N, N, N), bmin, bmax, w, h, K=K, near=near, far=far, track={"auto": "auto", "on": True, "off": False}[policy])

    # synthetic depth stream, uploaded once: the timed region starts with inputs resident in HBM
    poses = [scenes.orbit_pose(i, N_ORBIT) for i in range(N_ORBIT)]
    frames = []
    for T_wc in poses:
        im = roo.Image(w, h, "f32", pitch=pipe.raw.pitch)
        im.MemcpyFromHost(scenes.render_depth(scene, w, h, T_wc, K))
        frames.append(im)

    def sync_all():
        if distributed:
            pipe.wait_composite()
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    # algorithmic bytes: 16 B x N_updated + 20 B x w*h per SdfFuse launch; N_updated
    # counted per pose by the diagnostics kernel (same predicate, no volume traffic), outside timing.
    n_updated = []
    for i in range(N_ORBIT):
        pipe.preprocess(frames[i])
        n_updated.append(roo.SdfFuseCount(pipe.vol, pipe.filtered, pipe.normals, scenes.se3_inverse(poses[i]), K,
                                          pipe.trunc, pipe.mincostheta, full_extent=distributed))
    # Everything the host has to prepare comes BEFORE the priming frames, so that the GPU runs the stream -- priming, warm-up,
    # timed steps -- with nothing but the contract's barrier + synchronize in between: an idle gap of tens of milliseconds
    # (event creation, garbage collection) lets the clocks drop, and in the frame loops that run against the power limit the
    # controller's overshoot afterwards stretches the next ~30 frames by up to 25 % (measured with KFX_BENCH_DUMP=1: 0.36 ->
    # 0.46 -> 0.37 ms per tracked SdfFuse) -- the whole of a 20-step run whose W = 5 warm-up steps cannot absorb it.
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    import gc
    gc.collect()
    gc.disable()   # a generation-2 collection inside the timed region stalls the launching thread for tens of ms (seen at --steps 200)
    for i in range(max(args.prime, 0)):   # the stream so far: whole orbits before the W warm-up steps
        pipe.step(poses[i % N_ORBIT], frames[i % N_ORBIT])
    if not distributed:   # track="auto" decides on frames 8-19 of the stream: before the timed region, whatever --prime / --warmup are
        extra = 0
        while pipe.track_policy == "auto" and pipe.track_decision is None and extra < 256:
            pipe.step(poses[extra % N_ORBIT], frames[extra % N_ORBIT])
            torch.cuda.synchronize()
            extra += 1
    use_summary = bool(getattr(pipe, "track", False))   # what the timed frames run with

    for i in range(args.warmup):
        pipe.step(poses[i % N_ORBIT], frames[i % N_ORBIT])
    sync_all()
    t0 = time.perf_counter()
    for s in range(args.steps):
        i = (args.warmup + s) % N_ORBIT
        T_wc = poses[i]
        pipe.preprocess(frames[i])
        ev[s][0].record()            # events on the stream the kernels are launched on (torch current stream)
        pipe.fuse(T_wc)
        ev[s][1].record()
        ev[s][2].record()
        pipe.raycast(T_wc)
        ev[s][3].record()
    sync_all()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    fuse_ms = [ev[s][0].elapsed_time(ev[s][1]) for s in range(args.steps)]
    ray_ms = [ev[s][2].elapsed_time(ev[s][3]) for s in range(args.steps)]
    idx = [(args.warmup + s) % N_ORBIT for s in range(args.steps)]
    if os.environ.get("KFX_BENCH_DUMP") and rank == 0:   # per-step kernel windows of the timed region (transients)
        print("fuse_ms " + " ".join("%.3f" % v for v in fuse_ms[:64]), file=sys.stderr)
        print("ray_ms " + " ".join("%.3f" % v for v in ray_ms[:64]), file=sys.stderr)
    alg_bytes = [16.0 * n_updated[i] + 20.0 * w * h for i in idx]
    fuse_avg_ms = float(np.mean(fuse_ms))
    bytes_avg = float(np.mean(alg_bytes))
    achieved = bytes_avg / (fuse_avg_ms * 1e-3) / 1e9
    local_voxels = pipe.vol.w * pipe.vol.h * pipe.vol.d

    # sanity: the run produced a model and an image
    hits = int(torch.isfinite(pipe.ray_d.tensor()).sum())
    assert hits > 0, "raycast produced no hits"
    ranks_agree = None
    if distributed and args.images == "root":
        ranks_agree = None   # only rank 0 holds the merged images
    elif distributed:   # after the composite every rank must hold the same images: compare a checksum of the depth bits
        bits = torch.nan_to_num(pipe.ray_d.tensor(), nan=-1.0).contiguous().view(torch.int32).to(torch.int64)
        chk = torch.stack([bits.sum(), -bits.sum()])
        dist.all_reduce(chk, op=dist.ReduceOp.MAX)
        ranks_agree = bool(int(chk[0].item()) == -int(chk[1].item()))
        assert ranks_agree, "ranks hold different composite images"

    # The same SdfFuse launched back to back (no RaycastSdf in between), 24 launches after 6 untimed: in the frame loop the
    # plain march leaves ~250 MB of the volume in the 256 MiB memory-side cache and the SdfFuse that follows reads them from
    # there; back to back nothing precedes a launch but the previous sweep (EXPERIMENTS.md 5.4).
    back_to_back = None
    if not distributed:
        evb = []
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(90)]
        for k in range(90):   # 66 untimed launches first: the host has just read results back, and the clocks take tens of frames to settle after an idle gap
            i = (args.warmup + args.steps + k) % N_ORBIT
            pipe.preprocess(frames[i])
            b0, b1 = evs[k]
            b0.record()
            pipe.fuse(poses[i])
            b1.record()
            evb.append((i, b0, b1))
        torch.cuda.synchronize()
        bb_ms = float(np.mean([a.elapsed_time(b) for _, a, b in evb[66:]]))
        bb_bytes = float(np.mean([16.0 * n_updated[i] + 20.0 * w * h for i, _, _ in evb[66:]]))
        back_to_back = {"avg_launch_ms": round(bb_ms, 5), "achieved": round(bb_bytes / (bb_ms * 1e-3) / 1e9, 1),
                        "frac": round(bb_bytes / (bb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "note": "the same frames with no RaycastSdf between the SdfFuse launches (24 timed after 66): in the frame loop the plain march leaves part of the volume in the 256 MiB memory-side cache for the next SdfFuse"}

    # RaycastSdf and BilateralFilter by SURVEY 8(d)'s figures (1 GPU; the volume is in the timed loop's steady state).
    # RaycastSdf: algorithmic bytes 8 B x U + 24 B x w h, U = distinct voxels the reference march reads for the pose
    # (kfx_raycast_sdf_count: the same march with a bitmap, untimed), against the kernel times of the timed loop; the march
    # is bound by its chain of dependent misses, so the sample rate and the 64-byte gather rate are given beside it.
    roofline_raycast, bilateral_line, transfer_line = None, None, None
    if not distributed:
        try:
            cnt = [roo.RaycastSdfCount(pipe.vol, w, h, poses[i], K, near, far, pipe.trunc, True) for i in range(N_ORBIT)]
            ray_avg_ms = float(np.mean(ray_ms))
            U = float(np.mean([cnt[i]["U"] for i in idx]))
            smp = float(np.mean([cnt[i]["samples"] for i in idx]))
            ray_bytes = 8.0 * U + 24.0 * w * h
            roofline_raycast = {
                "kernel": "k_raycast_sdf_classes (march through the class tables)" if getattr(pipe, "track", False) else "k_raycast_sdf (plain march)",
                "bound": "hbm (by unique bytes; the march itself is latency-bound)", "achieved": round(ray_bytes / (ray_avg_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(ray_bytes / (ray_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                "algorithmic_bytes_per_launch": round(ray_bytes), "distinct_voxels": round(U), "avg_launch_ms": round(ray_avg_ms, 5),
                "samples_per_launch": round(smp), "Gsamples_per_s": round(smp / (ray_avg_ms * 1e-3) / 1e9, 3),
                "gather_64B_GBps": round(64.0 * 4 * smp / (ray_avg_ms * 1e-3) / 1e9, 1),
                "rays_in_box": round(float(np.mean([cnt[i]["rays"] for i in idx]))), "hits": round(float(np.mean([cnt[i]["hits"] for i in idx]))),
                "note": "U and samples are those of the reference march for the pose (what the images depend on); the table march takes the same steps and reads fewer cells"}
            # BilateralFilter: 8 B x w h of traffic, 2 x (2r+1)^2 = 98 exponentials per pixel in the reference's loop
            # (cu_bilateral.cu:72-88); the kernel here evaluates the 49 range weights per pixel, the spatial ones once per workgroup
            nb = 200
            for _ in range(10):
                roo.BilateralFilter(pipe.filtered, frames[0], **scenes.BILATERAL)
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b0.record()
            for k in range(nb):
                roo.BilateralFilter(pipe.filtered, frames[k % N_ORBIT], **scenes.BILATERAL)
            b1.record()
            torch.cuda.synchronize()
            bil_ms = b0.elapsed_time(b1) / nb
            bilateral_line = {"kernel": "k_bilateral%s<float, 3>" % ("_fast" if args.math == "fast" else ""), "avg_launch_ms": round(bil_ms, 5),
                              "GBps": round(8.0 * w * h / (bil_ms * 1e-3) / 1e9, 1), "Gexp_per_s": round(98.0 * w * h / (bil_ms * 1e-3) / 1e9, 1),
                              "Gexp_per_s_evaluated": round(49.0 * w * h / (bil_ms * 1e-3) / 1e9, 1),
                              "note": "%d launches back to back; 98 exp per pixel is the reference loop's count (radius 3), 49 of them are evaluated per pixel here" % nb}
            # the same frames with the depth image uploaded every frame, as the application does (main.cpp:203,
            # dKinectMeters.CopyFrom): 4 B x w h from page-locked host memory, asynchronous on the launch stream, inside the
            # timed loop -- the PCIe-inclusive rate (never `value`)
            pinned = [pipe.raw.pinned_like(scenes.render_depth(scene, w, h, poses[i], K)) for i in range(N_ORBIT)]
            n_tr = min(args.steps, 4 * N_ORBIT)
            for s_ in range(N_ORBIT):
                pipe.raw.MemcpyFromPinned(pinned[s_ % N_ORBIT])
                pipe.step(poses[s_ % N_ORBIT])
            sync_all()
            t_tr = time.perf_counter()
            for s_ in range(n_tr):
                i = (args.warmup + s_) % N_ORBIT
                pipe.raw.MemcpyFromPinned(pinned[i])
                pipe.step(poses[i])
            sync_all()
            dt_tr = time.perf_counter() - t_tr
            transfer_line = {"frames_per_sec": round(n_tr / dt_tr, 1), "steps": n_tr, "bytes_per_frame": 4 * w * h,
                             "note": "per frame: hipMemcpyAsync of the raw depth image from pinned host memory on the launch stream, then the same step"}
            del pinned
        except Exception as e:   # noqa: BLE001
            roofline_raycast = roofline_raycast or {"error": repr(e)[:300]}

    # the other numerics mode, same frames, SdfFuse only (reported beside the headline; not part of `value`)
    other = "exact" if args.math == "fast" else "fast"
    roo.set_math_mode(other)
    if getattr(pipe, "track", False):   # the other mode is timed on the plain kernels; the summary no longer describes the volume
        pipe.track = False
        pipe.summary.invalidate()
    pipe._cal = None
    n_other = min(args.steps, 2 * N_ORBIT)   # whole orbits: launch times depend on the pose
    ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_other)]
    for s in range(2 * N_ORBIT):   # untimed: the first launches after the mode switch run on cold instruction caches / a settling clock
        i = (args.warmup + s) % N_ORBIT
        pipe.preprocess(frames[i])
        pipe.fuse(poses[i])
        pipe.raycast(poses[i])
    sync_all()
    t_other = time.perf_counter()
    for s in range(n_other):
        i = (args.warmup + s) % N_ORBIT
        pipe.preprocess(frames[i])
        ev2[s][0].record()
        pipe.fuse(poses[i])
        ev2[s][1].record()
        pipe.raycast(poses[i])
    sync_all()
    other_fps = n_other / (time.perf_counter() - t_other)
    other_ms = float(np.mean([a.elapsed_time(b) for a, b in ev2]))
    other_bytes = float(np.mean([16.0 * n_updated[(args.warmup + s) % N_ORBIT] + 20.0 * w * h for s in range(n_other)]))
    roo.set_math_mode(args.math)

    # N > 1: the same frames with the other ghost-plane policy (RCCL neighbour exchange vs redundant integration: same bits)
    # and with the composite merge overlapped / not overlapped with the next frame -- reported beside the headline so that
    # one multi-GPU run of the default command measures all of them (not part of `value`)
    variants = None
    if distributed:
        def timed_fps(n):
            for s in range(3):
                i = (args.warmup + s) % N_ORBIT
                pipe.step(poses[i], frames[i])
            sync_all()
            t_v = time.perf_counter()
            for s in range(n):
                i = (args.warmup + s) % N_ORBIT
                pipe.step(poses[i], frames[i])
            sync_all()
            tt = torch.tensor([time.perf_counter() - t_v], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return round(n / float(tt.item()), 1)
        n_var = min(args.steps, 2 * N_ORBIT)
        variants = {"steps": n_var}
        base_halo, base_overlap, base_inputs, base_images = pipe.halo, pipe.overlap, pipe.inputs, pipe.images
        try:   # reported extras must never cost the headline line (errors in collectives are symmetric across ranks)
            variants["as_configured_fps"] = timed_fps(n_var)
            pipe.wait_composite()
            pipe.overlap = False   # the ghost-plane exchange never runs beside an overlapped merge (SlabPipeline.__init__)
            pipe.halo = "exchange" if base_halo == "recompute" else "recompute"
            variants["halo_%s_fps" % pipe.halo] = timed_fps(n_var)
            pipe.halo = base_halo
            pipe.overlap = base_overlap
            if args.raycast == "composite":
                pipe.wait_composite()
                pipe.overlap = not base_overlap
                variants["overlap_%s_fps" % ("on" if pipe.overlap else "off")] = timed_fps(n_var)
                pipe.wait_composite()
            pipe.overlap = base_overlap
            pipe.inputs = "broadcast" if base_inputs == "replicate" else "replicate"
            variants["inputs_%s_fps" % pipe.inputs] = timed_fps(n_var)
            pipe.inputs = base_inputs
            if args.raycast == "composite":
                pipe.wait_composite()
                pipe.images = "root" if base_images == "all" else "all"
                variants["images_%s_fps" % pipe.images] = timed_fps(n_var)
                pipe.wait_composite()
        except Exception as e:   # noqa: BLE001
            variants["error"] = repr(e)[:300]
        pipe.halo, pipe.overlap, pipe.inputs, pipe.images = base_halo, base_overlap, base_inputs, base_images

    # 1 GPU, fast numerics: the same frames with the brick summary switched on (tracked SdfFuse + RaycastSdf that steps through
    # uniform regions without reading the volume), reported beside the headline (not part of `value`)
    summary_variant, plain_variant = None, None
    if not distributed and args.math == "fast" and use_summary:
        try:   # the headline ran through the tables: the same frames with the plain kernels beside it
            pipe.track = False
            n_pv = min(args.steps, 2 * N_ORBIT)
            for s in range(N_ORBIT):
                pipe.step(poses[s % N_ORBIT], frames[s % N_ORBIT])
            evp = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n_pv)]
            sync_all()
            t_pv = time.perf_counter()
            for s in range(n_pv):
                i = (args.warmup + s) % N_ORBIT
                pipe.preprocess(frames[i])
                evp[s][0].record()
                pipe.fuse(poses[i])
                evp[s][1].record()
                evp[s][2].record()
                pipe.raycast(poses[i])
                evp[s][3].record()
            sync_all()
            dt_pv = time.perf_counter() - t_pv
            pv_fuse = float(np.mean([e[0].elapsed_time(e[1]) for e in evp]))
            pv_bytes = float(np.mean([16.0 * n_updated[(args.warmup + s) % N_ORBIT] + 20.0 * w * h for s in range(n_pv)]))
            plain_variant = {"frames_per_sec": round(n_pv / dt_pv, 1), "steps": n_pv, "sdf_fuse_ms": round(pv_fuse, 5),
                             "sdf_fuse_frac_of_peak": round(pv_bytes / (pv_fuse * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "raycast_sdf_ms": round(float(np.mean([e[2].elapsed_time(e[3]) for e in evp])), 5),
                             "note": "kfx_sdf_fuse + kfx_raycast_sdf (no summary), same frames (bench.py --summary off makes it the headline)"}
        except Exception as e:   # noqa: BLE001
            plain_variant = {"error": repr(e)[:300]}
    if not distributed and args.math == "fast" and not use_summary and hasattr(roo, "SdfSummary"):
        try:   # a reported extra must never cost the headline line
            pipe.track = True
            if pipe.summary is None:
                pipe.summary = roo.SdfSummary(pipe.vol)
            pipe.reset()                      # SdfReset of volume and summary together
            n_sv = min(args.steps, 2 * N_ORBIT)
            for s in range(2 * N_ORBIT):   # untimed: the summary of a freshly reset volume settles within the first orbit
                i = s % N_ORBIT
                pipe.step(poses[i], frames[i])
            ev3 = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n_sv)]
            sync_all()
            t_sv = time.perf_counter()
            for s in range(n_sv):
                i = (args.warmup + s) % N_ORBIT
                pipe.preprocess(frames[i])
                ev3[s][0].record()
                pipe.fuse(poses[i])
                ev3[s][1].record()
                ev3[s][2].record()
                pipe.raycast(poses[i])
                ev3[s][3].record()
            sync_all()
            dt_sv = time.perf_counter() - t_sv
            summary_variant = {"frames_per_sec": round(n_sv / dt_sv, 1), "steps": n_sv,
                               "sdf_fuse_tracked_ms": round(float(np.mean([e[0].elapsed_time(e[1]) for e in ev3])), 5),
                               "raycast_sdf_tracked_ms": round(float(np.mean([e[2].elapsed_time(e[3]) for e in ev3])), 5),
                               "note": "kfx_sdf_fuse_tracked + kfx_raycast_sdf_tracked on a freshly reset volume, same frames (bench.py --summary on makes it the headline)"}
            pipe.track = False
            pipe.summary.invalidate()
        except Exception as e:   # noqa: BLE001
            summary_variant = {"error": repr(e)[:300]}
            pipe.track = False

    # Measured ceilings of this GPU in the same run: in-place 16-byte read-modify-write sweeps of a volume of the
    # same size with no arithmetic (libkfx_debug.so, kfx_debug_rmw: the fuse kernel's own brick mapping with and without
    # nontemporal accesses, other brick shapes, a linear sweep) -- the access pattern SdfFuse has to live with -- and a plain
    # device-to-device copy.  Each probe: 2 untimed + 5 timed launches; bytes = 16 B x cells (8 B read + 8 B written).
    rmw_probe, copy_GBps = None, None
    if not distributed:
        try:
            import ctypes as C
            from kangaroo_amd import _lib
            D = _lib.load_debug()
            D.kfx_debug_rmw.restype = C.c_int
            D.kfx_debug_rmw.argtypes = [_lib.PV, C.c_int, C.c_void_p]
            scratch = roo.BoundedVolume(N, N, N, bmin, bmax)
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            best, per = 0.0, {}
            for variant in (0, 1, 10, 11, 12, 13, 14, 15, 16, 17):
                if D.kfx_debug_rmw(scratch.ref(), variant, st) != 0:
                    continue
                D.kfx_debug_rmw(scratch.ref(), variant, st)
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record()
                for _ in range(5):
                    D.kfx_debug_rmw(scratch.ref(), variant, st)
                c1.record()
                torch.cuda.synchronize()
                gbps = 5 * 16.0 * N ** 3 / (c0.elapsed_time(c1) * 1e-3) / 1e9
                per[str(variant)] = round(gbps, 1)
                best = max(best, gbps)
            rmw_probe = {"best_GBps": round(best, 1), "per_variant_GBps": per,
                         "note": "kfx_debug_rmw variants (include/kfx_debug.h): 0 linear sweep, 1 the fuse kernel's 64x8x16 brick, 10-17 generated brick shapes with / without nontemporal accesses; launched back to back"}
            del scratch
            src = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
            dst = torch.empty_like(src)
            dst.copy_(src)
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(5):
                dst.copy_(src)
            c1.record()
            torch.cuda.synchronize()
            copy_GBps = 5 * 2.0 * src.numel() * 4 / (c0.elapsed_time(c1) * 1e-3) / 1e9
            del src, dst
        except Exception as e:   # noqa: BLE001  (reported extras never cost the headline line)
            rmw_probe = {"error": repr(e)[:200]}

    # HBM bytes per launch from PMC passes.  PMC collection needs its own rocprofv3 runs (FETCH_SIZE and WRITE_SIZE do
    # not fit one pass and must not be combined with the timed run), so the figure comes from the committed summary of
    # those passes over this same command (scripts/gpu_profile.sh -> profiles/<tag>/summary.txt -> PMC_TRAFFIC_FILE);
    # `traffic_source` names the file and the commit it was collected at: it is NOT measured in this run.
    traffic, traffic_source = None, None
    if not distributed and (N, w, h) == (512, 640, 480):
        try:
            with open(os.path.join(ROOT, PMC_TRAFFIC_FILE)) as fh:
                tj = json.load(fh)
            traffic = tj.get("%s_%s%s" % (scene, args.math, "_tracked" if use_summary else ""), {}).get("traffic_bytes")
            if traffic is not None:
                traffic_source = "%s (separate rocprofv3 --pmc passes of this command, kernels of commit %s; not measured in this run)" % (
                    PMC_TRAFFIC_FILE, tj.get("_commit", "?"))
        except (OSError, ValueError):
            pass

    if rank == 0:
        fps = args.steps / elapsed
        out = {
            "metric": "kinectfusion_frames_per_sec_640x480_to_512cubed_tsdf",
            "value": round(fps, 3),
            "unit": "frames/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "prime": max(args.prime, 0),
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: %d^3 TSDF (SDF_t f32 val+weight, %.2f GiB), %dx%d synthetic depth "
                            "resident in HBM, scene S_%s, %d-pose orbit with known poses; per frame: BilateralFilter(7x7) "
                            "-> DepthToVbo -> NormalsFromVbo -> SdfFuse -> RaycastSdf; %d untimed frames of the stream precede the warm-up steps" % (
                                N, 8.0 * N ** 3 / 2 ** 30, w, h, scene, N_ORBIT, max(args.prime, 0)),
                "volume": [N, N, N], "image": [w, h], "scene": scene,
                "backend": os.environ.get("KFX_BENCH_BACKEND", "nccl (RCCL)") if distributed else None,
                "ranks_agree": ranks_agree,
                "raycast": ("march through the class tables of the brick summary, kept current by the tracked SdfFuse (kfx_sdf_fuse_tracked + kfx_raycast_sdf_tracked)"
                            if (not distributed and use_summary) else "plain march (kfx_raycast_sdf)"),
                "summary_policy": None if distributed else {"requested": args.summary if args.math == "fast" else "off (exact numerics)",
                                                            "decision": getattr(pipe, "track_decision", None)},
                "partition": ("z-slabs x%d, inputs %s, ghost planes %s%s, raycast %s" % (n_gpus, "preprocessed by every rank" if args.inputs == "replicate" else "preprocessed by rank 0 and broadcast", args.halo, ", merge overlapped with the next frame" if args.overlap else "", {"composite": "composite = all_reduce(MIN key) + %s(SUM payload)" % ("reduce-to-rank-0" if args.images == "root" else "all_reduce"), "exact": "exact = march state handed from slab to slab: world + 1 stages, neighbour send/recv between them, one all_reduce of the finalised pixels at the end", "exact_allreduce": "exact (cross-check) = one SUM all_reduce of the march state + a host-side termination test per round"}[args.raycast]))
                             if distributed else "single volume",
                "math": {"fast": "fast (fp32 rcp/rsq + FMA: the reference's own -use_fast_math regime, running average as old + (new - old) w / (w + old.w); whole chain against the exact oracle at this size, "
                                 "tests/test_gpu_chain.py: TSDF L-inf < 1e-4 on identically classified voxels (<= 2e-6 of them classified differently); raycast images: "
                                 "no hit / miss flip, depth < 1e-4 m on all but <= 2e-5 of the hits, normals < 2e-2 rad, shade < 1e-2)",
                         "exact": "exact (IEEE fp32, no FMA contraction, reference operation order; bit-identical to the oracle)"}[args.math],
            },
            "roofline": {
                "kernel": "k_sdf_fuse_tiled<%s%s> (SdfFuse, %s math%s)" % ("true" if args.math == "fast" else "false", ", TRACK" if use_summary else "", args.math,
                                                                            ", keeping the brick summary current" if use_summary else ""),
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": round(bytes_avg),
                "avg_launch_ms": round(fuse_avg_ms, 5),
                "updated_fraction": round(float(np.mean([n_updated[i] for i in idx])) / local_voxels, 4),
                "full_sweep_GBps": round(16.0 * local_voxels / (fuse_avg_ms * 1e-3) / 1e9, 1),
                "back_to_back": back_to_back,
                "rmw_probe": rmw_probe,
                "full_sweep_frac_of_best_rmw_probe": (round(16.0 * local_voxels / (fuse_avg_ms * 1e-3) / 1e9 / rmw_probe["best_GBps"], 4)
                                                      if rmw_probe and rmw_probe.get("best_GBps") else None),
                "torch_copy_GBps": None if copy_GBps is None else round(copy_GBps, 1),
                "note": "rank-0 slab" if distributed else "whole volume",
            },
            "kernels_ms": {"sdf_fuse": round(fuse_avg_ms, 5), "raycast_sdf%s" % ("+composite" if distributed else ""): round(float(np.mean(ray_ms)), 5),
                           "frame_total": round(1e3 * elapsed / args.steps, 5)},
        }
        out["sdf_fuse_other_mode"] = {"math": other, "avg_launch_ms": round(other_ms, 5),
                                      "achieved_GBps": round(other_bytes / (other_ms * 1e-3) / 1e9, 1),
                                      "frac": round(other_bytes / (other_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                      "frames_per_sec": round(other_fps, 1),
                                      "note": "same frames, whole step (preprocess + fuse + raycast), %d steps after %d untimed ones" % (n_other, 2 * N_ORBIT)}
        if roofline_raycast is not None:
            out["roofline_raycast"] = roofline_raycast
        if bilateral_line is not None:
            out["bilateral"] = bilateral_line
        if transfer_line is not None:
            out["transfer_inclusive"] = transfer_line
        if variants is not None:
            out["multi_gpu_variants"] = variants
        if summary_variant is not None:
            out["brick_summary_variant"] = summary_variant
        if plain_variant is not None:
            out["plain_variant"] = plain_variant
        if not args.no_cpu_baseline and n_gpus == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args, scene, args.cpu_frames)
            except Exception as e:  # the baseline is a reported extra; never lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
