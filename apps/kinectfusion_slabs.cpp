// kinectfusion_slabs.cpp -- the headless KinectFusion frame loop (kinectfusion_headless.cpp, known poses) with the TSDF volume
// partitioned into Z-slabs over several ranks, in C++ only: roo::SlabVolume (include/kangaroo/SlabVolume.h) over the C ABI of
// include/kfx_slab.h.  Every rank preprocesses the depth frame (replicated: < 0.03 ms), integrates its own planes, refreshes
// its ghost planes (RCCL send / recv with the two neighbours, or redundant integration), and the model is rendered by all
// ranks together (nearest-hit composite, or the exact march hand-over).
//
//   --transport rccl     one PROCESS per GPU; rank / world from RANK, WORLD_SIZE, LOCAL_RANK (torchrun --no-python, or
//                        scripts/launch_ranks.sh), ncclUniqueId through --rendezvous FILE
//   --transport threads  --ranks R host threads of this process share one GPU (emulation: exercises the same slab code on
//                        a 1-GPU box; not a performance mode)
//
// Rank 0 prints the frame time and checksums of the final raycast images and of the whole volume (sum of the owned cells'
// bit patterns over all ranks): with --raycast exact they equal the checksums of a --ranks 1 run bit for bit.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include <kangaroo/kangaroo.h>
#include <kangaroo/SlabVolume.h>

using namespace roo;

struct Options {
    int volres = 256, frames = 10, w = 640, h = 480, ranks = 1;
    bool fast = false, rccl = false, broadcast_inputs = false;
    bool frame_driver = false;   // --driver frame: one kfx_slab_frame_step call per frame instead of the roo:: calls
    bool overlap = false;        // --overlap (frame driver, composite): the merge of frame k under frame k + 1
    int pipeline = 0;            // --pipeline D (frame driver, exact): frames pipelined across the ranks, D image / buffer sets in flight (2 .. 4)
    bool p2p = false;            // --transport threads-p2p: neighbour exchanges matched pairwise like RCCL's send / recv (a mismatch blocks, then times out)
    int ghost = 2;               // --ghost G | auto: ghost planes per side (auto: kfx_slab_exact_ghost, the hand-over without its last stage)
    int tiles = 0;               // --tiles T: exact hand-over pipelined over T image row-tiles (0: whole-image stages; the frame driver's default: 4)
    SlabVolume::HaloMode halo = SlabVolume::HaloExchange;
    SlabVolume::RaycastMode raycast = SlabVolume::Composite;
    SlabVolume::MergeMode merge = SlabVolume::MergeDirect;
    std::string rendezvous;   // default: /tmp/kfx_slabs.<uid>.<launch id>.id (default_rendezvous)
};

// analytic depth of the synthetic room: same scene as kinectfusion_headless.cpp / kangaroo_amd/scenes.py
static void RenderRoom(std::vector<float>& out, int w, int h, const Mat<float,3,4>& T_wc, const ImageIntrinsics& K)
{
    out.resize((size_t)w * h);
    const float3 c = SE3Translation(T_wc);
    for (int v = 0; v < h; ++v)
        for (int u = 0; u < w; ++u) {
            const float3 r = mulSO3(T_wc, K.Unproject((float)u, (float)v));
            float best = INFINITY;
            const float3 lo = make_float3(-0.9f, -0.9f, -10.f), hi = make_float3(0.9f, 0.9f, 3.8f);
            const float3 a = div_cw(sub(lo, c), r), b = div_cw(sub(hi, c), r);
            const float texit = fminf(fminf(fmaxf(a.x, b.x), fmaxf(a.y, b.y)), fmaxf(a.z, b.z));
            if (texit > 0) best = texit;
            const float3 oc = sub(make_float3(0, 0, 3.0f), c);
            const float ldotc = dot(r, oc), lsq = dot(r, r), csq = dot(oc, oc);
            const float disc = ldotc * ldotc - lsq * (csq - 0.25f);
            if (disc >= 0) {
                const float ts = (ldotc - sqrtf(disc)) / lsq;
                if (ts > 0 && ts < best) best = ts;
            }
            out[(size_t)v * w + u] = std::isfinite(best) ? best : NAN;
        }
}

static Mat<float,3,4> OrbitPose(int i, int n)
{
    const float ph = 2.0f * (float)M_PI * i / n;
    const float yaw = 5.0f * (float)M_PI / 180.0f * sinf(ph);
    const float c = cosf(yaw), s = sinf(yaw);
    Mat<float,3,4> T = SE3Identity();
    T(0,0) = c; T(0,2) = s; T(2,0) = -s; T(2,2) = c;
    T(0,3) = 0.05f * sinf(ph);
    T(1,3) = 0.025f * (1.0f - cosf(ph)) - 0.025f;
    return T;
}

static unsigned BitSum(const void* p, size_t bytes)
{
    const unsigned* u = static_cast<const unsigned*>(p);
    unsigned s = 0;
    for (size_t i = 0; i < bytes / 4; ++i) s += u[i] * 2654435761u + (unsigned)i;   // position-dependent: a permutation changes it
    return s;
}

struct Result { double ms_per_frame = 0; unsigned chk_d = 0, chk_n = 0, chk_i = 0, chk_vol = 0, chk_hist = 0; size_t hits = 0; int rounds = 0; int status = 0; };

// the frame loop of one rank
static void RunRank(const Options& o, kfx_comm* comm, const std::vector<std::vector<float> >& depth_mm, const std::vector<Mat<float,3,4> >& poses,
                    Result* res)
{
    const int w = o.w, h = o.h;
    const double depth_focal = w * 570.342 / 640.0;
    const ImageIntrinsics K(depth_focal, depth_focal, w / 2.0 - 0.5, h / 2.0 - 0.5);
    const float knear = 0.4f, kfar = 4.0f, bigs = 1.5f, bigr = 0.1f;
    const int biwin = 3;
    const float max_w = 1000.0f, mincostheta = 0.1f;
    const BoundingBox bb(make_float3(-1, -1, 2), make_float3(1, 1, 4));

    std::vector<std::unique_ptr<Image<float, TargetDevice, Manage> > > set_d, set_i;
    std::vector<std::unique_ptr<Image<float4, TargetDevice, Manage> > > set_n;
    Image<float, TargetDevice, Manage> dMeters(w, h), dFiltered(w, h), ray_d(w, h), ray_i(w, h);
    Image<float4, TargetDevice, Manage> dVbo(w, h), dNormals(w, h), ray_n(w, h);
    // the whole volume's voxel size (the local view has the same spacing in x / y; z spacing is the full volume's)
    const float3 vs = make_float3(2.0f / (o.volres - 1), 2.0f / (o.volres - 1), 2.0f / (o.volres - 1));
    const float trunc_dist = 2.0f * length(vs);
    // --ghost G: ghost planes per side; "auto" (-1): as wide as lets the exact hand-over drop its last stage (kfx_slab_exact_ghost) where
    // every rank owns that many planes, else 2
    int ghost = o.ghost;
    if (ghost < 0) {
        const float Kf[4] = {(float)K.fu, (float)K.fv, (float)K.u0, (float)K.v0};
        ghost = kfx_slab_exact_ghost((size_t)o.volres, bb.Min().z, bb.Max().z, bb.Max().x - bb.Min().x, (size_t)o.volres, trunc_dist, Kf, w, h);
        if (comm->world > 1 && o.volres / comm->world < ghost) ghost = 2;
    }
    SlabVolume slab(o.volres, o.volres, o.volres, bb, comm, o.halo, o.raycast, ghost);
    slab.merge = o.merge;
    slab.tiles = o.tiles;
    SdfReset(slab.local, std::numeric_limits<float>::quiet_NaN());

    // --driver frame: the same frame as ONE library call per rank (kfx_slab_frame, include/kfx_slab.h)
    kfx_slab_frame* kframe = nullptr;
    if (o.frame_driver) {
        if (o.raycast == SlabVolume::ExactAllReduce) { fprintf(stderr, "--driver frame: --raycast exact or composite\n"); exit(2); }
        kfx_slab_frame_config fc;
        memset(&fc, 0, sizeof(fc));
        fc.local = *slab.local.abi();
        fc.layout = slab.layout;
        fc.raw = *dMeters.abi(); fc.filtered = *dFiltered.abi(); fc.vbo = *dVbo.abi(); fc.normals = *dNormals.abi();
        fc.ray_depth = *ray_d.abi(); fc.ray_norm = *ray_n.abi(); fc.ray_img = *ray_i.abi();
        fc.K[0] = K.fu; fc.K[1] = K.fv; fc.K[2] = K.u0; fc.K[3] = K.v0;
        fc.bilateral_gs = bigs; fc.bilateral_gr = bigr; fc.bilateral_minval = 0.2f; fc.bilateral_size = biwin;
        fc.near = knear; fc.far = kfar; fc.trunc_dist = trunc_dist; fc.max_w = max_w; fc.mincostheta = mincostheta;
        fc.halo = o.halo == SlabVolume::HaloExchange ? KFX_SLAB_HALO_EXCHANGE : KFX_SLAB_HALO_RECOMPUTE;
        fc.raycast = o.raycast == SlabVolume::Exact ? KFX_SLAB_RAYCAST_EXACT : KFX_SLAB_RAYCAST_COMPOSITE;
        fc.merge = o.merge == SlabVolume::MergeDirect ? KFX_SLAB_MERGE_DIRECT : KFX_SLAB_MERGE_ALLREDUCE;
        fc.inputs = o.broadcast_inputs ? KFX_SLAB_INPUTS_BROADCAST : KFX_SLAB_INPUTS_REPLICATE;
        fc.overlap = (o.overlap || o.pipeline) ? 1 : 0;
        fc.tiles = o.tiles;
        fc.timing_slots = 16;
        if (o.pipeline) {   // the image sets 1 .. D - 1 of the pipelined exact raycast (set 0: ray_d / ray_n / ray_i)
            fc.pipe_depth = o.pipeline;
            for (int k = 1; k < o.pipeline; ++k) {
                set_d.emplace_back(new Image<float, TargetDevice, Manage>(w, h));
                set_n.emplace_back(new Image<float4, TargetDevice, Manage>(w, h));
                set_i.emplace_back(new Image<float, TargetDevice, Manage>(w, h));
                fc.pipe_images[3 * (k - 1)] = *set_d.back()->abi(); fc.pipe_images[3 * (k - 1) + 1] = *set_n.back()->abi(); fc.pipe_images[3 * (k - 1) + 2] = *set_i.back()->abi();
            }
        }
        GpuCheckStatus(kfx_slab_frame_create(&kframe, &fc, comm));
    }
    // every frame's rendered depth image, kept on the device (one row-block per frame): `history` in the output is a checksum over ALL
    // frames' renderings, not only the last one's
    Image<float, TargetDevice, Manage> hist(w, (size_t)h * o.frames);
    const auto keep = [&](int frame, const kfx_image& d) {
        GpuCheckStatus(kfx_memcpy_2d((unsigned char*)hist.ptr + (size_t)frame * h * hist.pitch, hist.pitch, d.ptr, d.pitch, (size_t)w * 4, h, 5, 0));
    };
    if (kframe && o.pipeline) {
        // Pipelined frames: the depth frames are resident before the loop (a blocking upload would synchronise the device every
        // frame), the frames are stepped back to back -- no synchronisation, no barrier -- and the rendering of frame k is picked up
        // when frame k + D is about to take its image set.
        std::vector<std::unique_ptr<Image<float, TargetDevice, Manage> > > dmm(o.frames);
        for (int f = 0; f < o.frames; ++f) {
            dmm[f].reset(new Image<float, TargetDevice, Manage>(w, h));
            dmm[f]->MemcpyFromHost(const_cast<float*>(depth_mm[f].data()));
        }
        kfx_stream_synchronize(0);
        comm->barrier(comm);
        const auto t0 = std::chrono::steady_clock::now();
        const int D = o.pipeline;
        for (int f = 0; f < o.frames; ++f) {
            kfx_image d, n, i;
            if (f >= D) {
                GpuCheckStatus(kfx_slab_frame_wait_frame(kframe, f - D, 0));
                GpuCheckStatus(kfx_slab_frame_images(kframe, f - D, &d, &n, &i));
                keep(f - D, d);
            }
            ElementwiseScaleBias<float,float,float>(dMeters, *dmm[f], 1.0f / 1000.0f);
            const Mat<float,3,4> T_cw = SE3inv(poses[f]);
            GpuCheckStatus(kfx_slab_frame_step(kframe, 0, poses[f].m, T_cw.m, 0, 0));
        }
        GpuCheckStatus(kfx_slab_frame_wait(kframe, 0));   // (a collective point: the trailing final exchanges are enqueued here)
        for (int f = o.frames > D ? o.frames - D : 0; f < o.frames; ++f) {
            kfx_image d, n, i;
            GpuCheckStatus(kfx_slab_frame_images(kframe, f, &d, &n, &i));
            keep(f, d);
            if (f == o.frames - 1) {   // the last rendering, where the checksums below look for it
                GpuCheckStatus(kfx_memcpy_2d(ray_d.ptr, ray_d.pitch, d.ptr, d.pitch, (size_t)w * 4, h, 5, 0));
                GpuCheckStatus(kfx_memcpy_2d(ray_n.ptr, ray_n.pitch, n.ptr, n.pitch, (size_t)w * 16, h, 5, 0));
                GpuCheckStatus(kfx_memcpy_2d(ray_i.ptr, ray_i.pitch, i.ptr, i.pitch, (size_t)w * 4, h, 5, 0));
            }
        }
        GpuCheckStatus(kfx_slab_frame_sync(kframe, 0));
        comm->barrier(comm);
        res->ms_per_frame = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / o.frames;
        res->rounds = kfx_slab_frame_last_steps(kframe);
        kfx_slab_frame_destroy(kframe);
        kframe = nullptr;
    }
    const bool piped = o.frame_driver && o.pipeline;

    double total_ms = 0;
    for (int f = 0; f < o.frames && !piped; ++f) {
        const Mat<float,3,4> T_wl = poses[f];
        dMeters.MemcpyFromHost(const_cast<float*>(depth_mm[f].data()));
        comm->barrier(comm);
        const auto t0 = std::chrono::steady_clock::now();
        if (kframe) {
            // (the sensor's millimetres -> metres is the application's step, main.cpp:208; every rank holds the raw frame)
            ElementwiseScaleBias<float,float,float>(dMeters, dMeters, 1.0f / 1000.0f);
            const Mat<float,3,4> T_cw = SE3inv(T_wl);
            GpuCheckStatus(kfx_slab_frame_step(kframe, 0, T_wl.m, T_cw.m, 0, 0));
            GpuCheckStatus(kfx_slab_frame_sync(kframe, 0));
            comm->barrier(comm);
            total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            keep(f, *ray_d.abi());
            continue;
        }
        if (!o.broadcast_inputs || comm->rank == 0) {   // --inputs broadcast: rank 0 preprocesses, the maps travel to the others
            ElementwiseScaleBias<float,float,float>(dMeters, dMeters, 1.0f / 1000.0f);
            BilateralFilter<float,float>(dFiltered, dMeters, bigs, bigr, biwin, 0.2f);
            DepthToVbo<float>(dVbo, dFiltered, K);
            NormalsFromVbo(dNormals, dVbo);
        }
        if (o.broadcast_inputs) slab.BroadcastInputs(dFiltered, dNormals, 0);
        slab.Fuse(dFiltered, dNormals, SE3inv(T_wl), K, trunc_dist, max_w, mincostheta);
        slab.Raycast(ray_d, ray_n, ray_i, T_wl, K, knear, kfar, trunc_dist, true);
        kfx_stream_synchronize(0);
        comm->barrier(comm);
        total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        keep(f, *ray_d.abi());
    }
    if (!piped) {
        res->ms_per_frame = total_ms / o.frames;
        res->rounds = kframe ? kfx_slab_frame_last_steps(kframe) : slab.last_rounds;
    }
    if (kframe) kfx_slab_frame_destroy(kframe);
    {   // the checksum over every frame's rendered depth
        std::vector<float> hh((size_t)w * h * o.frames);
        kfx_stream_synchronize(0);
        GpuCheckStatus(kfx_memcpy_2d(hh.data(), (size_t)w * 4, hist.ptr, hist.pitch, (size_t)w * 4, (size_t)h * o.frames, 2, 0));
        for (float& d : hh) if (!std::isfinite(d)) d = -1.0f;
        res->chk_hist = BitSum(hh.data(), hh.size() * 4);
    }

    // checksums: images (identical on every rank after the merge) and the owned planes of the volume, summed over the ranks
    std::vector<float> hd((size_t)w * h), hi((size_t)w * h);
    std::vector<float4> hn((size_t)w * h);
    ray_d.MemcpyToHost(hd.data());
    ray_n.MemcpyToHost(hn.data());
    ray_i.MemcpyToHost(hi.data());
    for (float d : hd) res->hits += std::isfinite(d) ? 1 : 0;
    for (float& d : hd) if (!std::isfinite(d)) d = -1.0f;   // NaN payload bits are not part of the contract
    res->chk_d = BitSum(hd.data(), hd.size() * 4);
    res->chk_n = BitSum(hn.data(), hn.size() * 16);
    res->chk_i = BitSum(hi.data(), hi.size() * 4);
    const kfx_slab_layout& L = slab.layout;
    const size_t plane = slab.local.img_pitch, own = L.z1 - L.z0;
    std::vector<unsigned char> hv(plane * own);
    GpuCheckStatus(kfx_memcpy_2d(hv.data(), plane, (unsigned char*)slab.local.ptr + (L.z0 - L.s0) * plane, plane, plane, own, 2, 0));
    unsigned vs_sum = 0;
    for (size_t z = 0; z < own; ++z)
        for (size_t y = 0; y < slab.local.h; ++y) {
            const unsigned* row = reinterpret_cast<const unsigned*>(hv.data() + z * plane + y * slab.local.pitch);
            for (size_t x = 0; x < 2 * slab.local.w; ++x) {
                unsigned bits = row[x];
                if ((bits & 0x7fffffffu) > 0x7f800000u) bits = 0x7fc00000u;   // any NaN
                vs_sum += bits * 2654435761u + (unsigned)((L.z0 + z) * 7919u + y * 104729u + x);
            }
        }
    void* dsum = nullptr;
    size_t pitch;
    GpuCheckStatus(kfx_alloc_pitched(&dsum, &pitch, 64, 1));
    int hsum[2] = {(int)vs_sum, 0};
    GpuCheckStatus(kfx_memcpy_2d(dsum, 64, hsum, 8, 8, 1, 1, 0));
    GpuCheckStatus(comm->all_reduce(comm, dsum, 2, KFX_COMM_SUM_I32, 0));
    GpuCheckStatus(kfx_memcpy_2d(hsum, 8, dsum, 64, 8, 1, 2, 0));
    kfx_free(dsum);
    res->chk_vol = (unsigned)hsum[0];
}

int main(int argc, char** argv)
{
    Options o;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--res") && i + 1 < argc) o.volres = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--frames") && i + 1 < argc) o.frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--width") && i + 1 < argc) o.w = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--height") && i + 1 < argc) o.h = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--ranks") && i + 1 < argc) o.ranks = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--fast")) o.fast = true;
        else if (!strcmp(argv[i], "--transport") && i + 1 < argc) { ++i; o.rccl = !strcmp(argv[i], "rccl"); o.p2p = !strcmp(argv[i], "threads-p2p"); }
        else if (!strcmp(argv[i], "--pipeline") && i + 1 < argc) o.pipeline = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--halo") && i + 1 < argc) o.halo = !strcmp(argv[++i], "recompute") ? SlabVolume::HaloRecompute : SlabVolume::HaloExchange;
        else if (!strcmp(argv[i], "--raycast") && i + 1 < argc) {
            ++i;
            o.raycast = !strcmp(argv[i], "exact") ? SlabVolume::Exact : (!strcmp(argv[i], "exact-allreduce") ? SlabVolume::ExactAllReduce : SlabVolume::Composite);
        }
        else if (!strcmp(argv[i], "--merge") && i + 1 < argc) o.merge = !strcmp(argv[++i], "allreduce") ? SlabVolume::MergeAllReduce : SlabVolume::MergeDirect;
        else if (!strcmp(argv[i], "--inputs") && i + 1 < argc) o.broadcast_inputs = !strcmp(argv[++i], "broadcast");
        else if (!strcmp(argv[i], "--rendezvous") && i + 1 < argc) o.rendezvous = argv[++i];
        else if (!strcmp(argv[i], "--driver") && i + 1 < argc) o.frame_driver = !strcmp(argv[++i], "frame");
        else if (!strcmp(argv[i], "--tiles") && i + 1 < argc) o.tiles = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--ghost") && i + 1 < argc) { ++i; o.ghost = !strcmp(argv[i], "auto") ? -1 : atoi(argv[i]); }
        else if (!strcmp(argv[i], "--overlap")) o.overlap = true;
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    const int ndev = kfx_device_count();
    if (ndev < 1) { fprintf(stderr, "no HIP device\n"); return 2; }
    kfx_set_math_mode(o.fast ? KFX_MATH_FAST : KFX_MATH_EXACT);

    const double depth_focal = o.w * 570.342 / 640.0;
    const ImageIntrinsics K(depth_focal, depth_focal, o.w / 2.0 - 0.5, o.h / 2.0 - 0.5);
    std::vector<std::vector<float> > depth_mm(o.frames);
    std::vector<Mat<float,3,4> > poses(o.frames);
    for (int f = 0; f < o.frames; ++f) {
        poses[f] = OrbitPose(f, 30);
        RenderRoom(depth_mm[f], o.w, o.h, poses[f], K);
        for (float& d : depth_mm[f]) d *= 1000.0f;
    }

    Result r0;
    int world = o.ranks, rank = 0;
    if (o.rccl) {
        const char* er = getenv("RANK"); const char* ew = getenv("WORLD_SIZE"); const char* el = getenv("LOCAL_RANK");
        rank = er ? atoi(er) : 0;
        world = ew ? atoi(ew) : 1;
        const int local = el ? atoi(el) : rank;
        if (world > ndev) { fprintf(stderr, "kinectfusion_slabs: %d ranks need %d GPUs, this node shows %d\n", world, world, ndev); return 2; }
        GpuCheckStatus(kfx_set_device(local % ndev));
        if (o.rendezvous.empty()) {   // unique per user and launch: the launcher's run id / port, else the parent (launcher) process
            const char* id = getenv("TORCHELASTIC_RUN_ID");
            if (!id || !*id) id = getenv("MASTER_PORT");
            o.rendezvous = "/tmp/kfx_slabs." + std::to_string((long long)getuid()) + "." + (id && *id ? std::string(id) : "ppid" + std::to_string((long long)getppid())) + ".id";
        }
        kfx_comm comm;
        const int st = kfx_comm_create_rccl(&comm, rank, world, o.rendezvous.c_str(), 120);
        if (st != 0) { fprintf(stderr, "kfx_comm_create_rccl failed: %d\n", st); return 3; }
        RunRank(o, &comm, depth_mm, poses, &r0);
        comm.destroy(&comm);
    } else {
        std::vector<kfx_comm> comms(world);
        if (o.p2p) GpuCheckStatus(kfx_comm_create_threads_p2p(comms.data(), world, 20000));
        else GpuCheckStatus(kfx_comm_create_threads(comms.data(), world));
        std::vector<Result> results(world);
        std::vector<std::thread> threads;
        for (int r = 1; r < world; ++r) threads.emplace_back(RunRank, std::cref(o), &comms[r], std::cref(depth_mm), std::cref(poses), &results[r]);
        RunRank(o, &comms[0], depth_mm, poses, &results[0]);
        for (auto& t : threads) t.join();
        r0 = results[0];
        for (int r = 1; r < world; ++r)   // after the merge every rank must hold the same images
            if (results[r].chk_d != r0.chk_d || results[r].chk_n != r0.chk_n || results[r].chk_i != r0.chk_i || results[r].chk_hist != r0.chk_hist) r0.status = 4;
        comms[0].destroy(&comms[0]);
    }
    if (rank == 0) {
        printf("kinectfusion_slabs: %d^3 volume in %d slab(s) [%s%s], %dx%d, %d frames, %s math, halo %s, raycast %s%s: %.3f ms/frame (%.1f fps)\n",
               o.volres, world, o.rccl ? "RCCL, one process per GPU" : "threads sharing one GPU", o.frame_driver ? "; one kfx_slab_frame_step per frame" : "", o.w, o.h, o.frames, o.fast ? "fast" : "exact",
               o.halo == SlabVolume::HaloExchange ? "exchange" : "recompute", o.raycast == SlabVolume::Exact ? "exact (hand-over)" : (o.raycast == SlabVolume::ExactAllReduce ? "exact (all-reduce per round)" : (o.merge == SlabVolume::MergeDirect ? "composite (direct-send merge)" : "composite (all-reduce merge)")),
               o.raycast != SlabVolume::Composite ? (" (" + std::to_string(r0.rounds) + " rounds)").c_str() : "", r0.ms_per_frame, 1e3 / r0.ms_per_frame);
        printf("checksums depth=%08x norm=%08x img=%08x volume=%08x history=%08x hits=%zu ranks_agree=%d%s\n", r0.chk_d, r0.chk_n, r0.chk_i, r0.chk_vol, r0.chk_hist,
               r0.hits, r0.status == 0 ? 1 : 0, o.pipeline ? (" pipeline=" + std::to_string(o.pipeline)).c_str() : "");
    }
    if (r0.status) return r0.status;
    return r0.hits > (size_t)(o.w * o.h) / 4 ? 0 : 1;
}
