// kinectfusion_headless.cpp -- the KinectFusion frame loop of the reference application
// (applications/kinectfusion/main.cpp:190-393) without its GUI and sensor: a synthetic camera orbits a
// synthetic room, and every frame runs the same roo:: calls in the same order on the same container
// types (poses: known by default; --track estimates them with the projective ICP of main.cpp:299-343):
//
//   ElementwiseScaleBias (mm -> m) -> BilateralFilter -> BoxReduceIgnoreInvalid -> per level
//   DepthToVbo -> NormalsFromVbo                                            (main.cpp:208-215)
//   [first frame] SdfReset(vol, NaN); SdfFuse                              (main.cpp:224-242)
//   roi = BoundingBox(T_wl, w, h, K, knear, kfar); work_vol = vol.SubBoundingVolume(roi)   (:275-276)
//   for levels with its[l] > 0: RaycastSdf(ray_d[l], ray_n[l], ray_i[l], work_vol, T_wl, K[l], ...);
//                               DepthToVbo(ray_v[l], ray_d[l], K[l])                       (:280-288)
//   [--track] for l = MaxLevels-1 .. 0, its[l] times: lss = PoseRefinementProjectiveIcpPointPlane(kin_v[l],
//             ray_v[l], ray_n[l], K[l]*T_lp, T_lp^-1, icp_c, dScratch, dDebug); solve; T_lp *= exp(x)  (:301-337)
//   SdfFuse(work_vol, kin_d, kin_n, T_wl^-1, K, trunc_dist, max_w, mincostheta)            (:345-356)
//
// Host code only; all device work happens in libkfx behind the roo:: wrappers.
// Usage: kinectfusion_headless [--res N] [--frames F] [--warmup F] [--width W] [--height H] [--fast] [--track | --device-icp] [--fused-launches] [--summary | --summary-auto]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include <kangaroo/kangaroo.h>
#include <kangaroo/SdfSummary.h>

#include "pose_solve.h"

using namespace roo;

// analytic depth of the synthetic room (box interior x,y in [-0.9,0.9], back wall z = 3.8, sphere
// c = (0,0,3) r = 0.5), z-depth along rays with ray_c.z = 1 -- same scene as kangaroo_amd/scenes.py
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

int main(int argc, char** argv)
{
    int volres = 256, frames = 30, w = 640, h = 480;   // the application's defaults (main.cpp:90-91)
    bool fast = false, track = false, device_icp = false, one_raycast = false, use_summary = false, summary_auto = false;
    int warm = 0;          // --warmup F: the first F frames run but do not count in the reported frame time (clocks, first launches)
    int drop_frame = -1;   // --drop-frame F: frame F arrives with no valid depth at all (a sensor drop-out): tracking is lost, the next frame recovers
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--res") && i + 1 < argc) volres = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--width") && i + 1 < argc) w = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--height") && i + 1 < argc) h = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--fast")) fast = true;
        else if (!strcmp(argv[i], "--track")) track = true;
        else if (!strcmp(argv[i], "--summary")) use_summary = true;   // roo::SdfSummary: SdfFuse keeps it current, RaycastSdf marches through its class tables
        else if (!strcmp(argv[i], "--summary-auto")) use_summary = summary_auto = true;   // ... and the application keeps it only if it pays (whole frames timed with and without it)
        else if (!strcmp(argv[i], "--one-raycast") || !strcmp(argv[i], "--fused-launches")) one_raycast = true;   // additions beside the reference API: all pyramid levels rendered by one launch, vbo + normals in one launch
        else if (!strcmp(argv[i], "--device-icp")) track = device_icp = true;   // the refinement loop as one device-side chain
        else if (!strcmp(argv[i], "--drop-frame") && i + 1 < argc) drop_frame = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) warm = atoi(argv[++i]);
    }
    if (warm < 0 || warm >= frames) warm = 0;
    if (kfx_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 2; }
    kfx_set_math_mode(fast ? KFX_MATH_FAST : KFX_MATH_EXACT);

    // generic camera model based on image dimensions (main.cpp:63-64)
    const double depth_focal = w * 570.342 / 640.0;
    const ImageIntrinsics K(depth_focal, depth_focal, w / 2.0 - 0.5, h / 2.0 - 0.5);
    const float knear = 0.4f, kfar = 4.0f;              // main.cpp:80-81
    const float bigs = 1.5f, bigr = 0.1f; const int biwin = 3;   // main.cpp:149-151
    const float trunc_dist_factor = 2.0f, max_w = 1000.0f, mincostheta = 0.1f;  // main.cpp:155-158
    const BoundingBox reset_bb(make_float3(-1, -1, 2), make_float3(1, 1, 4));

    const int MaxLevels = 4;
    const int its[] = {1, 0, 2, 3};                      // main.cpp:51-52
    // The frame's pre-amble images, twice: with --device-icp --fused-launches the NEXT frame's pre-amble is enqueued while this
    // frame's pose is on its way to the host (kfx_icp_refine_then) and must not overwrite the maps SdfFuse is about to read.
    struct Preamble {
        Image<float, TargetDevice, Manage> dKinectMeters;
        Pyramid<float, MaxLevels, TargetDevice, Manage> kin_d;
        Pyramid<float4, MaxLevels, TargetDevice, Manage> kin_v, kin_n;
        Preamble(int w, int h) : dKinectMeters(w, h), kin_d(w, h), kin_v(w, h), kin_n(w, h) {}
    };
    Preamble pre_a(w, h), pre_b(w, h);
    Preamble *pre = &pre_a, *pre_next = &pre_b;
    int prepared_frame = -1;   // the frame whose pre-amble `pre_next` holds
    Pyramid<float, MaxLevels, TargetDevice, Manage> ray_i(w, h), ray_d(w, h);
    Pyramid<float4, MaxLevels, TargetDevice, Manage> ray_n(w, h), ray_v(w, h);
    BoundedVolume<SDF_t, TargetDevice, Manage> vol(volres, volres, volres, reset_bb);
    Image<float4, TargetDevice, Manage> dDebug(w, h);                                             // main.cpp:110
    Image<unsigned char, TargetDevice, Manage> dScratch(w * sizeof(LeastSquaresSystem<float,12>), h);  // main.cpp:111
    const float icp_c = 0.1f, max_rmse = 0.10f;                                                   // main.cpp:154,162

    std::unique_ptr<SdfSummary> summary;
    // --summary-auto: the policy of FramePipeline (kangaroo_amd/pipeline.py) in the application's terms.  Three blocks of
    // CAL_BLOCK whole frames from frame CAL_FIRST on -- tracked pair of kernels, plain pair (the summary goes stale), tracked
    // pair again (summary rebuilt from the volume) --, each frame timed by the host clock between the loop's own
    // synchronisations; the tables stay only if BOTH tracked blocks beat the plain block's median frame by 5 %.
    const int CAL_FIRST = 4, CAL_BLOCK = 8;
    std::vector<double> cal_ms;
    bool cal_done = !summary_auto;
    if (use_summary) summary.reset(new SdfSummary(vol));
    const float3 vs = vol.VoxelSizeUnits();
    const float trunc_dist = trunc_dist_factor * length(vs);   // main.cpp:221

    std::vector<std::vector<float> > depth_frames(frames);
    std::vector<Mat<float,3,4> > poses(frames);
    for (int f = 0; f < frames; ++f) {
        poses[f] = OrbitPose(f, 30);
        RenderRoom(depth_frames[f], w, h, poses[f], K);
        for (float& d : depth_frames[f]) d *= 1000.0f;   // the sensor delivers millimetres (main.cpp:208)
        if (f == drop_frame) for (float& d : depth_frames[f]) d = std::numeric_limits<float>::quiet_NaN();
    }

    // the sensor's frames (millimetres), resident in device memory before the loop starts: the timed region begins with its inputs in
    // HBM (the application's per-frame host -> device copy, main.cpp:203, is not part of the path measured here)
    std::vector<std::unique_ptr<Image<float, TargetDevice, Manage> > > dKinect(frames);
    for (int f = 0; f < frames; ++f) {
        dKinect[f].reset(new Image<float, TargetDevice, Manage>(w, h));
        dKinect[f]->MemcpyFromHost(depth_frames[f].data());
        std::vector<float>().swap(depth_frames[f]);
    }
    std::vector<float> hdepth((size_t)w * h);
    std::chrono::steady_clock::time_point t_timed = std::chrono::steady_clock::now();   // start of the timed region (frame `warm`)
    double total_ms = 0, worst_pos_err = 0, rmse = 0;
    size_t hits = 0;
    unsigned long long depth_sum = 1469598103934665603ull;
    int lost = 0, resets = 0;
    posesolve::SE3d T_anchor;   // the world frame of the estimate in the frame of the known poses (identity until tracking is lost and the model is reset)
    posesolve::SE3d T_wl_est;   // tracked pose (double, as Sophus::SE3d in the application)
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T_wl_est.R[i][j] = poses[0](i, j);
        T_wl_est.t[i] = poses[0](i, 3);
    }
    for (int f = 0; f < frames; ++f) {
        Mat<float,3,4> T_wl = track ? T_wl_est.matrix3x4<Mat<float,3,4> >() : poses[f];
        const bool ahead = track && device_icp && one_raycast;
        if (prepared_frame == f) std::swap(pre, pre_next);   // its pre-amble ran under the previous frame's pose read-back
        Image<float, TargetDevice, Manage>& dKinectMeters = pre->dKinectMeters;
        Pyramid<float, MaxLevels, TargetDevice, Manage>& kin_d = pre->kin_d;
        Pyramid<float4, MaxLevels, TargetDevice, Manage> &kin_v = pre->kin_v, &kin_n = pre->kin_n;
        const auto preamble = [&](Preamble& q, int fr) {
            ElementwiseScaleBias<float,float,float>(q.dKinectMeters, *dKinect[fr], 1.0f / 1000.0f);   // main.cpp:208 (sensor image -> metres)
            BilateralFilter<float,float>(q.kin_d[0], q.dKinectMeters, bigs, bigr, biwin, 0.2f);
            if (one_raycast) {   // --fused-launches: the pyramid and both maps of every level from one launch, same images
                DepthPyramidVboNormals<MaxLevels>(q.kin_d, q.kin_v, q.kin_n, K);
            } else {
                BoxReduceIgnoreInvalid<float,MaxLevels,float>(q.kin_d);
                for (int l = 0; l < MaxLevels; ++l) {
                    DepthToVbo<float>(q.kin_v[l], q.kin_d[l], K[l]);
                    NormalsFromVbo(q.kin_n[l], q.kin_v[l]);
                }
            }
        };
        const auto t0 = std::chrono::steady_clock::now();
        if (!cal_done && summary) {   // the block this frame belongs to
            const int k = f - CAL_FIRST;
            if (k == CAL_BLOCK) use_summary = false;
            else if (k == 2 * CAL_BLOCK) { summary->Rebuild(); use_summary = true; }
        }
        if (prepared_frame != f) preamble(*pre, f);
        (void)dKinectMeters;
        // main.cpp:223-242: `if (Pushed(reset) || !std::isfinite(f_rmse))` -- the first frame, and the frame after tracking was lost
        // altogether (no correspondence left: rmse = sqrt(0 / 0)): the world frame restarts at the current camera (T_wl = SE3d()), the
        // model is reset to "never observed" and the current frame is fused; the frame then goes on like any other (it is tracked
        // against the model it has just founded and fused again, as in the reference's loop).
        const bool recover = track && f > 0 && !std::isfinite(rmse);
        if (recover) {
            for (int i = 0; i < 3; ++i) {   // where the new world frame sits in the frame of the known poses: at this frame's true camera
                for (int j = 0; j < 3; ++j) T_anchor.R[i][j] = poses[f](i, j);
                T_anchor.t[i] = poses[f](i, 3);
            }
            T_wl_est = posesolve::SE3d();
            T_wl = T_wl_est.matrix3x4<Mat<float,3,4> >();
            rmse = 0;
            ++resets;
        }
        if (f == 0 || recover) {
            if (use_summary) {
                SdfReset(vol, std::numeric_limits<float>::quiet_NaN(), *summary);
                SdfFuse(vol, *summary, kin_d[0], kin_n[0], SE3inv(T_wl), K, trunc_dist, max_w, mincostheta);
            } else {
                SdfReset(vol, std::numeric_limits<float>::quiet_NaN());
                SdfFuse(vol, kin_d[0], kin_n[0], SE3inv(T_wl), K, trunc_dist, max_w, mincostheta);
            }
        }
        const BoundingBox roi(T_wl, w, h, K, knear, kfar);
        BoundedVolume<SDF_t> work_vol = vol.SubBoundingVolume(roi);
        if (work_vol.IsValid()) {
            const auto render = [&](bool use_summary) {   // (the parameter shadows the setting: the calibration frames render both ways)
            if (one_raycast) {   // the same images from one launch: the levels' marches overlap (kfx_raycast_sdf_levels)
                Image<float> rd[MaxLevels], ri[MaxLevels];
                Image<float4> rn[MaxLevels], rv[MaxLevels];
                ImageIntrinsics Kl[MaxLevels];
                unsigned n = 0;
                for (int l = 0; l < MaxLevels; ++l)
                    if (its[l] > 0) { rd[n] = ray_d[l]; rn[n] = ray_n[l]; ri[n] = ray_i[l]; rv[n] = ray_v[l]; Kl[n] = K[l]; ++n; }
                if (use_summary) RaycastSdfLevels(rd, rn, ri, n, work_vol, *summary, T_wl, Kl, knear, kfar, trunc_dist, true, rv);
                else RaycastSdfLevels(rd, rn, ri, n, work_vol, T_wl, Kl, knear, kfar, trunc_dist, true, rv);   // rv[l] = DepthToVbo(rd[l], K[l])
            } else {
                for (int l = 0; l < MaxLevels; ++l) {
                    if (its[l] > 0) {
                        const ImageIntrinsics Kl = K[l];
                        if (use_summary) RaycastSdf(ray_d[l], ray_n[l], ray_i[l], work_vol, *summary, T_wl, Kl, knear, kfar, trunc_dist, true);
                        else RaycastSdf(ray_d[l], ray_n[l], ray_i[l], work_vol, T_wl, Kl, knear, kfar, trunc_dist, true);
                        DepthToVbo<float>(ray_v[l], ray_d[l], Kl);
                    }
                }
            }
            };
            render(use_summary);
            bool tracking_good = true;
            if (track && f > 0) {   // main.cpp:299-341
                posesolve::SE3d T_lp;
                if (device_icp) {   // kfx_icp_refine: all iterations and 6x6 solves on the GPU, one synchronisation
                    kfx_icp_level lv[MaxLevels];
                    for (int s = 0; s < MaxLevels; ++s) {
                        const int l = MaxLevels - 1 - s;
                        const ImageIntrinsics Kl = K[l];
                        lv[s].Pl = *kin_v[l].abi(); lv[s].Pr = *ray_v[l].abi(); lv[s].Nr = *ray_n[l].abi();
                        lv[s].K[0] = Kl.fu; lv[s].K[1] = Kl.fv; lv[s].K[2] = Kl.u0; lv[s].K[3] = Kl.v0;
                        lv[s].iterations = its[l];
                        lv[s].rotation_only = (l == MaxLevels - 1 && MaxLevels > 1) ? 1 : 0;
                    }
                    double T34[12];
                    float r = 0;
                    unsigned nobs = 0;
                    int good = 1;
                    if (ahead && f + 1 < frames) {   // the next frame's pre-amble runs while this thread waits for the pose
                        struct Hook { decltype(preamble)* fn; Preamble* q; int fr; } hook{&preamble, pre_next, f + 1};
                        GpuCheckStatus(kfx_icp_refine_then(lv, MaxLevels, icp_c, max_rmse, dScratch.abi(), dDebug.abi(), T34, &r, &nobs, &good,
                                                           [](void* u) { Hook* k = static_cast<Hook*>(u); (*k->fn)(*k->q, k->fr); }, &hook, 0));
                        prepared_frame = f + 1;
                    } else {
                        GpuCheckStatus(kfx_icp_refine(lv, MaxLevels, icp_c, max_rmse, dScratch.abi(), dDebug.abi(), T34, &r, &nobs, &good, 0));
                    }
                    for (int i = 0; i < 3; ++i) {
                        for (int j = 0; j < 3; ++j) T_lp.R[i][j] = T34[i * 4 + j];
                        T_lp.t[i] = T34[i * 4 + 3];
                    }
                    rmse = r;
                    tracking_good = good != 0;
                }
                for (int l = MaxLevels - 1; l >= 0 && !device_icp; --l) {
                    const ImageIntrinsics Kl = K[l];
                    for (int i = 0; i < its[l]; ++i) {
                        Mat<float,3,4> KT_lp, T_pl = T_lp.inverse().matrix3x4<Mat<float,3,4> >();
                        for (int c = 0; c < 4; ++c) {   // K * [R | t] in double, then to float
                            const double r0 = c < 3 ? T_lp.R[0][c] : T_lp.t[0], r1 = c < 3 ? T_lp.R[1][c] : T_lp.t[1],
                                         r2 = c < 3 ? T_lp.R[2][c] : T_lp.t[2];
                            KT_lp(0, c) = (float)((double)Kl.fu * r0 + (double)Kl.u0 * r2);
                            KT_lp(1, c) = (float)((double)Kl.fv * r1 + (double)Kl.v0 * r2);
                            KT_lp(2, c) = (float)r2;
                        }
                        const LeastSquaresSystem<float,6> lss = PoseRefinementProjectiveIcpPointPlane(
                            kin_v[l], ray_v[l], ray_n[l], KT_lp, T_pl, icp_c, dScratch, dDebug.SubImage(0, 0, w >> l, h >> l));
                        const Mat<double,6,6> JTJf = lss.JTJ;
                        double sysJTJ[36], sysJTy[6], x[6] = {0, 0, 0, 0, 0, 0};
                        for (int a = 0; a < 36; ++a) sysJTJ[a] = JTJf.m[a];
                        for (int a = 0; a < 6; ++a) { sysJTy[a] = lss.JTy.m[a]; sysJTJ[a * 7] += 0.1 / 0.2; }  // weak pose prior
                        rmse = sqrt(lss.sqErr / lss.obs);
                        tracking_good = rmse < max_rmse;
                        if (l == MaxLevels - 1 && MaxLevels > 1) {   // coarsest level: rotation only
                            double A3[9], b3[3];
                            for (int a = 0; a < 3; ++a) {
                                b3[a] = sysJTy[3 + a];
                                for (int b = 0; b < 3; ++b) A3[a * 3 + b] = sysJTJ[(3 + a) * 6 + 3 + b];
                            }
                            posesolve::FullPivLuSolve<3>(A3, b3, x + 3);
                            for (int a = 3; a < 6; ++a) x[a] = -x[a];
                            T_lp = T_lp * posesolve::Exp(x, true);
                        } else {
                            posesolve::FullPivLuSolve<6>(sysJTJ, sysJTy, x);
                            bool finite = true;
                            for (int a = 0; a < 6; ++a) { x[a] = -x[a]; finite = finite && std::isfinite(x[a]); }
                            if (finite) T_lp = T_lp * posesolve::Exp(x);
                        }
                    }
                }
                if (tracking_good) T_wl_est = T_wl_est * T_lp.inverse();
                else ++lost;
                T_wl = T_wl_est.matrix3x4<Mat<float,3,4> >();
                const posesolve::SE3d T_abs = T_anchor * T_wl_est;   // the estimate in the frame of the known poses
                double e = 0;
                for (int i = 0; i < 3; ++i) e += (T_abs.t[i] - poses[f](i, 3)) * (T_abs.t[i] - poses[f](i, 3));
                if (f != drop_frame) worst_pos_err = std::fmax(worst_pos_err, std::sqrt(e));   // (a frame without depth has no estimate)
            }
            if (f > 0 && tracking_good) {
                if (use_summary) SdfFuse(work_vol, *summary, kin_d[0], kin_n[0], SE3inv(T_wl), K, trunc_dist, max_w, mincostheta);
                else SdfFuse(work_vol, kin_d[0], kin_n[0], SE3inv(T_wl), K, trunc_dist, max_w, mincostheta);
            }
        }
        // The reported frame time is the wall clock over the frames from `warm` on, between two synchronisations.  The loop itself
        // synchronises only where the application must -- the pose's way to the host (--track: every ICP iteration; --device-icp: once
        // per frame) -- and where this program needs a frame's own time: the --summary-auto blocks, the frame before the timed region
        // and the last one.  (Round 5 synchronised after every frame: the device then idles while the host wakes up and issues the next
        // frame's first launch, 0.02 ms of a 0.6 ms frame that the Python loop, which never did, was ahead by.)
        const bool sync_now = !cal_done || f + 1 == warm || f + 1 == frames || f < warm;
        if (sync_now) kfx_stream_synchronize(0);
        const auto t1 = std::chrono::steady_clock::now();
        if (f + 1 == warm) t_timed = t1;
        if (f == 0 && warm == 0) t_timed = t0;
        const double frame_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        if (!cal_done && f >= CAL_FIRST) {
            cal_ms.push_back(frame_ms);
            if ((int)cal_ms.size() == 3 * CAL_BLOCK) {   // decide once
                const auto median = [&](int first) { std::vector<double> v(cal_ms.begin() + first, cal_ms.begin() + first + CAL_BLOCK);
                                                     std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
                const double t1 = median(0), plain = median(CAL_BLOCK), t2 = median(2 * CAL_BLOCK);
                const bool keep = std::fmax(t1, t2) <= 0.95 * plain;
                printf("  --summary-auto: frame with the tables %.3f / %.3f ms, plain kernels %.3f ms -> %s march\n", t1, t2, plain, keep ? "table" : "plain");
                use_summary = keep;
                if (!keep) summary.reset();
                cal_done = true;
            }
        }
        if (f == frames - 1) total_ms = std::chrono::duration<double, std::milli>(t1 - t_timed).count();
        if (f == frames - 1) {
            ray_d[0].MemcpyToHost(hdepth.data());
            for (float d : hdepth) hits += std::isfinite(d) ? 1 : 0;
            for (float d : hdepth) {   // FNV-1a over the bit patterns of the last rendering (tests compare runs by it)
                unsigned u;
                memcpy(&u, &d, 4);
                depth_sum = (depth_sum ^ u) * 1099511628211ull;
            }
        }
    }
    printf("kinectfusion_headless: %d^3 volume, %dx%d, %d frames, %s math, %s poses: %.3f ms/frame (%.1f fps), last raycast hits %zu/%d%s, depth checksum %016llx\n",
           volres, w, h, frames, fast ? "fast" : "exact", device_icp ? "ICP-tracked (device loop)" : (track ? "ICP-tracked" : "known"), total_ms / (frames - warm), 1e3 * (frames - warm) / total_ms, hits, w * h,
           use_summary ? " (brick summary)" : "", depth_sum);
    if (track) printf("  tracking: worst position error %.2f mm over the orbit (step between poses up to %.1f mm), final rmse %.4f, %d frames lost, %d resets\n",
                      1e3 * worst_pos_err, 1e3 * 0.0105, rmse, lost, resets);
    const int expect_lost = (track && drop_frame > 0 && drop_frame < frames) ? 1 : 0;
    if (track && (lost != expect_lost || resets != (expect_lost && drop_frame + 1 < frames ? 1 : 0) || worst_pos_err > 0.02)) return 1;
    return hits > (size_t)(w * h) / 4 ? 0 : 1;
}
