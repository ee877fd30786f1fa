// icp.hip -- projective point-to-plane ICP normal equations (roo::PoseRefinementProjectiveIcpPointPlane)
// for gfx950.  SURVEY.md 8(f) row f-2.
//
// Reference behaviour: src/cu_model_refinement.cu:541-608 (per-pixel residual / Jacobian / Tukey weight),
// LeastSquareSum.h:42-86 (per-block tree reduction in shared memory, then thrust::reduce over the blocks),
// Mat.h:353-545 (SymMat / LeastSquaresSystem: JTy[6], JTJ[21] lower triangle row-major, sqErr, obs).
//
// New kernel: a pixel's 29-word system lives in registers, not in a 116-byte LDS slot per thread.  The
// reference's tree (s[t] += s[t+S], S = n/2 .. 1) is reproduced with the same association: the levels
// S >= 64 cross wavefronts through a structure-of-arrays LDS buffer (conflict-free: word k of lane l at
// k*S + l), the levels S < 64 are wave64 lane shifts -- so a block's sum is bit-identical to the
// reference's.  The sum over blocks (thrust::reduce: order unspecified in the reference) is a fixed order
// here: thread t of one 256-thread group adds blocks t, t+256, ... in turn, then the same tree.
#include <cstdlib>
#include <string.h>

#include "kfx_device.h"

namespace kfx {

constexpr int LSS_WORDS = 29; // 6 JTy + 21 JTJ + sqErr + obs

struct Lss {
    float f[28]; // JTy[0..5], JTJ[0..20], sqErr
    unsigned obs;
};

__device__ __forceinline__ void lss_zero(Lss& s)
{
#pragma unroll
    for (int i = 0; i < 28; ++i) s.f[i] = 0.f;
    s.obs = 0u;
}

// s[t] += s[t+S] for S = n/2 .. 1 (LeastSquareSum.h:71-85); the result is valid in thread 0
__device__ __forceinline__ void lss_tree(Lss& s, const int tid, const int n, float* lds)
{
    for (int S = n >> 1; S >= 64; S >>= 1) {
        __syncthreads();
        if (tid >= S && tid < 2 * S) {
#pragma unroll
            for (int k = 0; k < 28; ++k) lds[k * S + (tid - S)] = s.f[k];
            lds[28 * S + (tid - S)] = __uint_as_float(s.obs);
        }
        __syncthreads();
        if (tid < S) {
#pragma unroll
            for (int k = 0; k < 28; ++k) s.f[k] += lds[k * S + tid];
            s.obs += __float_as_uint(lds[28 * S + tid]);
        }
    }
    if (tid < 64) {
        for (int S = (n >> 1) < 32 ? (n >> 1) : 32; S > 0; S >>= 1) {
#pragma unroll
            for (int k = 0; k < 28; ++k) s.f[k] += __shfl_down(s.f[k], S, 64);
            s.obs += (unsigned)__shfl_down((int)s.obs, S, 64);
        }
    }
}

__device__ __forceinline__ void lss_store(float* dst, const Lss& s)
{
#pragma unroll
    for (int k = 0; k < 28; ++k) dst[k] = s.f[k];
    dst[28] = __uint_as_float(s.obs);
}

// A block system for other workgroups of the same launch to read: device-scope atomic stores go to the device's coherence point,
// past the XCD's L2 (the eight L2s are not coherent with each other); they have arrived when the wave's memory counter has drained.
__device__ __forceinline__ void lss_store_agent(float* dst, const Lss& s)
{
#pragma unroll
    for (int k = 0; k < 28; ++k) __hip_atomic_store(dst + k, s.f[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 28, __uint_as_float(s.obs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct IcpParams {
    ImgView Pl, Pr, Nr;
    unsigned char* dbg; // may be null
    size_t dbg_pitch;
    Pose KT_lr, T_rl;
    const float* dev_pose; // device-resident loop: KT_lr[12] then T_rl[12] live here instead (may be null)
    float c;
    float* sums;        // gridDim.x * gridDim.y systems of 29 words
};

// one pixel's contribution (cu_model_refinement.cu:541-590) and its debug colour
__device__ __forceinline__ void icp_pixel(Lss& sum, float4& dbg, const ImgView& Pl_img, const ImgView& Pr_img, const ImgView& Nr_img, const Pose& KT_lr,
                                          const Pose& T_rl, const float c, const unsigned u, const unsigned v)
{
    const float4 Pr = row<float4>(Pr_img, v)[u];
    const float4 Nr = row<float4>(Nr_img, v)[u];
    const V3 KPl = se3_mul(KT_lr, v3(Pr.x, Pr.y, Pr.z));
    const float plx = KPl.x / KPl.z, ply = KPl.y / KPl.z;
    // Image::InBounds(pl, 3): border <= x && x < (w - border) with w converted to float (Image.h:288-291)
    if (isfinite(Pr.z) && Nr.w == 1.0f && 3.0f <= plx && plx < ((float)Pl_img.w - 3.0f) && 3.0f <= ply && ply < ((float)Pl_img.h - 3.0f)) {
        // GetNearestNeighbour: Get(u + 0.5, v + 0.5) -- the sum is a double, truncated to int (Image.h:337-340)
        const int nx = (int)((double)plx + 0.5), ny = (int)((double)ply + 0.5);
        const float4 Pl = row<float4>(Pl_img, ny)[nx];
        if (isfinite(Pl.z)) {
            const V3 _Pr = se3_mul(T_rl, v3(Pl.x, Pl.y, Pl.z));
            const V3 Dr = v3(_Pr.x - Pr.x, _Pr.y - Pr.y, _Pr.z - Pr.z);
            const V3 N = v3(Nr.x, Nr.y, Nr.z);
            const float y = dot(Dr, N);
            // -dot(SE3gen_i * _Pr, Nr): the generators' zero / one entries are multiplied out as the reference does
            float J[6];
            J[0] = -dot(v3(1.f, 0.f, 0.f), N);
            J[1] = -dot(v3(0.f, 1.f, 0.f), N);
            J[2] = -dot(v3(0.f, 0.f, 1.f), N);
            J[3] = -dot(v3(0.f, -_Pr.z, _Pr.y), N);
            J[4] = -dot(v3(_Pr.z, 0.f, -_Pr.x), N);
            J[5] = -dot(v3(-_Pr.y, _Pr.x, 0.f), N);
            // LSReweightTukey (reweighting.h:22-28)
            const float absr = fabsf(y);
            const float roc = y / c;
            const float omroc2 = 1.0f - roc * roc;
            const float tukey = (absr <= c) ? omroc2 * omroc2 : 0.0f;
            const float w = (1.0f / Pr.z) * tukey;
            const float yw = y * w;
#pragma unroll
            for (int r = 0; r < 6; ++r) sum.f[r] = J[r] * yw;           // mul_aTb(Jr, y*w)
            int i = 6;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int cc = 0; cc <= r; ++cc) sum.f[i++] = J[r] * J[cc] * w; // OuterProduct(Jr, w)
            sum.f[27] = y * y;
            sum.obs = 1u;
            dbg = make_float4(absr, absr, absr, 1.f);
        } else {
            dbg = make_float4(0.f, 0.f, 1.f, 1.f);
        }
    } else {
        dbg = make_float4(1.f, 0.f, 0.f, 1.f);
    }
}

__global__ __launch_bounds__(256) void k_icp_point_plane(const IcpParams p)
{
    extern __shared__ float lds[];
    const int n = blockDim.x * blockDim.y;
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const unsigned u = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned v = blockIdx.y * blockDim.y + threadIdx.y;

    Lss sum;
    lss_zero(sum);

    Pose KT_lr = p.KT_lr, T_rl = p.T_rl;
    if (p.dev_pose) { // uniform loads: the pose was written by k_lss_final_solve of the previous iteration
#pragma unroll
        for (int i = 0; i < 12; ++i) { KT_lr.m[i] = p.dev_pose[i]; T_rl.m[i] = p.dev_pose[12 + i]; }
    }
    float4 dbg;
    icp_pixel(sum, dbg, p.Pl, p.Pr, p.Nr, KT_lr, T_rl, p.c, u, v);
    if (p.dbg) reinterpret_cast<float4*>(p.dbg + (size_t)v * p.dbg_pitch)[u] = dbg;

    lss_tree(sum, tid, n, lds);
    if (tid == 0) lss_store(p.sums + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * LSS_WORDS, sum);
}

// sum of the per-block systems, fixed order; the result replaces sums[0..28].  mailbox (optional): 32 words of host memory mapped
// into the device's address space -- the system again, then the sequence word `seq` with system-scope release: the host thread
// that launched this spins on that word instead of enqueueing a copy and synchronising the stream (kfx_icp_point_plane).
__global__ __launch_bounds__(256) void k_lss_final(float* sums, const int nblocks, float* mailbox, const unsigned seq)
{
    __shared__ float lds[LSS_WORDS * 128];
    const int tid = threadIdx.x;
    Lss acc;
    lss_zero(acc);
    for (int b = tid; b < nblocks; b += 256) {
        const float* s = sums + (size_t)b * LSS_WORDS;
#pragma unroll
        for (int k = 0; k < 28; ++k) acc.f[k] += s[k];
        acc.obs += __float_as_uint(s[28]);
    }
    lss_tree(acc, tid, 256, lds); // its first __syncthreads orders every read above before the store below
    if (tid == 0) {
        lss_store(sums, acc);
        if (mailbox) {
            lss_store(mailbox, acc);
            __hip_atomic_store(reinterpret_cast<unsigned*>(mailbox) + 31, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Device-resident refinement loop.  The reference application reads every summed system back, solves it with
// Eigen on the host and uploads the new transforms (main.cpp:301-337): six launch + synchronise + read-back round
// trips per frame.  Here the 6x6 step runs on the GPU as well -- one thread, float64, the same algorithm as
// kangaroo_amd/tracking.py / apps/pose_solve.h (weak prior, rotation-only solve on the coarsest level, LU with
// complete pivoting and Eigen's rank threshold, closed-form SE(3) exponential) -- and leaves K*T_lp / T_lp^-1 in
// device memory for the next k_icp_point_plane.  The host enqueues all iterations back to back and synchronises once.
// State (doubles): T_lp row-major 3x4 [0..11], rmse [12], obs [13], tracking flag [14]; floats: KT_lr[12], T_rl[12].
// ---------------------------------------------------------------------------------------
struct RefineState {
    double T[12];
    double rmse, obs, good;
    float pose[24];
};

// The working set of the solve lives in LDS: the pivoting indexes it with run-time indices, which as per-thread arrays meant
// scratch memory -- 320 scratch instructions in the one-lane solve, each a dependent round trip to memory: 12.5 us per
// k_lss_final_solve, more than the k_icp_point_plane before it (profiles/r05_tracked).  Same arithmetic in the same order.
struct SolveLds {
    double lu[6][6];
    double b[6], c[6], y[6], x[6];
    int rows[6], cols[6];
};

template <int N>
__device__ void lu_solve_full_piv(SolveLds& w, const double* A, const double* b, double* x)
{
    double (&lu)[6][6] = w.lu;
    int (&rows)[6] = w.rows, (&cols)[6] = w.cols;
    for (int i = 0; i < N; ++i) {
        rows[i] = cols[i] = i;
        w.b[i] = b[i];
        for (int j = 0; j < N; ++j) lu[i][j] = A[i * N + j];
    }
    int nonzero = N;
    double maxpivot = 0.0;
    for (int k = 0; k < N; ++k) {
        int pr = k, pc = k;
        double biggest = 0.0;
        for (int i = k; i < N; ++i)
            for (int j = k; j < N; ++j)
                if (fabs(lu[i][j]) > biggest) { biggest = fabs(lu[i][j]); pr = i; pc = j; }
        if (biggest == 0.0) { nonzero = k; break; }
        if (biggest > maxpivot) maxpivot = biggest;
        if (pr != k) {
            for (int j = 0; j < N; ++j) { const double t = lu[k][j]; lu[k][j] = lu[pr][j]; lu[pr][j] = t; }
            const int t = rows[k]; rows[k] = rows[pr]; rows[pr] = t;
        }
        if (pc != k) {
            for (int i = 0; i < N; ++i) { const double t = lu[i][k]; lu[i][k] = lu[i][pc]; lu[i][pc] = t; }
            const int t = cols[k]; cols[k] = cols[pc]; cols[pc] = t;
        }
        const double pivot = lu[k][k];
        for (int i = k + 1; i < N; ++i) {
            const double f = lu[i][k] / pivot;
            lu[i][k] = f;
            for (int j = k + 1; j < N; ++j) lu[i][j] -= f * lu[k][j];
        }
    }
    const double thresh = 2.220446049250313e-16 * N * maxpivot;
    int rank = 0;
    for (int i = 0; i < nonzero; ++i) rank += fabs(lu[i][i]) > thresh ? 1 : 0;
    for (int i = 0; i < N; ++i) w.x[i] = 0.0;
    if (rank != 0) {
        for (int i = 0; i < N; ++i) {
            double ci = w.b[rows[i]];
            for (int j = 0; j < i; ++j) ci -= lu[i][j] * w.c[j];
            w.c[i] = ci;
            w.y[i] = 0.0;
        }
        for (int i = rank - 1; i >= 0; --i) {
            double sum = w.c[i];
            for (int j = i + 1; j < rank; ++j) sum -= lu[i][j] * w.y[j];
            w.y[i] = sum / lu[i][i];
        }
        for (int i = 0; i < N; ++i) w.x[cols[i]] = w.y[i];
    }
    for (int i = 0; i < N; ++i) x[i] = w.x[i];
}

// T <- T * exp(x); rotation_only uses exp(omega) with zero translation
__device__ void se3_right_multiply_exp(double T[12], const double x[6], bool rotation_only)
{
    const double* w = x + 3;
    const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = sqrt(t2);
    const double W[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
    double W2[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) W2[i][j] = W[i][0] * W[0][j] + W[i][1] * W[1][j] + W[i][2] * W[2][j];
    const bool small = t < 1e-10;
    double st, ct;
    sincos(t, &st, &ct);   // (one argument reduction for both: the values of sin(t) and cos(t))
    const double a = small ? 1.0 : st / t, b = small ? 0.5 : (1.0 - ct) / t2, c = small ? 1.0 / 6.0 : (t - st) / (t2 * t);
    double E[12];
    for (int i = 0; i < 3; ++i) {
        double p = 0.0;
        for (int j = 0; j < 3; ++j) {
            E[i * 4 + j] = (i == j ? 1.0 : 0.0) + a * W[i][j] + b * W2[i][j];
            p += ((i == j ? 1.0 : 0.0) + b * W[i][j] + c * W2[i][j]) * x[j];
        }
        E[i * 4 + 3] = rotation_only ? 0.0 : p;
    }
    double R[12];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) R[i * 4 + j] = T[i * 4 + 0] * E[0 * 4 + j] + T[i * 4 + 1] * E[1 * 4 + j] + T[i * 4 + 2] * E[2 * 4 + j];
        R[i * 4 + 3] = T[i * 4 + 0] * E[3] + T[i * 4 + 1] * E[7] + T[i * 4 + 2] * E[11] + T[i * 4 + 3];
    }
    for (int i = 0; i < 12; ++i) T[i] = R[i];
}

// KT_lr = K * T (3x4) and T_rl = T^-1 as floats (host and device: the first evaluation's come from the host, see kfx_icp_refine)
__host__ __device__ inline void pose_floats(float* pose, const float K[4], const double* T)
{
    for (int c = 0; c < 4; ++c) {
        pose[0 * 4 + c] = (float)((double)K[0] * T[0 * 4 + c] + (double)K[2] * T[2 * 4 + c]);
        pose[1 * 4 + c] = (float)((double)K[1] * T[1 * 4 + c] + (double)K[3] * T[2 * 4 + c]);
        pose[2 * 4 + c] = (float)T[2 * 4 + c];
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) pose[12 + i * 4 + j] = (float)T[j * 4 + i];
        pose[12 + i * 4 + 3] = (float)(-(T[0 * 4 + i] * T[3] + T[1 * 4 + i] * T[7] + T[2 * 4 + i] * T[11]));
    }
}
__device__ void publish_pose(RefineState* st, const float K[4], const double* T) { pose_floats(st->pose, K, T); }

__device__ void publish_pose(RefineState* st, const float K[4]) { publish_pose(st, K, st->T); }

struct K4 { float k[4]; };

__global__ void k_icp_refine_init(RefineState* st, const K4 K)
{
    for (int i = 0; i < 12; ++i) st->T[i] = (i % 5 == 0) ? 1.0 : 0.0;
    st->rmse = 0.0; st->obs = 0.0; st->good = 1.0;
    publish_pose(st, K.k);
}

// one Gauss-Newton step from the summed system in sums[0..28] (main.cpp:312-333); K_next: intrinsics of the level the
// NEXT evaluation runs on
__device__ void icp_solve_step(RefineState* st, const float* sums, const int rotation_only, const float max_rmse, const float K_next[4], SolveLds& w)
{
    double JTJ[36], JTy[6], x[6] = {0, 0, 0, 0, 0, 0};
    int i = 6;
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c <= r; ++c) {
            const double e = (double)sums[i++];
            JTJ[r * 6 + c] = e;
            JTJ[c * 6 + r] = e;
        }
    for (int r = 0; r < 6; ++r) { JTy[r] = (double)sums[r]; JTJ[r * 7] += 0.1 / 0.2; } // weak pose prior: depthSigma / motionSigma
    const float sq = sums[27];
    const unsigned obs = __float_as_uint(sums[28]);
    const float rmse = sqrtf(sq / (float)obs);
    st->rmse = (double)rmse;
    st->obs = (double)obs;
    st->good = rmse < max_rmse ? 1.0 : 0.0;
    if (rotation_only) {
        double A3[9], b3[3];
        for (int a = 0; a < 3; ++a) {
            b3[a] = JTy[3 + a];
            for (int b = 0; b < 3; ++b) A3[a * 3 + b] = JTJ[(3 + a) * 6 + 3 + b];
        }
        lu_solve_full_piv<3>(w, A3, b3, x + 3);
        for (int a = 3; a < 6; ++a) x[a] = -x[a];
        se3_right_multiply_exp(st->T, x, true);
    } else {
        lu_solve_full_piv<6>(w, JTJ, JTy, x);
        bool finite = true;
        for (int a = 0; a < 6; ++a) { x[a] = -x[a]; finite = finite && isfinite(x[a]); }
        if (finite) se3_right_multiply_exp(st->T, x, false);
    }
    publish_pose(st, K_next);
}


// ---------------------------------------------------------------------------------------
// The solve by one WAVE instead of one lane (round 5).  A lone lane runs ~3800 dependent instructions (13 us per launch, six
// launches per frame: more than the k_icp_point_plane launches they follow, profiles/r05_tracked); most of them are the
// full-pivot LU -- a 91-element pivot search, swaps and 55 multiply-subtracts, one after the other.  Here lane i * N + j holds
// lu[i][j]: the pivot is a six-step arg-max over the wave (largest magnitude, ties to the first element in row-major order:
// what the serial scan's strict `>` selects) through DPP and lane swaps, the two swaps and the elimination's operands are one round
// of lane gathers, an elimination step is one division and one multiply-subtract in every lane, the substitutions read rows
// and columns through v_readlane.  Every element sees
// the operations of lu_solve_full_piv in the same order with the same operands, so the solution is the serial one bit for bit
// (the persistent launch keeps the one-lane solve: test_gpu_persistent_icp_kernel_equals_the_chain_of_launches
// compares the two).  All 64 lanes of the wave must be active.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double shfl_f64(double v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ double readlane_f64(double v, int src)   // src uniform: two v_readlane_b32 instead of the LDS crossbar
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
// the pivot candidate of a lane and the better of two: larger magnitude, ties to the smaller lane (= the earlier element in
// row-major order, which is what the serial scan's strict `>` keeps)
struct Pivot { double v; int idx; };
__device__ __forceinline__ Pivot better(const Pivot a, const Pivot b)
{
    const bool take = b.v > a.v || (b.v == a.v && b.idx < a.idx);
    return take ? b : a;
}
template <int CTRL>
__device__ __forceinline__ Pivot pivot_dpp(const Pivot p)   // the candidate of the lane a DPP pattern pairs this one with
{
    Pivot o;
    o.v = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(p.v), CTRL, 0xF, 0xF, true),
                           __builtin_amdgcn_update_dpp(0, __double2loint(p.v), CTRL, 0xF, 0xF, true));
    o.idx = __builtin_amdgcn_update_dpp(0, p.idx, CTRL, 0xF, 0xF, true);
    return o;
}
// the best candidate of the wave in every lane, without the LDS crossbar: quad permutes, the mirrors of 8 and 16 lanes, then
// v_permlane16_swap / v_permlane32_swap (gfx950), which with both operands equal return the values of both partners
__device__ __forceinline__ Pivot pivot_wave_best(Pivot p)
{
    p = better(p, pivot_dpp<0xB1>(p));    // quad_perm [1,0,3,2]
    p = better(p, pivot_dpp<0x4E>(p));    // quad_perm [2,3,0,1]
    p = better(p, pivot_dpp<0x141>(p));   // row_half_mirror
    p = better(p, pivot_dpp<0x140>(p));   // row_mirror
    {
        const auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(p.v), (unsigned)__double2hiint(p.v), false, false);
        const auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(p.v), (unsigned)__double2loint(p.v), false, false);
        const auto x = __builtin_amdgcn_permlane16_swap((unsigned)p.idx, (unsigned)p.idx, false, false);
        p = better(Pivot{__hiloint2double((int)h[0], (int)l[0]), (int)x[0]}, Pivot{__hiloint2double((int)h[1], (int)l[1]), (int)x[1]});
    }
    {
        const auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(p.v), (unsigned)__double2hiint(p.v), false, false);
        const auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(p.v), (unsigned)__double2loint(p.v), false, false);
        const auto x = __builtin_amdgcn_permlane32_swap((unsigned)p.idx, (unsigned)p.idx, false, false);
        p = better(Pivot{__hiloint2double((int)h[0], (int)l[0]), (int)x[0]}, Pivot{__hiloint2double((int)h[1], (int)l[1]), (int)x[1]});
    }
    return p;
}

template <int N>
__device__ void lu_solve_wave(const float* sums, SolveLds& w)   // leaves x[0 .. N) in w.x
{
    constexpr int O = 6 - N;   // N = 3 (rotation only): the lower-right block of the system and its last three right-hand sides
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // (the lane within the wave, whatever the block's shape)
    const bool in = lane < N * N;
    const int i = in ? lane / N : 0, j = in ? lane % N : 0;
    double a = 0.0;
    {
        const int r = O + i, c = O + j, hi = r > c ? r : c, lo = r > c ? c : r;
        const double e = (double)sums[6 + hi * (hi + 1) / 2 + lo];
        a = i == j ? e + 0.1 / 0.2 : e;   // weak pose prior on the diagonal: depthSigma / motionSigma
        if (!in) a = 0.0;
    }
    int rl = lane, cl = lane;   // rows[lane], cols[lane] (lanes below N)
    int nonzero = N;
    double maxpivot = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const Pivot best = pivot_wave_best(Pivot{(in && i >= k && j >= k && fabs(a) > 0.0) ? fabs(a) : -1.0, lane});
        if (!(best.v > 0.0)) { nonzero = k; break; }   // (uniform: every lane holds the same candidate)
        if (best.v > maxpivot) maxpivot = best.v;
        const int idx = __builtin_amdgcn_readfirstlane(best.idx);
        const int pr = idx / N, pc = idx % N;
        // rows k and pr, then columns k and pc change places: element (i, j) comes from (si, sj), its column-k and row-k
        // partners of the elimination from (si, pc) and (pr, sj), the pivot from (pr, pc) -- one round of gathers
        const int si = i == k ? pr : (i == pr ? k : i), sj = j == k ? pc : (j == pc ? k : j);
        const double pivot = readlane_f64(a, idx);
        const double aij = shfl_f64(a, in ? si * N + sj : lane), aik = shfl_f64(a, in ? si * N + pc : lane), akj = shfl_f64(a, in ? pr * N + sj : lane);
        rl = __shfl(rl, lane == k ? pr : (lane == pr ? k : lane), 64);
        cl = __shfl(cl, lane == k ? pc : (lane == pc ? k : lane), 64);
        const double f = aik / pivot;
        a = aij;
        if (in && i > k) {
            if (j == k) a = f;
            else if (j > k) a = aij - f * akj;
        }
    }
    const double thresh = 2.220446049250313e-16 * N * maxpivot;
    const int rank = __popcll(__ballot(in && i == j && i < nonzero && fabs(a) > thresh));
    if (lane < N) w.x[lane] = 0.0;
    if (rank == 0) return;   // (uniform)
    // c[i] = b[rows[i]] - sum_{j < i} lu[i][j] c[j], j ascending
    const double bl = lane < N ? (double)sums[O + lane] : 0.0;
    double c = shfl_f64(bl, lane < N ? rl : lane);
#pragma unroll
    for (int jj = 0; jj < N - 1; ++jj) {
        const double cj = readlane_f64(c, jj), lij = shfl_f64(a, lane < N ? lane * N + jj : lane);
        if (lane < N && lane > jj) c = c - lij * cj;
    }
    // y[i] = (c[i] - sum_{i < j < rank} lu[i][j] y[j]) / lu[i][i], i descending, j ascending (every lane follows the same chain)
    double y = 0.0;
    for (int ii = rank - 1; ii >= 0; --ii) {
        double sum = readlane_f64(c, ii);
        for (int jj = ii + 1; jj < rank; ++jj) sum = sum - readlane_f64(a, ii * N + jj) * readlane_f64(y, jj);
        const double yi = sum / readlane_f64(a, ii * N + ii);
        if (lane == ii) y = yi;
    }
    if (lane < N) w.x[cl] = y;   // x[cols[i]] = y[i]: cols is a permutation
}

// icp_solve_step with the LU spread over the calling wave (all 64 lanes active); lane 0 does what remains
// T_lane: st->T[lane] in lanes 0 .. 11, requested by the caller before the block sums were added up (one memory round trip
// less on the one-lane tail)
// always_store: write T even when the step is refused (the first evaluation of a call: st->T still holds the previous call's pose)
// mailbox (optional): 16 doubles of host memory mapped into the device's address space -- the call's result {T, rmse, obs, good} goes
// there too, then the sequence word (system-scope release) the host spins on (kfx_icp_refine: no copy command for the read-back)
__device__ void icp_solve_step_wave(RefineState* st, const float* sums, const int rotation_only, const float max_rmse, const float K_next[4], SolveLds& w,
                                    const double T_lane, const bool always_store = false, double* mailbox = nullptr, const unsigned seq = 0u)
{
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // (the lane within the wave, whatever the block's shape)
    if (rotation_only) lu_solve_wave<3>(sums, w);
    else lu_solve_wave<6>(sums, w);
    __builtin_amdgcn_wave_barrier();   // (one wave: its LDS operations complete in order)
    double T[12];
#pragma unroll
    for (int m = 0; m < 12; ++m) T[m] = readlane_f64(T_lane, m);
    if (lane != 0) return;
    const float sq = sums[27];
    const unsigned obs = __float_as_uint(sums[28]);
    const float rmse = sqrtf(sq / (float)obs);
    st->rmse = (double)rmse;
    st->obs = (double)obs;
    st->good = rmse < max_rmse ? 1.0 : 0.0;
    double x[6] = {0, 0, 0, 0, 0, 0};
    bool moved = true;
    if (rotation_only) {
        for (int a = 0; a < 3; ++a) x[3 + a] = -w.x[a];
        se3_right_multiply_exp(T, x, true);
    } else {
        for (int a = 0; a < 6; ++a) { x[a] = -w.x[a]; moved = moved && isfinite(x[a]); }
        if (moved) se3_right_multiply_exp(T, x, false);
    }
    if (moved || always_store)
        for (int m = 0; m < 12; ++m) st->T[m] = T[m];
    publish_pose(st, K_next, T);
    if (mailbox) {   // (T is what st->T holds now: the step, or the pose before it when the step was refused)
        for (int m = 0; m < 12; ++m) mailbox[m] = T[m];
        mailbox[12] = (double)rmse;
        mailbox[13] = (double)obs;
        mailbox[14] = rmse < max_rmse ? 1.0 : 0.0;
        __hip_atomic_store(reinterpret_cast<unsigned*>(mailbox + 15), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// k_lss_final followed by the 6x6 step as one launch (the device-resident loop is a chain of ~5 us kernels: every launch saved is
// time saved).  Thread 0 holds the summed system after the tree; it stores it (so sums[0..28] is what k_lss_final leaves) and
// the first wave solves it.
// first: the first evaluation of a kfx_icp_refine call -- the pose before the step is the identity (no initialising launch)
__global__ __launch_bounds__(256) void k_lss_final_solve(float* sums, const int nblocks, RefineState* st, const int rotation_only,
                                                         const float max_rmse, const K4 K_next, const int first, double* mailbox, const unsigned seq)
{
    __shared__ float lds[LSS_WORDS * 128];
    __shared__ float s_sum[LSS_WORDS];
    __shared__ SolveLds s_solve;
    const int tid = threadIdx.x;
    const double T_lane = tid < 12 ? (first ? ((tid % 5 == 0) ? 1.0 : 0.0) : st->T[tid]) : 0.0;   // (for the tail of the solve: on its way while the sums are added up)
    Lss acc;
    lss_zero(acc);
    for (int b = tid; b < nblocks; b += 256) {
        const float* s = sums + (size_t)b * LSS_WORDS;
#pragma unroll
        for (int k = 0; k < 28; ++k) acc.f[k] += s[k];
        acc.obs += __float_as_uint(s[28]);
    }
    lss_tree(acc, tid, 256, lds);
    if (tid == 0) {
        lss_store(sums, acc);
        lss_store(s_sum, acc);
    }
    __syncthreads();
    if (tid < 64) icp_solve_step_wave(st, s_sum, rotation_only, max_rmse, K_next.k, s_solve, T_lane, first != 0, mailbox, seq);
}

// ---------------------------------------------------------------------------------------
// The whole refinement -- every level, every iteration -- as ONE launch (round-4 verdict, item 6: the chain above was 13 launches
// of 5-9 us of work each, and a frame's budget is 0.4 ms).  A fixed number of workgroups stays resident and walks the schedule:
//   A. the workgroups share out the level's pixel blocks (the reference's launch geometry: gcd(w, 16) x gcd(h, 16) pixels,
//      launch_utils.h:61-65) and leave each block's system in `sums` -- the same per-pixel function, the same tree
//      (lss_tree), so the same bits as k_icp_point_plane;
//   B. one grid-wide barrier (an arrival counter and a generation word in device memory: release fence + atomic add; the last
//      one to arrive advances the generation, the others poll it with s_sleep between two looks, then an acquire fence);
//   C. EVERY workgroup adds up the blocks' systems in k_lss_final's fixed order and takes the 6 x 6 step by itself (thread 0,
//      float64: deterministic, so all workgroups hold the same pose without a second barrier or a broadcast); workgroup 0 leaves
//      the result in the RefineState.
// `sums` is double-buffered by iteration parity: a workgroup may already be writing the next iteration's block systems while a
// slower one still adds up this iteration's.  Every poll is bounded: if a workgroup never arrives (the grid was not resident
// after all), the others set the abort word, leave, and the host reports KFX_E_RANGE instead of waiting for a watchdog.
// The grid is at most the device's resident capacity for this kernel (hipOccupancyMaxActiveBlocksPerMultiprocessor).
// ---------------------------------------------------------------------------------------
struct IcpLevelDev {
    ImgView Pl, Pr, Nr;
    float K[4];
    int iterations, rotation_only;
    int bx, by, gx, gy;   // block shape (pixels), blocks along x / y
};
struct IcpPersistent {
    IcpLevelDev lv[4];
    int n_levels;
    float c, max_rmse;
    unsigned char* dbg;
    size_t dbg_pitch;
    int dbg_w, dbg_h;
    float* sums[2];        // block systems, by iteration parity
    RefineState* st;       // result (workgroup 0)
    unsigned* bar;         // [0] arrivals, [1] generation, [2] abort
};

// The block systems cross workgroups (and XCDs) through device-scope atomic stores and loads (lss_store_agent), so that the barrier
// needs no cache-wide release / acquire (an agent-scope fence writes back and invalidates a whole L2: measured, it made the
// persistent launch slower than the chain of launches).
__device__ __forceinline__ bool grid_barrier(unsigned* bar, const unsigned n_groups, unsigned& gen)
{
    __shared__ int s_ok;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        int ok = 1;
        __builtin_amdgcn_s_waitcnt(0);   // this workgroup's block systems (thread 0 stored them) have reached the coherence point
        const unsigned arrived = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == n_groups - 1) {
            __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);
            __hip_atomic_store(&bar[1], gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                if (__hip_atomic_load(&bar[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || ++spins > (1u << 22)) {   // ~ a second
                    __hip_atomic_store(&bar[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        s_ok = ok;
    }
    __syncthreads();
    gen += 1;
    return s_ok != 0;
}

__global__ __launch_bounds__(256) void k_icp_refine_persistent(const IcpPersistent q)
{
    __shared__ float lds[LSS_WORDS * 128];
    __shared__ float s_sum[LSS_WORDS];
    __shared__ SolveLds s_solve;
    __shared__ RefineState s_st;
    const int tid = threadIdx.x;
    const unsigned G = gridDim.x;
    unsigned gen = 0;   // (the host zeroes the barrier words before the launch)
    int first = -1;
    for (int l = 0; l < q.n_levels; ++l)
        if (q.lv[l].iterations > 0 && q.lv[l].gx > 0 && q.lv[l].gy > 0) { first = l; break; }
    if (tid == 0) {
        for (int i = 0; i < 12; ++i) s_st.T[i] = (i % 5 == 0) ? 1.0 : 0.0;
        s_st.rmse = 0.0; s_st.obs = 0.0; s_st.good = 1.0;
        float k0[4] = {0.f, 0.f, 0.f, 0.f};
        if (first >= 0) for (int i = 0; i < 4; ++i) k0[i] = q.lv[first].K[i];
        publish_pose(&s_st, k0);
    }
    __syncthreads();
    int parity = 0;
    bool alive = true;
    for (int l = 0; l < q.n_levels && alive; ++l) {
        const IcpLevelDev& L = q.lv[l];
        if (L.iterations <= 0 || L.gx <= 0 || L.gy <= 0) continue;
        const int n = L.bx * L.by, nblocks = L.gx * L.gy;
        const int tx = tid % L.bx, ty = tid / L.bx;
        int nxt = -1;
        for (int m = l + 1; m < q.n_levels; ++m)
            if (q.lv[m].iterations > 0 && q.lv[m].gx > 0 && q.lv[m].gy > 0) { nxt = m; break; }
        const bool dbg_on = q.dbg && q.dbg_w >= L.Pl.w && q.dbg_h >= L.Pl.h;
        for (int it = 0; it < L.iterations && alive; ++it, parity ^= 1) {
            float* sums = q.sums[parity];
            Pose KT_lr, T_rl;
#pragma unroll
            for (int i = 0; i < 12; ++i) { KT_lr.m[i] = s_st.pose[i]; T_rl.m[i] = s_st.pose[12 + i]; }
            // ---- A: this workgroup's share of the pixel blocks ----
            for (int b = (int)blockIdx.x; b < nblocks; b += (int)G) {
                const unsigned u = (unsigned)((b % L.gx) * L.bx + tx), v = (unsigned)((b / L.gx) * L.by + ty);
                Lss sum;
                lss_zero(sum);
                if (tid < n) {
                    float4 dbg;
                    icp_pixel(sum, dbg, L.Pl, L.Pr, L.Nr, KT_lr, T_rl, q.c, u, v);
                    if (dbg_on) reinterpret_cast<float4*>(q.dbg + (size_t)v * q.dbg_pitch)[u] = dbg;
                }
                lss_tree(sum, tid, n, lds);
                if (tid == 0) lss_store_agent(sums + (size_t)b * LSS_WORDS, sum);
                __syncthreads();   // (lds is reused by the next block's tree)
            }
            // ---- B ----
            alive = grid_barrier(q.bar, G, gen);
            if (!alive) break;
            // ---- C: the sum over the blocks in k_lss_final's order, and the step ----
            Lss acc;
            lss_zero(acc);
            for (int b = tid; b < nblocks; b += 256) {
                const float* sp = sums + (size_t)b * LSS_WORDS;
#pragma unroll
                for (int k = 0; k < 28; ++k) acc.f[k] += __hip_atomic_load(sp + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (written by other workgroups during this launch)
                acc.obs += __float_as_uint(__hip_atomic_load(sp + 28, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            lss_tree(acc, tid, 256, lds);
            if (tid == 0) {
                lss_store(s_sum, acc);
                const float* Kn = (it + 1 < L.iterations || nxt < 0) ? L.K : q.lv[nxt].K;
                float kn[4] = {Kn[0], Kn[1], Kn[2], Kn[3]};
                icp_solve_step(&s_st, s_sum, L.rotation_only, q.max_rmse, kn, s_solve);
            }
            __syncthreads();
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        if (!alive) s_st.good = -1.0;   // the grid never met: reported by the host
        *q.st = s_st;
    }
}

static unsigned gcd_u(unsigned a, unsigned b) { return b == 0 ? a : gcd_u(b, a % b); }

} // namespace kfx

using namespace kfx;

// LeastSquaresSystem<float,6> PoseRefinementProjectiveIcpPointPlane(dPl, dPr, dNr, KT_lr, T_rl, c, dWorkspace,
// dDebug) (cu_model_refinement.cu:595-608).  Launch geometry as InitDimFromOutputImage(dPl, 16, 16)
// (launch_utils.h:61-65): block = (gcd(w,16), gcd(h,16)), grid = (w / bx, h / by).  Blocks until the 116-byte
// result is on the host (the reference's thrust::reduce blocks too).
extern "C" int kfx_icp_point_plane(const kfx_image* Pl, const kfx_image* Pr, const kfx_image* Nr, const float KT_lr[12],
                                   const float T_rl[12], float c, const kfx_image* workspace, const kfx_image* debug,
                                   kfx_lss6* out, kfx_stream stream)
{
    static_assert(sizeof(kfx_lss6) == LSS_WORDS * 4, "LeastSquaresSystem<float,6> layout");
    if (!Pl || !Pr || !Nr || !KT_lr || !T_rl || !workspace || !out || !Pl->ptr || !Pr->ptr || !Nr->ptr || !workspace->ptr)
        return set_error(KFX_E_NULL, "PoseRefinementProjectiveIcpPointPlane: null argument");
    memset(out, 0, sizeof(*out));
    if (Pl->w == 0 || Pl->h == 0) return 0;
    if (Pr->w < Pl->w || Pr->h < Pl->h || Nr->w < Pl->w || Nr->h < Pl->h || (debug && debug->ptr && (debug->w < Pl->w || debug->h < Pl->h)))
        return set_error(KFX_E_SHAPE, "PoseRefinementProjectiveIcpPointPlane: dPr / dNr / dDebug smaller than dPl");
    if ((((uintptr_t)Pl->ptr | Pl->pitch | (uintptr_t)Pr->ptr | Pr->pitch | (uintptr_t)Nr->ptr | Nr->pitch) & 15) || ((uintptr_t)workspace->ptr & 3) ||
        (debug && debug->ptr && (((uintptr_t)debug->ptr | debug->pitch) & 15)))
        return set_error(KFX_E_ALIGN, "PoseRefinementProjectiveIcpPointPlane: float4 images must be 16-byte aligned");
    const unsigned bx = gcd_u((unsigned)Pl->w, 16), by = gcd_u((unsigned)Pl->h, 16);
    const dim3 block(bx, by), grid((unsigned)(Pl->w / bx), (unsigned)(Pl->h / by));
    const size_t nblocks = (size_t)grid.x * grid.y;
    // Image::PackedImage asserts the workspace holds gridDim.x * gridDim.y systems (Image.h:464-468)
    if (nblocks * sizeof(kfx_lss6) > workspace->pitch * workspace->h)
        return set_error(KFX_E_SHAPE, "PoseRefinementProjectiveIcpPointPlane: workspace too small");
    if (nblocks > 0x7fffffff) return set_error(KFX_E_RANGE, "PoseRefinementProjectiveIcpPointPlane: image too large");
    IcpParams p;
    p.Pl = ImgView{(const unsigned char*)Pl->ptr, Pl->pitch, (int)Pl->w, (int)Pl->h};
    p.Pr = ImgView{(const unsigned char*)Pr->ptr, Pr->pitch, (int)Pr->w, (int)Pr->h};
    p.Nr = ImgView{(const unsigned char*)Nr->ptr, Nr->pitch, (int)Nr->w, (int)Nr->h};
    p.dbg = (debug && debug->ptr) ? (unsigned char*)debug->ptr : nullptr;
    p.dbg_pitch = p.dbg ? debug->pitch : 0;
    for (int i = 0; i < 12; ++i) { p.KT_lr.m[i] = KT_lr[i]; p.T_rl.m[i] = T_rl[i]; }
    p.c = c;
    p.dev_pose = nullptr;
    p.sums = (float*)workspace->ptr;
    const int n = (int)(bx * by);
    const size_t lds_bytes = n >= 128 ? (size_t)LSS_WORDS * (n / 2) * sizeof(float) : 0;
    // The result's way to the host (the reference's thrust::reduce blocks too; the application solves on the host between two of
    // these calls, six times per frame, main.cpp:301-342).  A mailbox per calling thread: 128 bytes of page-locked host memory
    // mapped into the device's address space; k_lss_final writes the 29 words and then a sequence word there (system-scope release)
    // and this thread spins on the word -- no copy command, no stream synchronisation (profiles/r06_tracked: the copy's own
    // 4.3 us and the wake-up after hipStreamSynchronize were 8 % of the drop-in application's frame).  KFX_ICP_MAILBOX=0, or no
    // mapped memory: the round-5 path (a copy into a pinned staging word + hipStreamSynchronize).
    struct Mailbox { float* host; float* dev; unsigned seq; };
    thread_local Mailbox mb = {nullptr, nullptr, 0u};
    static const bool mailbox_env = [] { const char* e = getenv("KFX_ICP_MAILBOX"); return !e || atoi(e) != 0; }();
    if (mailbox_env && !mb.host) {
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, 128, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
            memset(h, 0, 128);
            mb.host = (float*)h; mb.dev = (float*)d;
        } else {
            (void)hipGetLastError();
            if (h) (void)hipHostFree(h);
            mb.host = nullptr;
        }
    }
    const bool use_mailbox = mailbox_env && mb.host != nullptr;
    const unsigned seq = use_mailbox ? (++mb.seq ? mb.seq : ++mb.seq) : 0u;   // (never 0: the word's initial value)
    hipLaunchKernelGGL(k_icp_point_plane, grid, block, lds_bytes, (hipStream_t)stream, p);
    hipLaunchKernelGGL(k_lss_final, dim3(1), dim3(256), 0, (hipStream_t)stream, p.sums, (int)nblocks, use_mailbox ? mb.dev : nullptr, seq);
    int st = check_launch("kfx_icp_point_plane");
    if (st) return st;
    if (use_mailbox) {
        volatile unsigned* word = reinterpret_cast<volatile unsigned*>(mb.host) + 31;
        // bounded: every 4096 looks the stream is asked whether it is still running -- a launch that failed on the device, or a
        // word that never arrives, ends in the synchronising path below instead of a hang
        bool arrived = false;
        for (unsigned spins = 0; !arrived; ++spins) {
            if (__atomic_load_n(const_cast<unsigned*>(word), __ATOMIC_ACQUIRE) == seq) { arrived = true; break; }
            __builtin_ia32_pause();
            if ((spins & 4095u) == 4095u) {
                const hipError_t q = hipStreamQuery((hipStream_t)stream);
                if (q != hipErrorNotReady) {   // finished (or failed): one last look, then the copy path decides
                    (void)hipGetLastError();
                    arrived = __atomic_load_n(const_cast<unsigned*>(word), __ATOMIC_ACQUIRE) == seq;
                    break;
                }
            }
        }
        if (arrived) {
            memcpy(out, mb.host, sizeof(*out));
            return 0;
        }
    }
    // read-back through a pinned staging word per calling thread: a device -> pageable copy is staged by the runtime
    // and costs a second synchronisation
    thread_local kfx_lss6* stage = nullptr;
    if (!stage && hipHostMalloc((void**)&stage, sizeof(kfx_lss6), hipHostMallocDefault) != hipSuccess) stage = nullptr;
    kfx_lss6* dst = stage ? stage : out;
    hipError_t e = hipMemcpyAsync(dst, p.sums, sizeof(*out), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return set_error((int)e, hipGetErrorString(e));
    if (dst != out) memcpy(out, dst, sizeof(*out));
    return 0;
}

// The coarse-to-fine loop of main.cpp:301-337 enqueued as one chain of kernels: for every level (coarsest first) and
// iteration, k_icp_point_plane -> k_lss_final_solve (block sum + 6x6 step), all on `stream`, one synchronisation at the end.
// levels[l]: the three vertex / normal maps, the level's intrinsics and its iteration count, given COARSEST FIRST;
// the first level with more than one level in total is solved for rotation only, as the application does.
// workspace: >= max over levels of (blocks * 116) + 512 bytes; result: T_lp (row-major 3x4, float64), rmse, obs,
// tracking_good (rmse < max_rmse at the last evaluation).
static int icp_refine_impl(const kfx_icp_level* levels, int n_levels, float c, float max_rmse, const kfx_image* workspace,
                           const kfx_image* debug, double T_lp[12], float* rmse, unsigned* obs, int* tracking_good, void (*enqueue_more)(void*),
                           void* user, kfx_stream stream)
{
    if (!levels || n_levels <= 0 || !workspace || !workspace->ptr || !T_lp) return set_error(KFX_E_NULL, "kfx_icp_refine: null argument");
    size_t max_blocks = 0;
    for (int l = 0; l < n_levels; ++l) {
        const kfx_icp_level& L = levels[l];
        if (!L.Pl.ptr || !L.Pr.ptr || !L.Nr.ptr) return set_error(KFX_E_NULL, "kfx_icp_refine: null image");
        if (L.Pr.w < L.Pl.w || L.Pr.h < L.Pl.h || L.Nr.w < L.Pl.w || L.Nr.h < L.Pl.h) return set_error(KFX_E_SHAPE, "kfx_icp_refine: dPr / dNr smaller than dPl");
        if (((uintptr_t)L.Pl.ptr | L.Pl.pitch | (uintptr_t)L.Pr.ptr | L.Pr.pitch | (uintptr_t)L.Nr.ptr | L.Nr.pitch) & 15)
            return set_error(KFX_E_ALIGN, "kfx_icp_refine: float4 images must be 16-byte aligned");
        if (L.Pl.w == 0 || L.Pl.h == 0) continue;
        const unsigned bx = gcd_u((unsigned)L.Pl.w, 16), by = gcd_u((unsigned)L.Pl.h, 16);
        const size_t nb = (L.Pl.w / bx) * (L.Pl.h / by);
        if (nb > max_blocks) max_blocks = nb;
    }
    const size_t state_off = (max_blocks * sizeof(kfx_lss6) + 255) / 256 * 256;
    if (state_off + sizeof(RefineState) > workspace->pitch * workspace->h || ((uintptr_t)workspace->ptr & 7))
        return set_error(KFX_E_SHAPE, "kfx_icp_refine: workspace too small or misaligned");
    hipStream_t s = (hipStream_t)stream;
    float* sums = (float*)workspace->ptr;
    RefineState* st = (RefineState*)((unsigned char*)workspace->ptr + state_off);
    // KFX_ICP_PERSISTENT=1: one persistent launch (k_icp_refine_persistent) where the workspace has room for the second set of block
    // systems and the barrier words, the levels fit its table and the device can hold the grid.  Same bits as the chain of launches
    // below, which stays the default: measured on MI355X (512^3 tracked frame, C++ loop) the chain takes 0.655 ms per frame, the
    // persistent launch 0.754 (128 workgroups) - 0.844 ms (512) -- a grid-wide barrier across the eight XCDs and a 6 x 6 float64
    // step in every workgroup cost more than the launch boundaries they replace (EXPERIMENTS.md 7.3).
    static const int persistent_env = [] { const char* e = getenv("KFX_ICP_PERSISTENT"); return e ? atoi(e) : 0; }();
    const size_t sums2_off = (state_off + sizeof(RefineState) + 255) / 256 * 256, bar_off = sums2_off + state_off;
    bool fits = persistent_env != 0 && n_levels <= 4 && bar_off + 256 <= workspace->pitch * workspace->h;
    for (int l = 0; l < n_levels && fits; ++l) {
        const kfx_icp_level& L = levels[l];
        if (L.iterations <= 0 || L.Pl.w == 0 || L.Pl.h == 0) continue;
        if (gcd_u((unsigned)L.Pl.w, 16) * gcd_u((unsigned)L.Pl.h, 16) > 256u || L.Pl.w > 0x7fffffffull || L.Pl.h > 0x7fffffffull) fits = false;
    }
    if (fits) {
        static int capacity = -1;   // workgroups of this kernel the device holds at once
        if (capacity < 0) {
            int dev = 0, cus = 0, per_cu = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_icp_refine_persistent, 256, 0) == hipSuccess)
                capacity = cus * (per_cu < 2 ? per_cu : 2);   // two per CU are plenty (a level has at most a few blocks per workgroup); well inside what fits
            else
                capacity = 0;
            (void)hipGetLastError();
        }
        if (capacity >= 1) {
            IcpPersistent q;
            memset(&q, 0, sizeof(q));
            size_t most = 1;
            for (int l = 0; l < n_levels; ++l) {
                const kfx_icp_level& L = levels[l];
                IcpLevelDev& d = q.lv[l];
                d.iterations = (L.Pl.w && L.Pl.h) ? L.iterations : 0;
                d.rotation_only = L.rotation_only ? 1 : 0;
                for (int i = 0; i < 4; ++i) d.K[i] = L.K[i];
                if (d.iterations <= 0) continue;
                d.Pl = ImgView{(const unsigned char*)L.Pl.ptr, L.Pl.pitch, (int)L.Pl.w, (int)L.Pl.h};
                d.Pr = ImgView{(const unsigned char*)L.Pr.ptr, L.Pr.pitch, (int)L.Pr.w, (int)L.Pr.h};
                d.Nr = ImgView{(const unsigned char*)L.Nr.ptr, L.Nr.pitch, (int)L.Nr.w, (int)L.Nr.h};
                d.bx = (int)gcd_u((unsigned)L.Pl.w, 16); d.by = (int)gcd_u((unsigned)L.Pl.h, 16);
                d.gx = (int)(L.Pl.w / d.bx); d.gy = (int)(L.Pl.h / d.by);
                if ((size_t)d.gx * d.gy > most) most = (size_t)d.gx * d.gy;
            }
            q.n_levels = n_levels;
            q.c = c; q.max_rmse = max_rmse;
            const bool dbg = debug && debug->ptr && !(((uintptr_t)debug->ptr | debug->pitch) & 15);
            q.dbg = dbg ? (unsigned char*)debug->ptr : nullptr;
            q.dbg_pitch = dbg ? debug->pitch : 0;
            q.dbg_w = dbg ? (int)debug->w : 0; q.dbg_h = dbg ? (int)debug->h : 0;
            q.sums[0] = sums;
            q.sums[1] = (float*)((unsigned char*)workspace->ptr + sums2_off);
            q.st = st;
            q.bar = (unsigned*)((unsigned char*)workspace->ptr + bar_off);
            static const int grid_env = [] { const char* e = getenv("KFX_ICP_GRID"); return e ? atoi(e) : 0; }();   // (A/B: cap on the resident workgroups)
            size_t want = (size_t)capacity < most ? (size_t)capacity : most;
            if (grid_env > 0 && (size_t)grid_env < want) want = (size_t)grid_env;
            const unsigned grid = (unsigned)want;
            hipError_t he = hipMemsetAsync(q.bar, 0, 64, s);
            if (he != hipSuccess) { (void)hipGetLastError(); return set_error((int)he, "kfx_icp_refine: hipMemsetAsync"); }
            hipLaunchKernelGGL(k_icp_refine_persistent, dim3(grid), dim3(256), 0, s, q);
            if (int e0 = check_launch("kfx_icp_refine")) return e0;
            thread_local double* stage_p = nullptr;
            if (!stage_p && hipHostMalloc((void**)&stage_p, 15 * sizeof(double), hipHostMallocDefault) != hipSuccess) return set_error(KFX_E_RANGE, "kfx_icp_refine: pinned staging");
            hipError_t e = hipMemcpyAsync(stage_p, st, 15 * sizeof(double), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess && enqueue_more) enqueue_more(user);   // (the hook's contract: called once per successful enqueue)
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) return set_error((int)e, hipGetErrorString(e));
            if (stage_p[14] < 0.0) return set_error(KFX_E_RANGE, "kfx_icp_refine: the persistent grid never met at its barrier (not resident)");
            for (int i = 0; i < 12; ++i) T_lp[i] = stage_p[i];
            if (rmse) *rmse = (float)stage_p[12];
            if (obs) *obs = (unsigned)stage_p[13];
            if (tracking_good) *tracking_good = stage_p[14] != 0.0 ? 1 : 0;
            return 0;
        }
    }
    // flatten the schedule so that every solve knows the intrinsics of the evaluation that follows it
    int first = -1;
    for (int l = 0; l < n_levels; ++l)
        if (levels[l].iterations > 0 && levels[l].Pl.w && levels[l].Pl.h) { first = l; break; }
    K4 k0;
    for (int i = 0; i < 4; ++i) k0.k[i] = first >= 0 ? levels[first].K[i] : 0.f;
    // The pose starts at the identity: the first evaluation takes K * I and I^-1 as launch arguments (the floats pose_floats gives:
    // what the initialising launch used to leave in device memory) and its solve starts from the identity itself -- one launch
    // less per call.  Only a call without any evaluation initialises the state by a launch (it returns the identity).
    float pose0[24];
    {
        const double I12[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        pose_floats(pose0, k0.k, I12);
    }
    if (first < 0) hipLaunchKernelGGL(k_icp_refine_init, dim3(1), dim3(1), 0, s, st, k0);
    // The result's way to the host: the LAST solve writes {T, rmse, obs, good} and a sequence word into a mailbox of page-locked host
    // memory mapped into the device's address space (one per calling thread) and this thread spins on the word -- no copy command
    // (4 us of the stream per frame), no wait for an event (as kfx_icp_point_plane's mailbox; KFX_ICP_MAILBOX=0: the copy below)
    struct Mailbox { double* host; double* dev; unsigned seq; };
    thread_local Mailbox mb = {nullptr, nullptr, 0u};
    static const bool mailbox_env = [] { const char* e = getenv("KFX_ICP_MAILBOX"); return !e || atoi(e) != 0; }();
    if (mailbox_env && !mb.host) {
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, 128, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
            memset(h, 0, 128);
            mb.host = (double*)h; mb.dev = (double*)d;
        } else {
            (void)hipGetLastError();
            if (h) (void)hipHostFree(h);
            mb.host = nullptr;
        }
    }
    int evaluations = 0;
    for (int l = 0; l < n_levels; ++l)
        if (levels[l].iterations > 0 && levels[l].Pl.w && levels[l].Pl.h) evaluations += levels[l].iterations;
    const bool use_mailbox = mailbox_env && mb.host != nullptr && evaluations > 0;
    const unsigned seq = use_mailbox ? (++mb.seq ? mb.seq : ++mb.seq) : 0u;   // (never 0: the word's initial value)
    int evaluation = 0;
    bool first_eval = true;
    for (int l = 0; l < n_levels; ++l) {
        const kfx_icp_level& L = levels[l];
        if (L.iterations <= 0 || L.Pl.w == 0 || L.Pl.h == 0) continue;
        const unsigned bx = gcd_u((unsigned)L.Pl.w, 16), by = gcd_u((unsigned)L.Pl.h, 16);
        const dim3 block(bx, by), grid((unsigned)(L.Pl.w / bx), (unsigned)(L.Pl.h / by));
        const int nblocks = (int)(grid.x * grid.y), n = (int)(bx * by);
        const size_t lds_bytes = n >= 128 ? (size_t)LSS_WORDS * (n / 2) * sizeof(float) : 0;
        IcpParams p;
        p.Pl = ImgView{(const unsigned char*)L.Pl.ptr, L.Pl.pitch, (int)L.Pl.w, (int)L.Pl.h};
        p.Pr = ImgView{(const unsigned char*)L.Pr.ptr, L.Pr.pitch, (int)L.Pr.w, (int)L.Pr.h};
        p.Nr = ImgView{(const unsigned char*)L.Nr.ptr, L.Nr.pitch, (int)L.Nr.w, (int)L.Nr.h};
        const bool dbg = debug && debug->ptr && debug->w >= L.Pl.w && debug->h >= L.Pl.h && !(((uintptr_t)debug->ptr | debug->pitch) & 15);
        p.dbg = dbg ? (unsigned char*)debug->ptr : nullptr;
        p.dbg_pitch = dbg ? debug->pitch : 0;
        p.c = c;
        p.dev_pose = st->pose;
        p.sums = sums;
        // next evaluation's intrinsics: this level while iterations remain, else the next level that has any
        int nxt = -1;
        for (int m = l + 1; m < n_levels; ++m)
            if (levels[m].iterations > 0 && levels[m].Pl.w && levels[m].Pl.h) { nxt = m; break; }
        for (int it = 0; it < L.iterations; ++it) {
            K4 kn;
            const float* Kn = (it + 1 < L.iterations || nxt < 0) ? L.K : levels[nxt].K;
            for (int i = 0; i < 4; ++i) kn.k[i] = Kn[i];
            if (first_eval) {
                IcpParams p0 = p;
                p0.dev_pose = nullptr;
                for (int i = 0; i < 12; ++i) { p0.KT_lr.m[i] = pose0[i]; p0.T_rl.m[i] = pose0[12 + i]; }
                hipLaunchKernelGGL(k_icp_point_plane, grid, block, lds_bytes, s, p0);
            } else {
                hipLaunchKernelGGL(k_icp_point_plane, grid, block, lds_bytes, s, p);
            }
            const bool last = ++evaluation == evaluations;
            hipLaunchKernelGGL(k_lss_final_solve, dim3(1), dim3(256), 0, s, sums, nblocks, st, L.rotation_only ? 1 : 0, max_rmse, kn, first_eval ? 1 : 0,
                               (use_mailbox && last) ? mb.dev : nullptr, seq);
            first_eval = false;
        }
    }
    int e0 = check_launch("kfx_icp_refine");
    if (e0) return e0;
    const auto deliver = [&](const double* r) {
        for (int i = 0; i < 12; ++i) T_lp[i] = r[i];
        if (rmse) *rmse = (float)r[12];
        if (obs) *obs = (unsigned)r[13];
        if (tracking_good) *tracking_good = r[14] != 0.0 ? 1 : 0;
    };
    if (use_mailbox) {
        // the caller's work that does not depend on the pose (the next frame's pre-amble) goes behind the chain first and runs while
        // this thread looks at the word
        if (enqueue_more) enqueue_more(user);
        volatile unsigned* word = reinterpret_cast<volatile unsigned*>(mb.host + 15);
        // bounded: every 4096 looks the stream is asked whether it is still running -- a launch that failed on the device, or a word
        // that never arrives, ends in the synchronising path below instead of a hang
        bool arrived = false;
        for (unsigned spins = 0; !arrived; ++spins) {
            if (__atomic_load_n(const_cast<unsigned*>(word), __ATOMIC_ACQUIRE) == seq) { arrived = true; break; }
            __builtin_ia32_pause();
            if ((spins & 4095u) == 4095u) {
                const hipError_t q = hipStreamQuery(s);
                if (q != hipErrorNotReady) {
                    (void)hipGetLastError();
                    arrived = __atomic_load_n(const_cast<unsigned*>(word), __ATOMIC_ACQUIRE) == seq;
                    break;
                }
            }
        }
        if (arrived) {
            deliver(mb.host);
            return 0;
        }
        enqueue_more = nullptr;   // (already called)
    }
    thread_local double* stage = nullptr;
    if (!stage && hipHostMalloc((void**)&stage, 15 * sizeof(double), hipHostMallocDefault) != hipSuccess) return set_error(KFX_E_RANGE, "kfx_icp_refine: pinned staging");
    hipError_t e = hipMemcpyAsync(stage, st, 15 * sizeof(double), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && enqueue_more) {
        // the caller has work that does not depend on the pose (the next frame's pre-amble): enqueued behind the read-back, it
        // runs while this thread wakes up from the wait -- which is for the read-back only, not for the stream
        thread_local hipEvent_t arrived = nullptr;
        if (!arrived) e = hipEventCreateWithFlags(&arrived, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(arrived, s);
        if (e == hipSuccess) {
            enqueue_more(user);
            e = hipEventSynchronize(arrived);
        }
    } else if (e == hipSuccess) {
        e = hipStreamSynchronize(s);
    }
    if (e != hipSuccess) return set_error((int)e, hipGetErrorString(e));
    deliver(stage);
    return 0;
}

extern "C" int kfx_icp_refine(const kfx_icp_level* levels, int n_levels, float c, float max_rmse, const kfx_image* workspace,
                              const kfx_image* debug, double T_lp[12], float* rmse, unsigned* obs, int* tracking_good, kfx_stream stream)
{
    return icp_refine_impl(levels, n_levels, c, max_rmse, workspace, debug, T_lp, rmse, obs, tracking_good, nullptr, nullptr, stream);
}

// kfx_icp_refine with a hook between enqueueing the pose's read-back and waiting for it (include/kfx.h)
extern "C" int kfx_icp_refine_then(const kfx_icp_level* levels, int n_levels, float c, float max_rmse, const kfx_image* workspace,
                                   const kfx_image* debug, double T_lp[12], float* rmse, unsigned* obs, int* tracking_good,
                                   void (*enqueue_more)(void* user), void* user, kfx_stream stream)
{
    return icp_refine_impl(levels, n_levels, c, max_rmse, workspace, debug, T_lp, rmse, obs, tracking_good, enqueue_more, user, stream);
}
