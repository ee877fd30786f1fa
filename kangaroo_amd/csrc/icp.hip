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

struct IcpParams {
    ImgView Pl, Pr, Nr;
    unsigned char* dbg; // may be null
    size_t dbg_pitch;
    Pose KT_lr, T_rl;
    float c;
    float* sums;        // gridDim.x * gridDim.y systems of 29 words
};

__global__ __launch_bounds__(256) void k_icp_point_plane(const IcpParams p)
{
    extern __shared__ float lds[];
    const int n = blockDim.x * blockDim.y;
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const unsigned u = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned v = blockIdx.y * blockDim.y + threadIdx.y;

    Lss sum;
    lss_zero(sum);

    const float4 Pr = row<float4>(p.Pr, v)[u];
    const float4 Nr = row<float4>(p.Nr, v)[u];
    const V3 KPl = se3_mul(p.KT_lr, v3(Pr.x, Pr.y, Pr.z));
    const float plx = KPl.x / KPl.z, ply = KPl.y / KPl.z;
    float4 dbg;
    // Image::InBounds(pl, 3): border <= x && x < (w - border) with w converted to float (Image.h:288-291)
    if (isfinite(Pr.z) && Nr.w == 1.0f && 3.0f <= plx && plx < ((float)p.Pl.w - 3.0f) && 3.0f <= ply && ply < ((float)p.Pl.h - 3.0f)) {
        // GetNearestNeighbour: Get(u + 0.5, v + 0.5) -- the sum is a double, truncated to int (Image.h:337-340)
        const int nx = (int)((double)plx + 0.5), ny = (int)((double)ply + 0.5);
        const float4 Pl = row<float4>(p.Pl, ny)[nx];
        if (isfinite(Pl.z)) {
            const V3 _Pr = se3_mul(p.T_rl, v3(Pl.x, Pl.y, Pl.z));
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
            const float roc = y / p.c;
            const float omroc2 = 1.0f - roc * roc;
            const float tukey = (absr <= p.c) ? omroc2 * omroc2 : 0.0f;
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
    if (p.dbg) reinterpret_cast<float4*>(p.dbg + (size_t)v * p.dbg_pitch)[u] = dbg;

    lss_tree(sum, tid, n, lds);
    if (tid == 0) lss_store(p.sums + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * LSS_WORDS, sum);
}

// sum of the per-block systems, fixed order; the result replaces sums[0..28]
__global__ __launch_bounds__(256) void k_lss_final(float* sums, const int nblocks)
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
    if (tid == 0) lss_store(sums, acc);
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
    p.sums = (float*)workspace->ptr;
    const int n = (int)(bx * by);
    const size_t lds_bytes = n >= 128 ? (size_t)LSS_WORDS * (n / 2) * sizeof(float) : 0;
    hipLaunchKernelGGL(k_icp_point_plane, grid, block, lds_bytes, (hipStream_t)stream, p);
    hipLaunchKernelGGL(k_lss_final, dim3(1), dim3(256), 0, (hipStream_t)stream, p.sums, (int)nblocks);
    int st = check_launch("kfx_icp_point_plane");
    if (st) return st;
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
