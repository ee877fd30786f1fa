// preprocess.hip -- depth-map preprocessing for gfx950: BilateralFilter, DepthToVbo,
// NormalsFromVbo.
//
// Reference behaviour: src/cu_bilateral.cu:13-104, src/cu_depth_tools.cu:59-78,
// src/cu_normals.cu:12-45.  New kernels: the bilateral window is served from an LDS
// tile (workgroup tile + apron, loaded once with clamped coordinates, which is exactly
// Image::GetWithClampedRange) and the (2r+1)^2 spatial weights are evaluated once per
// workgroup into LDS instead of once per tap per pixel.
#include <cstdlib>

#include "kfx_device.h"

namespace kfx {

constexpr int BIL_MAX_R = 16; // LDS path up to a 33x33 window

struct BilParams {
    const unsigned char* in;
    size_t in_pitch;
    unsigned char* out;
    size_t out_pitch;
    int w, h;       // output extent (dOut.InBounds, cu_bilateral.cu:20,66)
    int iw, ih;     // input extent used for clamping (Image.h:297-303)
    int R;
    float gs, gr;
    float minval;
    int use_minval;
};

template <typename Ti>
__device__ __forceinline__ float load_clamped(const BilParams& p, int x, int y)
{
    x = min(max(x, 0), p.iw - 1);
    y = min(max(y, 0), p.ih - 1);
    return (float)reinterpret_cast<const Ti*>(p.in + (size_t)y * p.in_pitch)[x];
}

// Workgroup = TX x TY threads (256), each thread filters PY pixels of one column, so the LDS
// tile is (TX + 2R) x (TY*PY + 2R): taller tiles amortise the apron (7x7 window at TX x TY*PY =
// 32x8: 2.1 loads per output pixel, 32x32: 1.4).  Integer inputs are held as float in LDS: every
// uchar/ushort value and every difference of two is exactly representable, so (float)(p - q),
// q >= minval and w * q give the reference's results.
template <typename Ti, int TX, int TY, int PY>
__global__ __launch_bounds__(TX * TY) void k_bilateral(const BilParams p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = TX * TY;
    const int R = p.R, D = 2 * R + 1;
    const int TW = TX + 2 * R, TH = TY * PY + 2 * R;
    float* tile = lds;          // TH x TW
    float* sw = lds + TW * TH;  // D x D spatial weights
    const int tid = threadIdx.x;
    const int bx = blockIdx.x * TX, by = blockIdx.y * (TY * PY);

    for (int i = tid; i < TW * TH; i += NT) {
        const int ty = i / TW, tx = i - ty * TW;
        tile[i] = load_clamped<Ti>(p, bx + tx - R, by + ty - R);
    }
    for (int i = tid; i < D * D; i += NT) {
        const int r = i / D - R, c = i % D - R;
        const float sd2 = (float)(r * r + c * c);
        sw[i] = __expf(-(sd2) / (2 * p.gs * p.gs));
    }
    __syncthreads();

    const int lx = tid % TX, ly0 = tid / TX;
    const int x = bx + lx;
    if (x >= p.w) return;
    const float inv2gr = 2 * p.gr * p.gr;
#pragma unroll
    for (int k = 0; k < PY; ++k) {
        const int ly = ly0 + k * TY; // rows of one thread are TY apart: a wave still reads whole LDS rows
        const int y = by + ly;
        if (y >= p.h) break;
        const float pc = tile[(ly + R) * TW + lx + R];
        float sum = 0.f, sumw = 0.f;
        if (!p.use_minval || pc >= p.minval) {
            for (int r = 0; r < D; ++r) {
                const float* trow = tile + (ly + r) * TW + lx;
                const float* srow = sw + r * D;
                for (int c = 0; c < D; ++c) {
                    const float q = trow[c];
                    if (!p.use_minval || q >= p.minval) {
                        const float id = pc - q;
                        const float id2 = id * id;
                        const float iw = __expf(-(id2) / inv2gr);
                        const float w = srow[c] * iw;
                        sumw += w;
                        sum += w * q;
                    }
                }
            }
        }
        reinterpret_cast<float*>(p.out + (size_t)y * p.out_pitch)[x] = sum / sumw;
    }
}

// Fast-numerics variant (kfx_set_math_mode(KFX_MATH_FAST)) for a compile-time radius: the exact kernel spends most of
// its time in the correctly rounded division -(id2) / (2 gr^2) of every tap (49 per pixel at R = 3) and in a
// data-dependent branch per tap.  Here the exponent argument is id2 * (-1 / (2 gr^2)) (one reciprocal per thread:
// the argument moves by <= 1.5 ulp, far inside the accuracy of the hardware exp), validity is a select and the
// (2R+1)^2 taps are fully unrolled.  Same tile, same tap order, same hardware exp.
template <typename Ti, int R>
__global__ __launch_bounds__(256) void k_bilateral_fast(const BilParams p)
{
    constexpr int TX = 16, TY = 16, D = 2 * R + 1, TW = TX + 2 * R, TH = TY + 2 * R;
    __shared__ float tile[TH * TW];
    __shared__ float sw[D * D];
    const int tid = threadIdx.x;
    const int bx = blockIdx.x * TX, by = blockIdx.y * TY;
    for (int i = tid; i < TW * TH; i += 256) {
        const int ty = i / TW, tx = i - ty * TW;
        tile[i] = load_clamped<Ti>(p, bx + tx - R, by + ty - R);
    }
    if (tid < D * D) {
        const int r = tid / D - R, c = tid % D - R;
        sw[tid] = __expf(-((float)(r * r + c * c)) / (2 * p.gs * p.gs));
    }
    __syncthreads();
    const int lx = tid % TX, ly = tid / TX;
    const int x = bx + lx, y = by + ly;
    if (x >= p.w || y >= p.h) return;
    const float k = -1.0f / (2 * p.gr * p.gr);
    const float pc = tile[(ly + R) * TW + lx + R];
    const bool check = p.use_minval != 0;
    float sum = 0.f, sumw = 0.f;
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const float q = tile[(ly + r) * TW + lx + c];
            const bool ok = !check || q >= p.minval;
            const float id = pc - q;
            const float w = sw[r * D + c] * __expf(id * id * k);
            sumw += ok ? w : 0.f;
            sum += ok ? w * q : 0.f;
        }
    if (check && !(pc >= p.minval)) sum = sumw = 0.f; // the reference skips the window: 0 / 0 = NaN
    reinterpret_cast<float*>(p.out + (size_t)y * p.out_pitch)[x] = sum / sumw;
}

// Window too large for the LDS tile: straight global gathers.
template <typename Ti>
__global__ __launch_bounds__(256) void k_bilateral_global(const BilParams p)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.w || y >= p.h) return;
    const float pc = (float)reinterpret_cast<const Ti*>(p.in + (size_t)y * p.in_pitch)[x];
    float sum = 0.f, sumw = 0.f;
    if (!p.use_minval || pc >= p.minval) {
        for (int r = -p.R; r <= p.R; ++r)
            for (int c = -p.R; c <= p.R; ++c) {
                const float q = load_clamped<Ti>(p, x + c, y + r);
                if (!p.use_minval || q >= p.minval) {
                    const float sd2 = (float)(r * r + c * c);
                    const float id = pc - q;
                    const float id2 = id * id;
                    const float w = __expf(-(sd2) / (2 * p.gs * p.gs)) * __expf(-(id2) / (2 * p.gr * p.gr));
                    sumw += w;
                    sum += w * q;
                }
            }
    }
    reinterpret_cast<float*>(p.out + (size_t)y * p.out_pitch)[x] = sum / sumw;
}

// Joint ("cross") bilateral filter with an external guide image (cu_bilateral.cu:110-143): weights
// exp(-sd2 / 2gs^2) * exp(-rd2 / 2gr^2) * exp(-cd2 / 2gc^2) with cd the guide difference; sumw == 0 keeps the input.
struct GuidedParams {
    BilParams b;
    const unsigned char* guide;
    size_t gpitch;
    int gw, gh;
    float gc;
};
template <typename Tg>
__global__ __launch_bounds__(256) void k_bilateral_guided(const GuidedParams g)
{
    const BilParams& p = g.b;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.w || y >= p.h) return;
    const float pv = reinterpret_cast<const float*>(p.in + (size_t)y * p.in_pitch)[x];
    const float pcv = (float)reinterpret_cast<const Tg*>(g.guide + (size_t)y * g.gpitch)[x];
    float sum = 0.f, sumw = 0.f;
    for (int r = -p.R; r <= p.R; ++r)
        for (int c = -p.R; c <= p.R; ++c) {
            const float q = load_clamped<float>(p, x + c, y + r);
            const int gx = min(max(x + c, 0), g.gw - 1), gy = min(max(y + r, 0), g.gh - 1);
            const float qc = (float)reinterpret_cast<const Tg*>(g.guide + (size_t)gy * g.gpitch)[gx];
            const float rd = pv - q, cd = pcv - qc;
            const float sd2 = (float)(r * r + c * c);
            const float rd2 = rd * rd, cd2 = cd * cd;
            const float sw = __expf(-(sd2) / (2 * p.gs * p.gs));
            const float rw = __expf(-(rd2) / (2 * p.gr * p.gr));
            const float cw = __expf(-(cd2) / (2 * g.gc * g.gc));
            const float w = sw * rw * cw;
            sumw += w;
            sum += w * q;
        }
    reinterpret_cast<float*>(p.out + (size_t)y * p.out_pitch)[x] = sumw == 0 ? pv : sum / sumw;
}

struct VboParams {
    const unsigned char* in;
    size_t in_pitch;
    unsigned char* out;
    size_t out_pitch;
    int w, h;
    Intr K;
    float scale;
};

// KernDepthToVbo (cu_depth_tools.cu:59-70), Unproject(u,v,z) (ImageIntrinsics.h:127-131)
template <typename Ti>
__global__ __launch_bounds__(256) void k_depth_to_vbo(const VboParams p)
{
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= p.w || v >= p.h) return;
    const float kz = p.scale * (float)reinterpret_cast<const Ti*>(p.in + (size_t)v * p.in_pitch)[u];
    const float4 P = make_float4(kz * ((float)u - p.K.u0) / p.K.fu, kz * ((float)v - p.K.v0) / p.K.fv, kz, 1.0f);
    reinterpret_cast<float4*>(p.out + (size_t)v * p.out_pitch)[u] = P;
}

struct NrmParams {
    const unsigned char* in;
    size_t in_pitch;
    unsigned char* out;
    size_t out_pitch;
    int w, h;
};

// KernNormalsFromVbo (cu_normals.cu:12-38)
__global__ __launch_bounds__(256) void k_normals_from_vbo(const NrmParams p)
{
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= p.w || v >= p.h) return;
    float4 N = make_float4(0.f, 0.f, 0.f, 0.f);
    if (u + 1 < p.w && v + 1 < p.h) {
        const float4* r0 = reinterpret_cast<const float4*>(p.in + (size_t)v * p.in_pitch);
        const float4* r1 = reinterpret_cast<const float4*>(p.in + (size_t)(v + 1) * p.in_pitch);
        const float4 Vc = r0[u], Vr = r0[u + 1], Vu = r1[u];
        const V3 a = v3(Vr.x - Vc.x, Vr.y - Vc.y, Vr.z - Vc.z);
        const V3 b = v3(Vu.x - Vc.x, Vu.y - Vc.y, Vu.z - Vc.z);
        const V3 axb = v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
        const float mag = length(axb);
        N = make_float4(-axb.x / mag, -axb.y / mag, -axb.z / mag, 1.0f);
    }
    reinterpret_cast<float4*>(p.out + (size_t)v * p.out_pitch)[u] = N;
}

// DepthToVbo<float> and NormalsFromVbo in one launch (both are latency-sized at VGA: two 4-5 us launches become one).
// A pixel's normal needs the vertices at (u+1, v) and (u, v+1); they are recomputed from the depth image with the
// expression DepthToVbo uses, so both outputs carry exactly the bits the two separate operators write.
struct VboNrmParams {
    const unsigned char* in;
    size_t in_pitch;
    unsigned char *vbo, *nrm;
    size_t vpitch, npitch;
    int w, h;
    Intr K;
    float scale;
    unsigned char* tex;   // optional: the packed texel image {nx, ny, nz, depth} the tiled SdfFuse kernels stage by LDS-DMA (fuse.hip)
    size_t tpitch;
    float* bmax;          // ... with the maxima of its 8 x 4 pixel blocks behind it (rows of bw8 floats)
    unsigned bw8;
};
__device__ __forceinline__ float4 vertex_of(const VboNrmParams& p, int u, int v)
{
    const float kz = p.scale * reinterpret_cast<const float*>(p.in + (size_t)v * p.in_pitch)[u];
    return make_float4(kz * ((float)u - p.K.u0) / p.K.fu, kz * ((float)v - p.K.v0) / p.K.fv, kz, 1.0f);
}
__global__ __launch_bounds__(256) void k_vbo_normals_f32(const VboNrmParams p)
{
    __shared__ float s_max[4][8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int u = blockIdx.x * 64 + lane, v = blockIdx.y * 4 + wv;
    float d = -__builtin_inff();
    if (u < p.w && v < p.h) {
        const float4 Vc = vertex_of(p, u, v);
        reinterpret_cast<float4*>(p.vbo + (size_t)v * p.vpitch)[u] = Vc;
        float4 N = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u + 1 < p.w && v + 1 < p.h) {
            const float4 Vr = vertex_of(p, u + 1, v), Vu = vertex_of(p, u, v + 1);
            const V3 a = v3(Vr.x - Vc.x, Vr.y - Vc.y, Vr.z - Vc.z);
            const V3 b = v3(Vu.x - Vc.x, Vu.y - Vc.y, Vu.z - Vc.z);
            const V3 axb = v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
            const float mag = length(axb);
            N = make_float4(-axb.x / mag, -axb.y / mag, -axb.z / mag, 1.0f);
        }
        reinterpret_cast<float4*>(p.nrm + (size_t)v * p.npitch)[u] = N;
        if (p.tex) {   // (uniform) what k_pack_texels (fuse.hip) would write: the normal's xyz and the depth image's own value ...
            d = reinterpret_cast<const float*>(p.in + (size_t)v * p.in_pitch)[u];
            reinterpret_cast<float4*>(p.tex + (size_t)v * p.tpitch)[u] = make_float4(N.x, N.y, N.z, d);
        }
    }
    if (p.tex) {   // ... and the maximum finite depth of each of the tile's eight 8 x 4 pixel blocks
        d = wave8_combine(fmaxf(d, -__builtin_inff()), [](float a, float b) { return fmaxf(a, b); });
        if ((lane & 7) == 0) s_max[wv][lane >> 3] = d;
        __syncthreads();
        if (threadIdx.x < 8)
            p.bmax[(size_t)blockIdx.y * p.bw8 + blockIdx.x * 8 + threadIdx.x] =
                fmaxf(fmaxf(s_max[0][threadIdx.x], s_max[1][threadIdx.x]), fmaxf(s_max[2][threadIdx.x], s_max[3][threadIdx.x]));
    }
}

struct EwParams {
    const unsigned char* in;
    size_t in_pitch;
    unsigned char* out;
    size_t out_pitch;
    int w, h;
    float s, offset;
};

// KernElementwiseScaleBias<float,float,float> (cu_operations.cu:39-49): b = s*a + offset
__global__ __launch_bounds__(256) void k_scale_bias_f32(const EwParams p)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.w || y >= p.h) return;
    const float v1 = reinterpret_cast<const float*>(p.in + (size_t)y * p.in_pitch)[x];
    reinterpret_cast<float*>(p.out + (size_t)y * p.out_pitch)[x] = p.s * v1 + p.offset;
}

// KernBoxHalfIgnoreInvalid<float,float,float> (cu_resample.cu:89-111): mean of the finite samples of
// each 2x2 block, NaN if none; samples are added in the order tl, tr, bl, br
__global__ __launch_bounds__(256) void k_box_half_ignore_invalid_f32(const EwParams p)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.w || y >= p.h) return;
    const float2 t = *reinterpret_cast<const float2*>(p.in + (size_t)(2 * y) * p.in_pitch + (size_t)(2 * x) * 4);
    const float2 b = *reinterpret_cast<const float2*>(p.in + (size_t)(2 * y + 1) * p.in_pitch + (size_t)(2 * x) * 4);
    int n = 0;
    float sum = 0;
    if (isfinite(t.x)) { sum += t.x; n++; }
    if (isfinite(t.y)) { sum += t.y; n++; }
    if (isfinite(b.x)) { sum += b.x; n++; }
    if (isfinite(b.y)) { sum += b.y; n++; }
    reinterpret_cast<float*>(p.out + (size_t)y * p.out_pitch)[x] = n > 0 ? (sum / n) : __builtin_nanf("");
}


// ---- the small per-pixel tools of cu_depth_tools.h ------------------------------------------------------------
struct PixIO {
    const unsigned char* in;
    size_t ipitch;
    unsigned char* out;
    size_t opitch;
    int w, h;
};

// Disp2Depth (cu_depth_tools.cu:15-23): depth = fu * baseline / disparity, NaN below the minimum disparity
__global__ __launch_bounds__(256) void k_disp2depth(const PixIO p, float fu, float baseline, float min_disp)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.w || y >= p.h) return;
    const float d = *(reinterpret_cast<const float*>(p.in + (size_t)y * p.ipitch) + x);
    *(reinterpret_cast<float*>(p.out + (size_t)y * p.opitch) + x) = d >= min_disp ? fu * baseline / d : __builtin_nanf("");
}

// FilterBadKinectData (cu_depth_tools.cu:32-39): millimetre readings below 200 become NaN
template <typename Ti>
__global__ __launch_bounds__(256) void k_filter_bad_kinect(const PixIO p)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.w || y >= p.h) return;
    const float z_mm = (float)*(reinterpret_cast<const Ti*>(p.in + (size_t)y * p.ipitch) + x);
    *(reinterpret_cast<float*>(p.out + (size_t)y * p.opitch) + x) = z_mm >= 200 ? z_mm : __builtin_nanf("");
}

struct ColourVboParams {
    const unsigned char* vbo;   // Image<float4>
    size_t vpitch;
    const unsigned char* rgb;   // Image<uchar3>
    size_t rpitch;
    int rw, rh;
    unsigned char* out;         // Image<uchar4>
    size_t opitch;
    int w, h;
    Pose KT;                    // KT_cd
};
struct __attribute__((packed)) Rgb3 { unsigned char x, y, z; };

// ColourVbo (cu_depth_tools.cu:86-112): project every vertex with KT_cd, bilinear RGB (sampling.h lerp(uchar3...):
// integer difference, converted, times t, plus the first value; rows blended as float3), truncated to bytes
__global__ __launch_bounds__(256) void k_colour_vbo(const ColourVboParams p)
{
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= p.w || v >= p.h) return;
    const float4 Pd = *(reinterpret_cast<const float4*>(p.vbo + (size_t)v * p.vpitch) + u);
    // Mat<3,4> * Mat<4,1> with w = 1: row . (x, y, z, 1), summed left to right (Mat.h operator*)
    const float k0 = p.KT.m[0] * Pd.x + p.KT.m[1] * Pd.y + p.KT.m[2] * Pd.z + p.KT.m[3] * 1.0f;
    const float k1 = p.KT.m[4] * Pd.x + p.KT.m[5] * Pd.y + p.KT.m[6] * Pd.z + p.KT.m[7] * 1.0f;
    const float k2 = p.KT.m[8] * Pd.x + p.KT.m[9] * Pd.y + p.KT.m[10] * Pd.z + p.KT.m[11] * 1.0f;
    const float pu = k0 / k2, pv = k1 / k2;
    uchar4 id = make_uchar4(0, 0, 0, 0);
    if (1.0f <= pu && pu < ((float)p.rw - 1.0f) && 1.0f <= pv && pv < ((float)p.rh - 1.0f)) {
        const float ix = floorf(pu), iy = floorf(pv);
        const float fx = pu - ix, fy = pv - iy;
        const Rgb3* bl = reinterpret_cast<const Rgb3*>(p.rgb + (size_t)iy * p.rpitch) + (size_t)ix;
        const Rgb3* tl = reinterpret_cast<const Rgb3*>(p.rgb + (size_t)(iy + 1) * p.rpitch) + (size_t)ix;
        const Rgb3 b0 = bl[0], b1 = bl[1], t0 = tl[0], t1 = tl[1];
        const V3 lo = v3((float)b0.x + fx * (float)((int)b1.x - (int)b0.x), (float)b0.y + fx * (float)((int)b1.y - (int)b0.y),
                         (float)b0.z + fx * (float)((int)b1.z - (int)b0.z));
        const V3 hi = v3((float)t0.x + fx * (float)((int)t1.x - (int)t0.x), (float)t0.y + fx * (float)((int)t1.y - (int)t0.y),
                         (float)t0.z + fx * (float)((int)t1.z - (int)t0.z));
        const V3 c = v3(lo.x + fy * (hi.x - lo.x), lo.y + fy * (hi.y - lo.y), lo.z + fy * (hi.z - lo.z));
        id = make_uchar4((unsigned char)c.x, (unsigned char)c.y, (unsigned char)c.z, 255);
    }
    *(reinterpret_cast<uchar4*>(p.out + (size_t)v * p.opitch) + u) = id;
}

// ---- TextureDepth (cu_depth_tools.cu:123-207): colour a rendered depth image from RGB keyframes ------------------
struct KeyframeView {          // ImageKeyframe<uchar3>: {ImageIntrinsics K; Mat<float,3,4> T_iw; Image<uchar3> img}
    Intr K;
    Pose T_iw;
    const unsigned char* img;  // null: slot unused (ends the list, as `kfs[k].img.ptr` does in the reference)
    size_t pitch;
    int w, h;
};
constexpr int TEX_MAX_KF = 10;
struct TextureParams {
    unsigned char* out;        // Image<float4>
    size_t opitch;
    const unsigned char *depth, *norm, *phong;   // Image<float>, Image<float4>, Image<float> (phong: multi-keyframe variant)
    size_t dpitch, npitch, ppitch;
    int w, h, nkf, single;
    Pose T_wd;
    Intr Kd;
    KeyframeView kf[TEX_MAX_KF];
};
// Image<uchar3>::GetBilinear<float3> (see k_colour_vbo)
__device__ __forceinline__ V3 rgb_bilinear(const KeyframeView& k, float pu, float pv)
{
    const float ix = floorf(pu), iy = floorf(pv);
    const float fx = pu - ix, fy = pv - iy;
    const Rgb3* bl = reinterpret_cast<const Rgb3*>(k.img + (size_t)iy * k.pitch) + (size_t)ix;
    const Rgb3* tl = reinterpret_cast<const Rgb3*>(k.img + (size_t)(iy + 1) * k.pitch) + (size_t)ix;
    const Rgb3 b0 = bl[0], b1 = bl[1], t0 = tl[0], t1 = tl[1];
    const V3 lo = v3((float)b0.x + fx * (float)((int)b1.x - (int)b0.x), (float)b0.y + fx * (float)((int)b1.y - (int)b0.y),
                     (float)b0.z + fx * (float)((int)b1.z - (int)b0.z));
    const V3 hi = v3((float)t0.x + fx * (float)((int)t1.x - (int)t0.x), (float)t0.y + fx * (float)((int)t1.y - (int)t0.y),
                     (float)t0.z + fx * (float)((int)t1.z - (int)t0.z));
    return v3(lo.x + fy * (hi.x - lo.x), lo.y + fy * (hi.y - lo.y), lo.z + fy * (hi.z - lo.z));
}
__global__ __launch_bounds__(256) void k_texture_depth(const TextureParams p)
{
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= p.w || v >= p.h) return;
    const float d = reinterpret_cast<const float*>(p.depth + (size_t)v * p.dpitch)[u];
    const float4 N_d = reinterpret_cast<const float4*>(p.norm + (size_t)v * p.npitch)[u];
    const V3 N_w = so3_mul(p.T_wd, v3(N_d.x, N_d.y, N_d.z));
    const V3 P_d = v3(d * ((float)u - p.Kd.u0) / p.Kd.fu, d * ((float)v - p.Kd.v0) / p.Kd.fv, d);   // Unproject(u, v, z)
    const V3 P_w = se3_mul(p.T_wd, P_d);
    float4 out;
    if (p.single) { // one keyframe: nearest-facing test on the camera-frame normal's z (cu_depth_tools.cu:123-149)
        const KeyframeView& k = p.kf[0];
        const V3 P_kf = se3_mul(k.T_iw, P_w);
        const float pu = k.K.u0 + k.K.fu * P_kf.x / P_kf.z, pv = k.K.v0 + k.K.fv * P_kf.y / P_kf.z;
        const V3 N_c = so3_mul(k.T_iw, N_w);
        const float facing = N_c.x * 0.f + N_c.y * 0.f + N_c.z * 1.f;
        if (2.0f <= pu && pu < ((float)k.w - 2.0f) && 2.0f <= pv && pv < ((float)k.h - 2.0f) && (double)facing < -0.2) { // the literal is a double
            const V3 c = rgb_bilinear(k, pu, pv) * (1.0f / 255.0f);
            out = make_float4(c.x, c.y, c.z, 1.f);
        } else {
            out = make_float4(0.f, 0.f, 0.f, 1.f);
        }
    } else {        // up to 10 keyframes blended by the cosine to the viewing ray (:164-197)
        float w = 0.f;
        V3 color = v3(0.f, 0.f, 0.f);   // the reference leaves `color` uninitialised before accumulating; zero here
        for (int i = 0; i < p.nkf && p.kf[i].img; ++i) {
            const KeyframeView& k = p.kf[i];
            const V3 P_kf = se3_mul(k.T_iw, P_w);
            const float pu = k.K.u0 + k.K.fu * P_kf.x / P_kf.z, pv = k.K.v0 + k.K.fv * P_kf.y / P_kf.z;
            const V3 N_c = so3_mul(k.T_iw, N_w);
            const float ndot = dot(N_c, P_kf) / -length(P_kf);
            if (2.0f <= pu && pu < ((float)k.w - 2.0f) && 2.0f <= pv && pv < ((float)k.h - 2.0f) && (double)ndot > 0.1 && P_kf.z > 0.f) {
                color = color + rgb_bilinear(k, pu, pv) * (ndot / 255.0f);
                w += ndot;
            }
        }
        if (w == 0.f) {
            w = 1.f;
            const float ph = reinterpret_cast<const float*>(p.phong + (size_t)v * p.ppitch)[u];
            color = v3(ph, ph, ph);
        }
        const V3 c = div_s(color, w);
        out = make_float4(c.x, c.y, c.z, 1.f);
    }
    reinterpret_cast<float4*>(p.out + (size_t)v * p.opitch)[u] = out;
}

} // namespace kfx

using namespace kfx;

static int check_image(const kfx_image* im, size_t elem, const char* what)
{
    if (!im || !im->ptr) return set_error(KFX_E_NULL, what);
    if (im->pitch < im->w * elem) return set_error(KFX_E_SHAPE, what);
    const size_t al = elem >= 16 ? 16 : elem;
    if (((uintptr_t)im->ptr | im->pitch) & (al - 1)) return set_error(KFX_E_ALIGN, what);
    if (im->w > (1u << 30) || im->h > (1u << 30)) return set_error(KFX_E_SHAPE, what);
    return 0;
}

template <typename Ti>
static int bilateral_launch(const kfx_image* out, const kfx_image* in, float gs, float gr, unsigned size,
                            float minval, int use_minval, kfx_stream stream)
{
    if (int e = check_image(out, 4, "BilateralFilter: output image")) return e;
    if (int e = check_image(in, sizeof(Ti), "BilateralFilter: input image")) return e;
    if (out->w == 0 || out->h == 0) return 0;
    if (in->w == 0 || in->h == 0) return set_error(KFX_E_SHAPE, "BilateralFilter: empty input");
    if (in->w < out->w || in->h < out->h) return set_error(KFX_E_SHAPE, "BilateralFilter: input smaller than output");
    if (size > 1024) return set_error(KFX_E_RANGE, "BilateralFilter: window radius");
    BilParams p;
    p.in = (const unsigned char*)in->ptr;
    p.in_pitch = in->pitch;
    p.out = (unsigned char*)out->ptr;
    p.out_pitch = out->pitch;
    p.w = (int)out->w;
    p.h = (int)out->h;
    p.iw = (int)in->w;
    p.ih = (int)in->h;
    p.R = (int)size;
    p.gs = gs;
    p.gr = gr;
    p.minval = minval;
    p.use_minval = use_minval;
    hipStream_t s = (hipStream_t)stream;
    if (math_mode() == KFX_MATH_FAST && p.R >= 1 && p.R <= 3) {
        dim3 grid(ceil_div(p.w, 16), ceil_div(p.h, 16));
        if (p.R == 3) hipLaunchKernelGGL((k_bilateral_fast<Ti, 3>), grid, dim3(256), 0, s, p);
        else if (p.R == 2) hipLaunchKernelGGL((k_bilateral_fast<Ti, 2>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_bilateral_fast<Ti, 1>), grid, dim3(256), 0, s, p);
    } else if (p.R <= BIL_MAX_R) {
        const int D = 2 * p.R + 1;
        // tile shape (config C3 sweep, scripts/config_sweep.py): KFX_BILATERAL_TILE = 0..5
        static const int shape = [] { const char* e = getenv("KFX_BILATERAL_TILE"); return e ? atoi(e) : 2; }();
#define KFX_BIL(TX_, TY_, PY_)                                                                                   \
    do {                                                                                                        \
        const size_t lds = (size_t)((TX_ + 2 * p.R) * (TY_ * PY_ + 2 * p.R) + D * D) * sizeof(float);           \
        dim3 grid(ceil_div(p.w, TX_), ceil_div(p.h, TY_ * PY_));                                                \
        hipLaunchKernelGGL((k_bilateral<Ti, TX_, TY_, PY_>), grid, dim3(TX_ * TY_), lds, s, p);                 \
    } while (0)
        switch (shape) {
        case 0: KFX_BIL(32, 8, 1); break;
        case 1: KFX_BIL(64, 4, 1); break;
        case 3: KFX_BIL(32, 8, 2); break;
        case 4: KFX_BIL(32, 8, 4); break;
        case 5: KFX_BIL(64, 4, 4); break;
        default: KFX_BIL(16, 16, 1); break; // best of the sweep at 1280x960 (profiles/r01_config_sweep.txt)
        }
#undef KFX_BIL
    } else {
        dim3 grid(ceil_div(p.w, 64), ceil_div(p.h, 4));
        hipLaunchKernelGGL(k_bilateral_global<Ti>, grid, dim3(256), 0, s, p);
    }
    return check_launch("kfx_bilateral");
}

extern "C" int kfx_bilateral_f32(const kfx_image* out, const kfx_image* in, float gs, float gr, unsigned size,
                                 float minval, int use_minval, kfx_stream stream)
{
    return bilateral_launch<float>(out, in, gs, gr, size, minval, use_minval ? 1 : 0, stream);
}
extern "C" int kfx_bilateral_u16(const kfx_image* out, const kfx_image* in, float gs, float gr, unsigned size,
                                 unsigned short minval, kfx_stream stream)
{
    return bilateral_launch<unsigned short>(out, in, gs, gr, size, (float)minval, 1, stream);
}
extern "C" int kfx_bilateral_u8(const kfx_image* out, const kfx_image* in, float gs, float gr, unsigned size,
                                kfx_stream stream)
{
    return bilateral_launch<unsigned char>(out, in, gs, gr, size, 0.f, 0, stream);
}

template <typename Ti>
static int vbo_launch(const kfx_image* vbo, const kfx_image* depth, const float K[4], float scale, kfx_stream stream)
{
    if (int e = check_image(vbo, 16, "DepthToVbo: vbo image")) return e;
    if (int e = check_image(depth, sizeof(Ti), "DepthToVbo: depth image")) return e;
    if (!K) return set_error(KFX_E_NULL, "DepthToVbo: null intrinsics");
    if (vbo->w == 0 || vbo->h == 0) return 0;
    if (depth->w < vbo->w || depth->h < vbo->h) return set_error(KFX_E_SHAPE, "DepthToVbo: depth smaller than vbo");
    VboParams p{(const unsigned char*)depth->ptr, depth->pitch, (unsigned char*)vbo->ptr, vbo->pitch,
                (int)vbo->w, (int)vbo->h, Intr{K[0], K[1], K[2], K[3]}, scale};
    dim3 grid(ceil_div(p.w, 64), ceil_div(p.h, 4));
    hipLaunchKernelGGL(k_depth_to_vbo<Ti>, grid, dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_depth_to_vbo");
}

extern "C" int kfx_depth_to_vbo_f32(const kfx_image* vbo, const kfx_image* depth, const float K[4], float scale,
                                    kfx_stream stream)
{
    return vbo_launch<float>(vbo, depth, K, scale, stream);
}
extern "C" int kfx_depth_to_vbo_u16(const kfx_image* vbo, const kfx_image* depth, const float K[4], float scale,
                                    kfx_stream stream)
{
    return vbo_launch<unsigned short>(vbo, depth, K, scale, stream);
}

extern "C" int kfx_normals_from_vbo(const kfx_image* nrm, const kfx_image* vbo, kfx_stream stream)
{
    if (int e = check_image(nrm, 16, "NormalsFromVbo: normal image")) return e;
    if (int e = check_image(vbo, 16, "NormalsFromVbo: vbo image")) return e;
    if (nrm->w == 0 || nrm->h == 0) return 0;
    if (vbo->w < nrm->w || vbo->h < nrm->h) return set_error(KFX_E_SHAPE, "NormalsFromVbo: vbo smaller than normals");
    NrmParams p{(const unsigned char*)vbo->ptr, vbo->pitch, (unsigned char*)nrm->ptr, nrm->pitch, (int)nrm->w, (int)nrm->h};
    dim3 grid(ceil_div(p.w, 64), ceil_div(p.h, 4));
    hipLaunchKernelGGL(k_normals_from_vbo, grid, dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_normals_from_vbo");
}

extern "C" int kfx_elementwise_scale_bias_f32(const kfx_image* out, const kfx_image* in, float s, float offset, kfx_stream stream)
{
    if (int e = check_image(out, 4, "ElementwiseScaleBias: output image")) return e;
    if (int e = check_image(in, 4, "ElementwiseScaleBias: input image")) return e;
    if (out->w == 0 || out->h == 0) return 0;
    if (in->w < out->w || in->h < out->h) return set_error(KFX_E_SHAPE, "ElementwiseScaleBias: input smaller than output");
    EwParams p{(const unsigned char*)in->ptr, in->pitch, (unsigned char*)out->ptr, out->pitch, (int)out->w, (int)out->h, s, offset};
    dim3 grid(ceil_div(p.w, 64), ceil_div(p.h, 4));
    hipLaunchKernelGGL(k_scale_bias_f32, grid, dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_elementwise_scale_bias_f32");
}

extern "C" int kfx_box_half_ignore_invalid_f32(const kfx_image* out, const kfx_image* in, kfx_stream stream)
{
    if (int e = check_image(out, 4, "BoxHalfIgnoreInvalid: output image")) return e;
    if (int e = check_image(in, 4, "BoxHalfIgnoreInvalid: input image")) return e;
    if (out->w == 0 || out->h == 0) return 0;
    if (in->w < 2 * out->w || in->h < 2 * out->h) return set_error(KFX_E_SHAPE, "BoxHalfIgnoreInvalid: input smaller than 2x output");
    if (((uintptr_t)in->ptr | in->pitch) & 7) return set_error(KFX_E_ALIGN, "BoxHalfIgnoreInvalid: input not 8-byte aligned");
    EwParams p{(const unsigned char*)in->ptr, in->pitch, (unsigned char*)out->ptr, out->pitch, (int)out->w, (int)out->h, 0.f, 0.f};
    dim3 grid(ceil_div(p.w, 64), ceil_div(p.h, 4));
    hipLaunchKernelGGL(k_box_half_ignore_invalid_f32, grid, dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_box_half_ignore_invalid_f32");
}

// Disp2Depth(dIn, dOut, fu, fBaseline, fMinDisp) (cu_depth_tools.cu:15-30; the launch is bounded by dOut)
extern "C" int kfx_disp2depth(const kfx_image* in, const kfx_image* out, float fu, float baseline, float min_disp, kfx_stream stream)
{
    if (int e = check_image(out, 4, "Disp2Depth: output image")) return e;
    if (int e = check_image(in, 4, "Disp2Depth: input image")) return e;
    if (out->w == 0 || out->h == 0) return 0;
    if (in->w < out->w || in->h < out->h) return set_error(KFX_E_SHAPE, "Disp2Depth: input smaller than output");
    PixIO p{(const unsigned char*)in->ptr, in->pitch, (unsigned char*)out->ptr, out->pitch, (int)out->w, (int)out->h};
    hipLaunchKernelGGL(k_disp2depth, dim3(ceil_div(p.w, 64), ceil_div(p.h, 4)), dim3(256), 0, (hipStream_t)stream, p, fu, baseline, min_disp);
    return check_launch("kfx_disp2depth");
}

// FilterBadKinectData(dFiltered, dKinectDepth) (cu_depth_tools.cu:32-53), float and unsigned short readings
extern "C" int kfx_filter_bad_kinect_f32(const kfx_image* out, const kfx_image* in, kfx_stream stream)
{
    if (int e = check_image(out, 4, "FilterBadKinectData: output image")) return e;
    if (int e = check_image(in, 4, "FilterBadKinectData: input image")) return e;
    if (out->w == 0 || out->h == 0) return 0;
    if (in->w < out->w || in->h < out->h) return set_error(KFX_E_SHAPE, "FilterBadKinectData: input smaller than output");
    PixIO p{(const unsigned char*)in->ptr, in->pitch, (unsigned char*)out->ptr, out->pitch, (int)out->w, (int)out->h};
    hipLaunchKernelGGL(k_filter_bad_kinect<float>, dim3(ceil_div(p.w, 64), ceil_div(p.h, 4)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_filter_bad_kinect_f32");
}
extern "C" int kfx_filter_bad_kinect_u16(const kfx_image* out, const kfx_image* in, kfx_stream stream)
{
    if (int e = check_image(out, 4, "FilterBadKinectData: output image")) return e;
    if (int e = check_image(in, 2, "FilterBadKinectData: input image")) return e;
    if (out->w == 0 || out->h == 0) return 0;
    if (in->w < out->w || in->h < out->h) return set_error(KFX_E_SHAPE, "FilterBadKinectData: input smaller than output");
    PixIO p{(const unsigned char*)in->ptr, in->pitch, (unsigned char*)out->ptr, out->pitch, (int)out->w, (int)out->h};
    hipLaunchKernelGGL(k_filter_bad_kinect<unsigned short>, dim3(ceil_div(p.w, 64), ceil_div(p.h, 4)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_filter_bad_kinect_u16");
}

// ColourVbo(dId, dPd, dIc, KT_cd) (cu_depth_tools.cu:86-119)
extern "C" int kfx_colour_vbo(const kfx_image* id, const kfx_image* vbo, const kfx_image* rgb, const float KT_cd[12], kfx_stream stream)
{
    if (int e = check_image(id, 4, "ColourVbo: output image")) return e;
    if (int e = check_image(vbo, 16, "ColourVbo: vbo image")) return e;
    if (int e = check_image(rgb, 1, "ColourVbo: rgb image")) return e;
    if (!KT_cd) return set_error(KFX_E_NULL, "ColourVbo: null transform");
    if (id->w == 0 || id->h == 0) return 0;
    if (vbo->w < id->w || vbo->h < id->h || rgb->pitch < rgb->w * 3) return set_error(KFX_E_SHAPE, "ColourVbo: image sizes");
    ColourVboParams p;
    p.vbo = (const unsigned char*)vbo->ptr; p.vpitch = vbo->pitch;
    p.rgb = (const unsigned char*)rgb->ptr; p.rpitch = rgb->pitch; p.rw = (int)rgb->w; p.rh = (int)rgb->h;
    p.out = (unsigned char*)id->ptr; p.opitch = id->pitch; p.w = (int)id->w; p.h = (int)id->h;
    for (int i = 0; i < 12; ++i) p.KT.m[i] = KT_cd[i];
    hipLaunchKernelGGL(k_colour_vbo, dim3(ceil_div(p.w, 64), ceil_div(p.h, 4)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_colour_vbo");
}

// BilateralFilter(dOut, dIn, dImg, gs, gr, gc, size) (cu_bilateral.cu:145-155): guide image float or unsigned char
template <typename Tg>
static int guided_launch(const kfx_image* out, const kfx_image* in, const kfx_image* guide, float gs, float gr, float gc, unsigned size,
                         kfx_stream stream)
{
    if (int e = check_image(out, 4, "BilateralFilter(guided): output image")) return e;
    if (int e = check_image(in, 4, "BilateralFilter(guided): input image")) return e;
    if (int e = check_image(guide, sizeof(Tg), "BilateralFilter(guided): guide image")) return e;
    if (out->w == 0 || out->h == 0) return 0;
    if (in->w < out->w || in->h < out->h || guide->w < out->w || guide->h < out->h)
        return set_error(KFX_E_SHAPE, "BilateralFilter(guided): inputs smaller than the output");
    if (size > 64) return set_error(KFX_E_RANGE, "BilateralFilter(guided): window too large");
    GuidedParams g;
    g.b = BilParams{(const unsigned char*)in->ptr, in->pitch, (unsigned char*)out->ptr, out->pitch, (int)out->w, (int)out->h,
                    (int)in->w, (int)in->h, (int)size, gs, gr, 0.f, 0};
    g.guide = (const unsigned char*)guide->ptr;
    g.gpitch = guide->pitch;
    g.gw = (int)guide->w;
    g.gh = (int)guide->h;
    g.gc = gc;
    hipLaunchKernelGGL(k_bilateral_guided<Tg>, dim3(ceil_div(g.b.w, 64), ceil_div(g.b.h, 4)), dim3(256), 0, (hipStream_t)stream, g);
    return check_launch("kfx_bilateral_guided");
}
extern "C" int kfx_bilateral_guided_f32(const kfx_image* out, const kfx_image* in, const kfx_image* guide, float gs, float gr, float gc,
                                        unsigned size, kfx_stream stream)
{
    return guided_launch<float>(out, in, guide, gs, gr, gc, size, stream);
}
extern "C" int kfx_bilateral_guided_u8(const kfx_image* out, const kfx_image* in, const kfx_image* guide, float gs, float gr, float gc,
                                       unsigned size, kfx_stream stream)
{
    return guided_launch<unsigned char>(out, in, guide, gs, gr, gc, size, stream);
}

// DepthToVbo<float>(vbo, depth, K, scale) followed by NormalsFromVbo(nrm, vbo), one launch, identical outputs
// (no reference counterpart: a launch-count optimisation for the frame pre-amble; vbo, nrm and depth of one size)
namespace kfx {
int depth_to_vbo_normals_texels(const kfx_image* vbo, const kfx_image* nrm, const kfx_image* depth, const float K[4], float scale,
                                const kfx_image* texels, kfx_stream stream);
}
extern "C" int kfx_depth_to_vbo_normals_f32(const kfx_image* vbo, const kfx_image* nrm, const kfx_image* depth, const float K[4], float scale,
                                            kfx_stream stream)
{
    return kfx::depth_to_vbo_normals_texels(vbo, nrm, depth, K, scale, nullptr, stream);
}
// ... and, for kfx_frame_step, the packed texel image of the SdfFuse that follows from the same launch (texels: w x h float4, or null)
int kfx::depth_to_vbo_normals_texels(const kfx_image* vbo, const kfx_image* nrm, const kfx_image* depth, const float K[4], float scale,
                                     const kfx_image* texels, kfx_stream stream)
{
    if (int e = check_image(vbo, 16, "DepthToVbo+Normals: vbo image")) return e;
    if (int e = check_image(nrm, 16, "DepthToVbo+Normals: normal image")) return e;
    if (int e = check_image(depth, 4, "DepthToVbo+Normals: depth image")) return e;
    if (!K) return set_error(KFX_E_NULL, "DepthToVbo+Normals: null intrinsics");
    if (vbo->w == 0 || vbo->h == 0) return 0;
    if (nrm->w != vbo->w || nrm->h != vbo->h || depth->w < vbo->w || depth->h < vbo->h)
        return set_error(KFX_E_SHAPE, "DepthToVbo+Normals: image sizes");
    // texels: a buffer of kfx::texel_image_bytes(w, h) bytes in the layout of fuse.hip's tex_layout -- rows of `pitch` = w * 16 rounded up
    // to 256 bytes, the block maxima behind them in rows of ceil(w / 64) * 8 floats
    const size_t tpitch = (vbo->w * 16 + 255) / 256 * 256;
    if (texels && (!texels->ptr || texels->w != vbo->w || texels->h != vbo->h || texels->pitch != tpitch || ((uintptr_t)texels->ptr & 15)))
        return set_error(KFX_E_SHAPE, "DepthToVbo+Normals: texel image");
    VboNrmParams p{(const unsigned char*)depth->ptr, depth->pitch, (unsigned char*)vbo->ptr, (unsigned char*)nrm->ptr, vbo->pitch, nrm->pitch,
                   (int)vbo->w, (int)vbo->h, Intr{K[0], K[1], K[2], K[3]}, scale, texels ? (unsigned char*)texels->ptr : nullptr, texels ? tpitch : 0,
                   texels ? reinterpret_cast<float*>((unsigned char*)texels->ptr + tpitch * vbo->h) : nullptr, (unsigned)((vbo->w + 63) / 64 * 8)};
    hipLaunchKernelGGL(k_vbo_normals_f32, dim3(ceil_div(p.w, 64), ceil_div(p.h, 4)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_depth_to_vbo_normals_f32");
}

// ---- the depth pyramid with its vertex and normal maps, one launch (round 5) -------------------------------------------------
// BoxReduceIgnoreInvalid (levels 1 .. L-1 from level 0) followed by DepthToVbo + NormalsFromVbo on every level are 2 L - 1
// launches of 3-5 us each (seven per frame of the tracked loop, profiles/r05_tracked: latency, not work).  Here a workgroup owns
// a 32 x 32 tile of level 0 and the 16 x 16 / 8 x 8 / 4 x 4 tiles below it: it stages the tile with an apron of 8 pixels to the
// right and below in LDS, halves it level by level there (k_box_half_ignore_invalid_f32's expression, same order of additions)
// -- the apron is what the next level's one-pixel apron is averaged from, and a level's one-pixel apron holds the neighbours its
// normals need -- writes each level's depth tile, and evaluates vertex_of / the normal's expression of k_vbo_normals_f32 from the
// LDS values.  Every output is the per-level launches' bit for bit.
struct PyrParams {
    const unsigned char* d0;
    size_t d0_pitch;
    unsigned char* d[4];      // depth of levels 1 .. (index 0 unused)
    size_t dpitch[4];
    unsigned char *vbo[4], *nrm[4];
    size_t vpitch[4], npitch[4];
    int w[4], h[4];
    Intr K[4];
    int levels;
    float scale;
};
__device__ __forceinline__ float4 pyr_vertex(const Intr& K, float scale, float depth, int u, int v)
{
    const float kz = scale * depth;
    return make_float4(kz * ((float)u - K.u0) / K.fu, kz * ((float)v - K.v0) / K.fv, kz, 1.0f);
}
__global__ __launch_bounds__(256) void k_depth_pyramid_vbo_normals(const PyrParams p)
{
    constexpr int E0 = 40, E1 = 20, E2 = 10, E3 = 5;   // staged extents: tile + apron
    __shared__ float s0[E0 * E0], s1[E1 * E1], s2[E2 * E2], s3[E3 * E3];
    const int tid = threadIdx.x;
    {   // level 0: the tile and its apron (zero beyond the image: such pixels only feed values nobody reads)
        const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
        for (int t = tid; t < E0 * E0; t += 256) {
            const int ly = t / E0, lx = t - ly * E0, x = x0 + lx, y = y0 + ly;
            s0[t] = (x < p.w[0] && y < p.h[0]) ? reinterpret_cast<const float*>(p.d0 + (size_t)y * p.d0_pitch)[x] : 0.f;
        }
    }
    __syncthreads();
    float* const sl[4] = {s0, s1, s2, s3};
    const int ext[4] = {E0, E1, E2, E3};
#pragma unroll
    for (int l = 1; l < 4; ++l) {
        if (l < p.levels) {   // (uniform)
            const int E = ext[l], Ef = ext[l - 1], S = 32 >> l;
            const int x0 = blockIdx.x * S, y0 = blockIdx.y * S;
            const float* f = sl[l - 1];
            for (int t = tid; t < E * E; t += 256) {
                const int ly = t / E, lx = t - ly * E;
                const float tx = f[(2 * ly) * Ef + 2 * lx], ty = f[(2 * ly) * Ef + 2 * lx + 1], bx = f[(2 * ly + 1) * Ef + 2 * lx], by = f[(2 * ly + 1) * Ef + 2 * lx + 1];
                int n = 0;
                float sum = 0;
                if (isfinite(tx)) { sum += tx; n++; }
                if (isfinite(ty)) { sum += ty; n++; }
                if (isfinite(bx)) { sum += bx; n++; }
                if (isfinite(by)) { sum += by; n++; }
                const float m = n > 0 ? (sum / n) : __builtin_nanf("");
                sl[l][t] = m;
                const int x = x0 + lx, y = y0 + ly;
                if (lx < S && ly < S && x < p.w[l] && y < p.h[l]) reinterpret_cast<float*>(p.d[l] + (size_t)y * p.dpitch[l])[x] = m;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        if (l >= p.levels) break;   // (uniform)
        const int E = ext[l], S = 32 >> l, w = p.w[l], h = p.h[l];
        const int x0 = blockIdx.x * S, y0 = blockIdx.y * S;
        const float* d = sl[l];
        const Intr K = p.K[l];
        for (int t = tid; t < S * S; t += 256) {
            const int ly = t / S, lx = t - ly * S, u = x0 + lx, v = y0 + ly;
            if (u >= w || v >= h) continue;
            const float4 Vc = pyr_vertex(K, p.scale, d[ly * E + lx], u, v);
            reinterpret_cast<float4*>(p.vbo[l] + (size_t)v * p.vpitch[l])[u] = Vc;
            float4 N = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u + 1 < w && v + 1 < h) {
                const float4 Vr = pyr_vertex(K, p.scale, d[ly * E + lx + 1], u + 1, v), Vu = pyr_vertex(K, p.scale, d[(ly + 1) * E + lx], u, v + 1);
                const V3 a = v3(Vr.x - Vc.x, Vr.y - Vc.y, Vr.z - Vc.z);
                const V3 b = v3(Vu.x - Vc.x, Vu.y - Vc.y, Vu.z - Vc.z);
                const V3 axb = v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
                const float mag = length(axb);
                N = make_float4(-axb.x / mag, -axb.y / mag, -axb.z / mag, 1.0f);
            }
            reinterpret_cast<float4*>(p.nrm[l] + (size_t)v * p.npitch[l])[u] = N;
        }
    }
}

// depth[0 .. levels): level 0 is the input, levels 1 .. are written (each at most half the size of the one before, as
// BoxHalfIgnoreInvalid requires); vbo[l] / nrm[l]: the maps of level l, of depth[l]'s size; K: levels x {fu, fv, u0, v0}
extern "C" int kfx_depth_pyramid_vbo_normals_f32(const kfx_image* depth, const kfx_image* vbo, const kfx_image* nrm, const float* K, int levels,
                                                 float scale, kfx_stream stream)
{
    if (!depth || !vbo || !nrm || !K) return set_error(KFX_E_NULL, "DepthPyramidVboNormals: null argument");
    if (levels < 1 || levels > 4) return set_error(KFX_E_RANGE, "DepthPyramidVboNormals: 1 to 4 levels");
    PyrParams p;
    p.levels = levels;
    p.scale = scale;
    for (int l = 0; l < 4; ++l) {
        p.d[l] = p.vbo[l] = p.nrm[l] = nullptr;
        p.dpitch[l] = p.vpitch[l] = p.npitch[l] = 0;
        p.w[l] = p.h[l] = 0;
        p.K[l] = Intr{1.f, 1.f, 0.f, 0.f};
    }
    for (int l = 0; l < levels; ++l) {
        if (int e = check_image(&depth[l], 4, "DepthPyramidVboNormals: depth image")) return e;
        if (int e = check_image(&vbo[l], 16, "DepthPyramidVboNormals: vbo image")) return e;
        if (int e = check_image(&nrm[l], 16, "DepthPyramidVboNormals: normal image")) return e;
        if (vbo[l].w != depth[l].w || vbo[l].h != depth[l].h || nrm[l].w != depth[l].w || nrm[l].h != depth[l].h)
            return set_error(KFX_E_SHAPE, "DepthPyramidVboNormals: the maps of a level differ in size from its depth image");
        if (l > 0 && (depth[l - 1].w < 2 * depth[l].w || depth[l - 1].h < 2 * depth[l].h))
            return set_error(KFX_E_SHAPE, "DepthPyramidVboNormals: a level larger than half the one before");
        if (depth[l].w == 0 || depth[l].h == 0) return set_error(KFX_E_SHAPE, "DepthPyramidVboNormals: empty level");
        p.d[l] = (unsigned char*)depth[l].ptr; p.dpitch[l] = depth[l].pitch;
        p.vbo[l] = (unsigned char*)vbo[l].ptr; p.vpitch[l] = vbo[l].pitch;
        p.nrm[l] = (unsigned char*)nrm[l].ptr; p.npitch[l] = nrm[l].pitch;
        p.w[l] = (int)depth[l].w; p.h[l] = (int)depth[l].h;
        p.K[l] = Intr{K[4 * l], K[4 * l + 1], K[4 * l + 2], K[4 * l + 3]};
    }
    p.d0 = (const unsigned char*)depth[0].ptr;
    p.d0_pitch = depth[0].pitch;
    hipLaunchKernelGGL(k_depth_pyramid_vbo_normals, dim3(ceil_div(p.w[0], 32), ceil_div(p.h[0], 32)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_depth_pyramid_vbo_normals_f32");
}

// TextureDepth<float4,uchar3>(img, kf, depth, norm, T_wd, Kdepth) (single keyframe: n_kf = 1, phong = NULL) and
// TextureDepth<float4,uchar3,10>(img, kfs, depth, norm, phong, T_wd, Kdepth) (cu_depth_tools.cu:123-207)
extern "C" int kfx_texture_depth(const kfx_image* img, const kfx_keyframe* kfs, int n_kf, const kfx_image* depth, const kfx_image* norm,
                                 const kfx_image* phong, const float T_wd[12], const float Kdepth[4], kfx_stream stream)
{
    if (int e = check_image(img, 16, "TextureDepth: output image")) return e;
    if (int e = check_image(depth, 4, "TextureDepth: depth image")) return e;
    if (int e = check_image(norm, 16, "TextureDepth: normal image")) return e;
    if (!kfs || n_kf < 1 || n_kf > TEX_MAX_KF || !T_wd || !Kdepth) return set_error(KFX_E_NULL, "TextureDepth: keyframes / transforms");
    const bool single = phong == nullptr;
    if (!single) { if (int e = check_image(phong, 4, "TextureDepth: phong image")) return e; }
    if (single && n_kf != 1) return set_error(KFX_E_RANGE, "TextureDepth: the single-keyframe form takes exactly one keyframe");
    if (img->w == 0 || img->h == 0) return 0;
    if (depth->w < img->w || depth->h < img->h || norm->w < img->w || norm->h < img->h || (!single && (phong->w < img->w || phong->h < img->h)))
        return set_error(KFX_E_SHAPE, "TextureDepth: inputs smaller than the output");
    TextureParams p;
    p.out = (unsigned char*)img->ptr; p.opitch = img->pitch;
    p.depth = (const unsigned char*)depth->ptr; p.dpitch = depth->pitch;
    p.norm = (const unsigned char*)norm->ptr; p.npitch = norm->pitch;
    p.phong = single ? nullptr : (const unsigned char*)phong->ptr; p.ppitch = single ? 0 : phong->pitch;
    p.w = (int)img->w; p.h = (int)img->h; p.nkf = n_kf; p.single = single ? 1 : 0;
    for (int i = 0; i < 12; ++i) p.T_wd.m[i] = T_wd[i];
    p.Kd = Intr{Kdepth[0], Kdepth[1], Kdepth[2], Kdepth[3]};
    for (int k = 0; k < TEX_MAX_KF; ++k) {
        KeyframeView& kv = p.kf[k];
        kv.img = nullptr; kv.pitch = 0; kv.w = kv.h = 0;
        kv.K = Intr{1.f, 1.f, 0.f, 0.f};
        for (int i = 0; i < 12; ++i) kv.T_iw.m[i] = 0.f;
        if (k < n_kf && kfs[k].img.ptr) {
            if (kfs[k].img.pitch < kfs[k].img.w * 3) return set_error(KFX_E_SHAPE, "TextureDepth: keyframe image pitch");
            kv.img = (const unsigned char*)kfs[k].img.ptr; kv.pitch = kfs[k].img.pitch; kv.w = (int)kfs[k].img.w; kv.h = (int)kfs[k].img.h;
            kv.K = Intr{kfs[k].K[0], kfs[k].K[1], kfs[k].K[2], kfs[k].K[3]};
            for (int i = 0; i < 12; ++i) kv.T_iw.m[i] = kfs[k].T_iw[i];
        }
    }
    if (single && !p.kf[0].img) return set_error(KFX_E_NULL, "TextureDepth: keyframe image is null");
    hipLaunchKernelGGL(k_texture_depth, dim3(ceil_div(p.w, 64), ceil_div(p.h, 4)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_texture_depth");
}
