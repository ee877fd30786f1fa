// kfx_device.h -- shared device-side helpers for the gfx950 kernels.
//
// Numerics contract ("exact" build): this translation unit is compiled with
// -ffp-contract=off and HIP's default correctly rounded fp32 divide/sqrt, and
// every expression keeps the reference's operand order and association, so the
// kernels are bit-comparable with the CPU oracle (oracle/kfx_oracle.c).
// Reference citations are paths inside the reference tree.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/kfx.h"
#include "../../include/kfx_extras.h"

namespace kfx {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
// float3 / float3 is a true division (CUDA_SDK/cutil_math.h:354-357)
__device__ __forceinline__ V3 div_cw(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
// float3 / float is multiply-by-reciprocal (cutil_math.h:358-362)
__device__ __forceinline__ V3 div_s(V3 a, float s) { const float inv = 1.0f / s; return a * inv; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float length(V3 a) { return sqrtf(dot(a, a)); }
// lerp = a + t*(b-a) (cutil_math.h:80-83, :375-378)
__device__ __forceinline__ float lerp(float a, float b, float t) { return a + t * (b - a); }
__device__ __forceinline__ V3 lerp(V3 a, V3 b, float t) { return a + (b - a) * t; }
// clamp = fmaxf(a, fminf(f, b)) (cutil_math.h:86-89)
__device__ __forceinline__ float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }

// a / b for a divisor that is uniform over the launch, with inv = 1.0f / b computed once on the host (IEEE, correctly
// rounded).  q0 = RN(a * inv) is within one ulp of the quotient, r = a - b * q0 is exact in an FMA, and RN(q0 + r * inv)
// is then the correctly rounded quotient (Markstein's correction step) -- the same float the IEEE division returns, as
// long as nothing over- or underflows: callers use it for |a|, |b| in [2^-40, 2^40] (div_uniform_safe) and take the
// hardware division otherwise.  Three instructions instead of the twelve of v_div_scale / v_rcp / fma... / v_div_fixup;
// checked exhaustively against the hardware division over all 2^32 numerators for a set of divisors
// (kfx_debug_div_uniform_check, tests/test_gpu_parity.py).
__device__ __forceinline__ float div_uniform(float a, float b, float inv)
{
    const float q0 = a * inv;
    const float r = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(r, inv, q0);
}
__device__ __forceinline__ bool div_uniform_safe(float a) { return fabsf(a) < 0x1p40f && fabsf(a) > 0x1p-40f; }
inline bool div_uniform_safe_host(float b) { const float m = b < 0 ? -b : b; return m < 0x1p40f && m > 0x1p-40f; }

// ---- IEEE quotients and square roots without the range-scaling wrapper ---------------------------------------------
// hipcc expands a correctly rounded fp32 a / b into v_div_scale x2, v_rcp, a Newton step on the reciprocal, a quotient
// with two residual corrections, v_div_fmas and v_div_fixup: eleven instructions, five of them of the slow class (measured
// on MI355X, scripts/ubench/valu_rate.hip: v_div_scale / v_div_fmas / v_div_fixup 5.1 cycles, v_rcp 8.8, v_fma 3.3 per
// wave-instruction).  v_div_scale only rescales operands whose exponents are extreme and v_div_fixup only patches
// zero / infinity / NaN operands; in between, the expansion is exactly rcp_nr + div_core below (multiplying by a power of
// two commutes with every rounding when nothing over- or underflows), so for |b| in [2^-40, 2^40] and a = 0 or |a| in
// [2^-60, 2^60] div_core(a, b, rcp_nr(b)) IS the IEEE quotient (up to the sign of a zero result) -- and the refined
// reciprocal can be shared by several numerators.  Checked against the hardware division on the GPU
// (kfx_debug_div_core_check, tests/test_gpu_parity.py).
__device__ __forceinline__ float rcp_nr(float b)
{
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float div_core(float a, float b, float y)
{
    const float q0 = a * y;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y, q1);
}
// Correctly rounded sqrtf for x in [2^-80, 2^80]: the reciprocal-square-root iteration LLVM itself uses for IEEE sqrt
// when denormals need no care (one v_rsq_f32 and seven multiply-adds; the denormal-safe expansion hipcc emits here is
// v_sqrt_f32 plus fifteen instructions, nine of them of the slow class).  Compared with sqrtf over every float of that
// range on the GPU (kfx_debug_sqrt_core_check).
__device__ __forceinline__ float sqrt_core(float x)
{
    const float r = __builtin_amdgcn_rsqf(x);
    float s = x * r;
    float h = r * 0.5f;
    const float e = __builtin_fmaf(-h, s, 0.5f);
    h = __builtin_fmaf(h, e, h);
    s = __builtin_fmaf(s, e, s);
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}

// A wave-uniform value the compiler would keep in a scalar register, moved to a vector register for good.  On gfx950 any
// VALU instruction with an SGPR source issues at the 5-cycle rate of the "slow" class instead of 3.3 cycles (measured:
// v_fma / v_mul / v_add / v_fmac with one SGPR operand 5.1 cycles per wave-instruction against 3.3-3.4 all-VGPR,
// scripts/ubench/valu_rate.hip), so the operands of an issue-bound inner loop belong in VGPRs even when they are uniform.
template <typename T>
__device__ __forceinline__ T in_vgpr(T x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// op(x[lane], x[lane ^ OFF]) for OFF in {1, 2, 16, 32} without the LDS crossbar of __shfl_xor (ds_bpermute): DPP quad
// permutes within 4 lanes; v_permlane16_swap / v_permlane32_swap (gfx950) exchange odd and even rows / the two halves of a
// wave -- called with both operands equal they return (even-side value, odd-side value) in every lane, which is all a
// commutative op needs.
template <int OFF, typename F>
__device__ __forceinline__ float wave_xor_combine(float x, F op)
{
    static_assert(OFF == 1 || OFF == 2 || OFF == 16 || OFF == 32, "lane distance");
    const unsigned u = __float_as_uint(x);
    if constexpr (OFF == 1 || OFF == 2) {
        const int o = __builtin_amdgcn_update_dpp(0, (int)u, OFF == 1 ? 0xB1 : 0x4E, 0xF, 0xF, true); // quad_perm [1,0,3,2] / [2,3,0,1]
        return op(x, __uint_as_float((unsigned)o));
    } else if constexpr (OFF == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        return op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        return op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
}


// op over the 8 lanes {8 k .. 8 k + 7} a lane belongs to, result in all of them: two quad permutes and DPP's row_half_mirror (lane i <->
// lane 7 - i of the half row, which pairs the two quads) -- three VALU instructions per value, no LDS
template <typename F>
__device__ __forceinline__ float wave8_combine(float x, F op)
{
    x = wave_xor_combine<1>(x, op);
    x = wave_xor_combine<2>(x, op);
    const int o = __builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), 0x141, 0xF, 0xF, true);   // row_half_mirror
    return op(x, __uint_as_float((unsigned)o));
}

// Row-major 3x4 pose, roo::Mat<float,3,4> (Mat.h:33-163)
struct Pose { float m[12]; };
// ImageIntrinsics {fu, fv, u0, v0} (ImageIntrinsics.h:51-200)
struct Intr { float fu, fv, u0, v0; };

// T * p (MatUtils.h:117-125)
__device__ __forceinline__ V3 se3_mul(const Pose& T, V3 p)
{
    return V3{T.m[0] * p.x + T.m[1] * p.y + T.m[2] * p.z + T.m[3],
              T.m[4] * p.x + T.m[5] * p.y + T.m[6] * p.z + T.m[7],
              T.m[8] * p.x + T.m[9] * p.y + T.m[10] * p.z + T.m[11]};
}
// mulSO3 (MatUtils.h:147-155)
__device__ __forceinline__ V3 so3_mul(const Pose& T, V3 r)
{
    return V3{T.m[0] * r.x + T.m[1] * r.y + T.m[2] * r.z,
              T.m[4] * r.x + T.m[5] * r.y + T.m[6] * r.z,
              T.m[8] * r.x + T.m[9] * r.y + T.m[10] * r.z};
}
// mulSO3inv (MatUtils.h:177-185)
__device__ __forceinline__ V3 so3_mul_inv(const Pose& T, V3 r)
{
    return V3{T.m[0] * r.x + T.m[4] * r.y + T.m[8] * r.z,
              T.m[1] * r.x + T.m[5] * r.y + T.m[9] * r.z,
              T.m[2] * r.x + T.m[6] * r.y + T.m[10] * r.z};
}

// Device view of a pitched image.
struct ImgView {
    const unsigned char* ptr;
    size_t pitch;
    int w, h;
};
template <typename T>
__device__ __forceinline__ const T* row(const ImgView& im, size_t y)
{
    return reinterpret_cast<const T*>(im.ptr + y * im.pitch);
}

// Device view of BoundedVolume<SDF_t>.
struct VolView {
    unsigned char* ptr;
    size_t pitch, img_pitch;
    int w, h, d;
    V3 bmin, bmax;
};

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

} // namespace kfx

// ---- brick summary of a TSDF volume (kfx_sdf_summary, include/kfx.h; summary.hip) -----------------------------------
// R: one float4 {lo, hi, state, -} per 8 x 8 x 8 cells, kept current by the TRACK fuse kernels (state 0: every cell holds
// a value in [lo, hi]; 1: every cell is NaN; 2: every cell is NaN or holds a value in [lo, hi] -- an invalidated brick has
// the infinite range).  C: what the ray-march reads (ClassView below), built from R on demand.
#define KFX_SUMMARY_RING 8
struct kfx_sdf_summary {
    float4* R;
    int nbx, nby, nbz;
    int w, h, d;                 // parent volume (cells)
    const unsigned char* base;   // parent volume storage
    size_t pitch, img_pitch;
    // class tables of the march (ClassView below): two bit planes per entry, fine level (8^3 or 16^3 cells) then 32^3 cells
    unsigned* C;                 // device, sized for the finest level
    int c_dirty;                 // R changed since C was built
    float c_tol, c_vref;         // what C was built with
    int c_shift;                 // fine level C was built for (log2 of its cells per entry)
    int n_coarse;                // 32^3-cell entries
    int* d_count;                // device: {running count of 32^3-cell entries of class != 0, workgroups that have added theirs}
    // The table builds publish their count of 32^3-cell entries of class != 0 in a host-visible ring (pinned, mapped): build b
    // writes slot b % KFX_SUMMARY_RING and records build_done[b % KFX_SUMMARY_RING] behind it.  The tracked raycast chooses its
    // kernel from the count of build b - 2 (raycast.hip, class_view): old enough to have finished without anybody waiting,
    // and -- unlike "whatever the last finished build said" -- the same choice for the same sequence of calls.
    int* h_skippable;            // the ring, host address (nullptr: no pinned memory; the tables are always used)
    int* d_skippable;            // the ring, device address
    hipEvent_t build_done[8];
    unsigned builds;             // table builds issued so far
    unsigned plain_calls;        // tracked raycasts since the choice last fell on the plain march (the tables are then rebuilt every 8th call only)
    unsigned sweeps;             // tracked SdfFuse launches so far: every other one walks the planes from the far end (fuse.hip)
};
namespace kfx {
// The class tables the march stages in LDS.  Per entry (a cube of 2^shift cells, together with the +1 cells a trilinear sample
// based in it reads) two bits: 0 = sample; 1 = every cell holds vref (exact numerics: that bit pattern; fast numerics: within
// tol) -- a sample there IS vref; 2 = every cell is NaN; 3 = every cell is NaN or holds vref -- the reference's step from
// there is trunc either way (a NaN sample steps by trunc, cu_raycast.cu:77-80, and vref = trunc >= min_delta), only
// last_sdf is undecided, and the march settles it with one sample if the next one is a crossing.  A row of entries along x
// is stored as uint2 {plane 0, plane 1} per 32 entries.
struct ClassLevel {
    int shift;        // log2(cells per entry along an axis)
    int ny;           // entries along y
    int rw;           // 32-bit words per row of entries: 2 * ceil(nx / 32)
    int first;        // first word of the level in the table
};
struct ClassView {
    const unsigned* C;   // global copy (fine level, then the 32^3 level)
    ClassLevel fine, coarse;
    int words;           // total 32-bit words
    float vref;          // the value of class 1 (= the launch's trunc_dist)
    float tol;           // relative tolerance the tables were built with (0: exact numerics)
    int amb_ok;          // class 3 may be skipped (trunc >= min_delta)
    int ox, oy, oz;      // cell offset of the view inside the parent volume
    float eps;           // margin (cells) that covers the error of the affine cell estimate
    // Coarser levels (64^3 and 128^3 cells) are not built in global memory: every raycast workgroup derives them in LDS from
    // the 32^3-cell level it has just staged (an entry = the combination of its 2 x 2 x 2 children, which include their +1
    // cells), behind the staged words.  nx5 / nz5: entries of the 32^3-cell level along x / z (ny5 = coarse.ny); top_n: how
    // many coarser levels there are (0-2); lds_words: staged words + the derived levels.
    int nx5, nz5, top_n, lds_words;
};
// geometry of a level derived from a finer one with nx x ny x nz entries, placed at word `first`
inline __host__ __device__ void class_level_up(const int nx, const int ny, const int nz, const int shift, const int first, ClassLevel& L, int& ux, int& uz, int& words)
{
    ux = (nx + 1) >> 1; uz = (nz + 1) >> 1;
    L.shift = shift; L.ny = (ny + 1) >> 1; L.rw = 2 * ((ux + 31) >> 5); L.first = first;
    words = (L.rw * L.ny * uz + 3) & ~3;
}
int summary_classes_prepare(kfx_sdf_summary* s, float tol, float vref, int fine_shift, hipStream_t stream);
void summary_class_layout(const kfx_sdf_summary* s, int fine_shift, ClassView& cv);
// cell offset of a view of the summary's parent volume (same pitches, pointer inside the parent): 0 on success
int summary_view_offset(const kfx_sdf_summary* s, const kfx_volume* view, int* ox, int* oy, int* oz);
} // namespace kfx

// ---- host-side launch helpers (capi.hip) --------------------------------------
namespace kfx {
int set_error(int code, const char* what);
int check_launch(const char* what);
int math_mode(); // KFX_MATH_EXACT / KFX_MATH_FAST
// kfx_frame_step's pair (frame.hip): the fused vbo / normals launch also writes the packed texel image {nx, ny, nz, depth} that the
// SdfFuse of the same frame stages by LDS-DMA (preprocess.hip, fuse.hip); texels may be null
size_t texel_image_bytes(size_t w, size_t h);   // the packed image of a w x h depth image: texel rows + block maxima (fuse.hip, tex_layout)
int depth_to_vbo_normals_texels(const kfx_image* vbo, const kfx_image* nrm, const kfx_image* depth, const float K[4], float scale,
                                const kfx_image* texels, kfx_stream stream);
int sdf_fuse_slab_texels(const kfx_volume* vol, const kfx_slab* slab, const kfx_image* depth, const kfx_image* norm, const kfx_image* texels,
                         const float T_cw[12], const float K[4], float trunc_dist, float max_w, float mincostheta, unsigned flags, kfx_stream stream);
int sdf_fuse_texels(const kfx_volume* vol, kfx_sdf_summary* summary, const kfx_image* depth, const kfx_image* norm, const kfx_image* texels,
                    const float T_cw[12], const float K[4], float trunc_dist, float max_w, float mincostheta, unsigned flags, kfx_stream stream);
} // namespace kfx
