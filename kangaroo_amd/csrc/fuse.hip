// fuse.hip -- per-voxel projective TSDF integration (roo::SdfFuse) and the volume
// initialisers (SdfReset, SdfSphere) for gfx950.
//
// Reference behaviour: src/cu_sdffusion.cu:16-61 (KernSdfFuse), :153-164 (SdfReset),
// :175-195 (KernSdfSphere).  New kernels, not a translation.  The reference launches (8,8,8)
// blocks whose warps straddle four rows and gathers depth/normals straight from global memory.
// Here:
//   * a lane owns two x-adjacent voxels = one 16-byte read-modify-write; a wave64 therefore
//     moves 512 B - 1 KiB contiguous per instruction, and marches FUSE_ZC slices re-using the
//     x/y part of the world->camera transform;
//   * the volume is touched only where the update predicate holds, like the reference
//     (cu_sdffusion.cu:44-49): HBM traffic is 16 B x updated voxels, streamed nontemporally;
//   * k_sdf_fuse_tiled stages the pixel rectangle a 64 x 8 x 16 voxel brick projects into
//     once in LDS, as {nx, ny, nz, depth} texels (LDS-DMA from a packed texel image), and serves all bilinear lookups from LDS.
//     Measured on MI355X (512^3, 640x480): with global gathers the vector L1 saturates
//     (288 M line accesses per launch, 0.69 ms regardless of arithmetic); tiled: 35 M accesses.
//   * two numerics modes (kfx_set_math_mode): exact = IEEE fp32 in the reference's operation
//     order, bit-identical to the CPU oracle, VALU-bound (five correctly rounded divisions and
//     a square root per voxel); fast = hardware rcp/rsq + FMA, the regime of the reference's
//     own -use_fast_math build, memory-bound.
// Packed v_pk_*_f32 arithmetic was tried for the exact mode and dropped: on gfx950 a packed
// op issues in 4 cycles, twice a scalar op, so it only saves issue slots (0.755 -> 0.726 ms).
#include <cmath>
#include <cstdlib>
#include <cstring>

#include <sys/syscall.h>
#include <unistd.h>

#include <hip/hip_fp16.h>

#include <type_traits>

#include "kfx_device.h"

// The tiled kernels stage their pixel rectangle by LDS-DMA from a packed texel image (round 6: k_pack_texels / the fused preprocess of
// kfx_frame_step write it; C3 / S_room SdfFuse 0.400 -> 0.361 ms, S_room 0.2737 -> 0.2645, S_full 0.3490 -> 0.3452, same bits:
// profiles/r06_stage_ab2).  A/B builds: -DKFX_FUSE_STAGE_DMA=0 compiles round 5's staging through registers (per texel: an index
// division, loads of the normal and the depth, a repack, a ds_write_b128) -- scripts/build_ab.sh, scripts/fuse_stage_ab.sh.
#ifndef KFX_FUSE_STAGE_DMA
#define KFX_FUSE_STAGE_DMA 1
#endif

namespace kfx {

constexpr int FUSE_ZC = 16;   // z-slices marched per workgroup
constexpr int FUSE_ROWS = 4;  // generic kernel: y-rows per workgroup (one per wave)
constexpr int TB_X = 64, TB_Y = 8; // tiled kernel: brick footprint (x: 32 lanes x 2 voxels, y: 4 waves x 2 half-waves)

struct FuseParams {
    unsigned char* vptr;
    size_t vpitch, vimg_pitch;
    int X, Y, Z;            // extents to integrate (reference: (dim/8)*8, quirk Q1)
    int zoff;               // global z index of local plane 0 (Z-slab of a larger volume, else 0)
    float w1, h1, d1;       // (float)(w-1), (float)(h-1), (float)(d-1)
    V3 bmin, size;          // bbox.Min(), bbox.Size()
    Pose T;                 // T_cw
    Intr K;
    ImgView depth;          // Image<float>
    ImgView norm;           // Image<float4>
    float dwb, dhb;         // (float)depth.w - 2, (float)depth.h - 2  (InBounds border, Image.h:287-291)
    float trunc, max_w, mincos;
    unsigned dpitch, npitch; // image pitches as 32-bit values (valid when `small_images`)
    int exact_shared;        // exact mode: camera / thresholds allow the shared-reciprocal arithmetic (see finish_shared)
    // serpentine sweep of the tracked launches (fuse_launch): z-bricks from the far end, and the planes [keep_z0, keep_z1) of
    // this launch read with ordinary loads
    int z_rev, keep_z0, keep_z1;
    // brick summary maintained by the TRACK kernels (kfx_sdf_summary, summary.hip): one float4 {lo, hi, state, -} per
    // 8 x 8 x 8 cells of the PARENT volume; (sum_bx0, sum_by0, sum_bz0) = brick index of this view's first cell
    float4* sum_R;
    int sum_nbx, sum_nby;
    int sum_bx0, sum_by0, sum_bz0;
    int sum_w, sum_h, sum_d; // parent volume dimensions in cells
    int zoff_local;          // first plane of this launch within the view (fuse_launch splits the view into z-ranges)
    int xcd_swizzle;         // tiled kernels: n > 0 rotates the x-brick of a workgroup by (z-brick >> (n - 1)) (see k_sdf_fuse_tiled)
    int fuse_cull;           // fast tiled kernels: the brick cull by the tile's own costheta bound (KFX_FUSE_CULL=0 switches it off)
    // packed texel image {nx, ny, nz, depth} (float4 per pixel of the depth image, rows `tpitch` bytes apart): what the tiled
    // kernels stage by LDS-DMA (k_pack_texels / the fused preprocess write it; null: the kernels gather depth and normals themselves)
    const unsigned char* tex;
    unsigned tpitch;
    // ... and, behind the texels, the maximum finite depth of every block of 8 x 4 pixels (-inf: none), rows of bw8 floats
    const float* bmax;
    unsigned bw8;
    // voxel positions (BoundedVolume.h:115-125): min + size * i / (float)(dim - 1) divides by a launch-uniform value, so the tiled
    // kernels take div_uniform (kfx_device.h: the host's correctly rounded reciprocal + one Markstein correction = the IEEE quotient,
    // three instructions instead of twelve) when sizes and extents are in its safe range (pos_div); else the hardware division
    float inv_w1, inv_h1;
    int pos_div;
};

// min + size * i / n1 with the reference's operations: the product, the division (IEEE quotient either way), the addition
__device__ __forceinline__ float voxel_pos(float bmin, float size, int i, float n1, float inv_n1, int pos_div)
{
    const float a = size * (float)i;
    return bmin + (pos_div ? div_uniform(a, n1, inv_n1) : a / n1);
}

struct Obs {
    float val, w;
    bool ok;
};

// {nx, ny, nz, depth} at the four corners of a bilinear cell: (ix,iy) (ix+1,iy) (ix,iy+1) (ix+1,iy+1)
struct Corners { float4 c00, c01, c10, c11; };

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lerp_f(float a, float b, float t) { return __builtin_fmaf(t, b - a, a); }

// ---- camera-frame position of a voxel --------------------------------------------------
// BoundedVolume::VoxelPositionInUnits (BoundedVolume.h:115-125) then T_cw * P_w
// (MatUtils.h:117-125).  The x/y terms are hoisted out of the z-march; in exact mode they are
// the leading partial sum T(i,0)*x + T(i,1)*y of the reference expression, so the association
// is unchanged.
template <bool FAST>
struct CamXY {
    float ax, ay, az;
    __device__ __forceinline__ void init(const FuseParams& p, float px, float py)
    {
        if constexpr (FAST) {
            ax = __builtin_fmaf(p.T.m[0], px, __builtin_fmaf(p.T.m[1], py, p.T.m[3]));
            ay = __builtin_fmaf(p.T.m[4], px, __builtin_fmaf(p.T.m[5], py, p.T.m[7]));
            az = __builtin_fmaf(p.T.m[8], px, __builtin_fmaf(p.T.m[9], py, p.T.m[11]));
        } else {
            ax = p.T.m[0] * px + p.T.m[1] * py;
            ay = p.T.m[4] * px + p.T.m[5] * py;
            az = p.T.m[8] * px + p.T.m[9] * py;
        }
    }
    __device__ __forceinline__ V3 at(const FuseParams& p, float pz) const
    {
        if constexpr (FAST)
            return v3(__builtin_fmaf(p.T.m[2], pz, ax), __builtin_fmaf(p.T.m[6], pz, ay), __builtin_fmaf(p.T.m[10], pz, az));
        else
            return v3(ax + p.T.m[2] * pz + p.T.m[3], ay + p.T.m[6] * pz + p.T.m[7], az + p.T.m[10] * pz + p.T.m[11]);
    }
};

// K.Project (ImageIntrinsics.h:87-91); fast mode keeps 1/Z for the weight
template <bool FAST>
__device__ __forceinline__ void project(const FuseParams& p, const V3 Pc, float& pu, float& pv, float& iz)
{
    if constexpr (FAST) {
        iz = __builtin_amdgcn_rcpf(Pc.z);
        pu = __builtin_fmaf(p.K.fu * Pc.x, iz, p.K.u0);
        pv = __builtin_fmaf(p.K.fv * Pc.y, iz, p.K.v0);
    } else {
        iz = 0.f;
        pu = p.K.u0 + p.K.fu * Pc.x / Pc.z;
        pv = p.K.v0 + p.K.fv * Pc.y / Pc.z;
    }
}

// depth.InBounds(p_c, 2) (Image.h:287-291)
__device__ __forceinline__ bool in_bounds(const FuseParams& p, float pu, float pv)
{
    return 2.0f <= pu && pu < p.dwb && 2.0f <= pv && pv < p.dhb;
}

// Global-memory corner fetch, any image size (64-bit addressing)
__device__ __forceinline__ Corners fetch_global64(const FuseParams& p, int ix, int iy)
{
    const float* dbl = row<float>(p.depth, (size_t)iy) + ix;
    const float* dtl = row<float>(p.depth, (size_t)iy + 1) + ix;
    const float4* nbl = row<float4>(p.norm, (size_t)iy) + ix;
    const float4* ntl = row<float4>(p.norm, (size_t)iy + 1) + ix;
    const float4 n00 = nbl[0], n01 = nbl[1], n10 = ntl[0], n11 = ntl[1];
    Corners c;
    c.c00 = make_float4(n00.x, n00.y, n00.z, dbl[0]);
    c.c01 = make_float4(n01.x, n01.y, n01.z, dbl[1]);
    c.c10 = make_float4(n10.x, n10.y, n10.z, dtl[0]);
    c.c11 = make_float4(n11.x, n11.y, n11.z, dtl[1]);
    return c;
}

// Global-memory corner fetch with a wave-uniform base (SGPR pair) + 32-bit lane offset: the
// global_load "saddr" form, no 64-bit address arithmetic per lane.
struct __attribute__((packed, aligned(4))) P2 { float a, b; };
__device__ __forceinline__ Corners fetch_global32(const FuseParams& p, int ix, int iy)
{
    const unsigned od = __umul24((unsigned)iy, p.dpitch) + (unsigned)ix * 4u;
    const unsigned on = __umul24((unsigned)iy, p.npitch) + (unsigned)ix * 16u;
    const P2 db = *reinterpret_cast<const P2*>(p.depth.ptr + od);
    const P2 dt = *reinterpret_cast<const P2*>(p.depth.ptr + p.dpitch + od);
    const float4* nb = reinterpret_cast<const float4*>(p.norm.ptr + on);
    const float4* nt = reinterpret_cast<const float4*>(p.norm.ptr + p.npitch + on);
    const float4 n00 = nb[0], n01 = nb[1], n10 = nt[0], n11 = nt[1];
    Corners c;
    c.c00 = make_float4(n00.x, n00.y, n00.z, db.a);
    c.c01 = make_float4(n01.x, n01.y, n01.z, db.b);
    c.c10 = make_float4(n10.x, n10.y, n10.z, dt.a);
    c.c11 = make_float4(n11.x, n11.y, n11.z, dt.b);
    return c;
}

// Bilinear lookup (Image::GetBilinear, Image.h:317-334), signed distance, weight and the
// update predicate (cu_sdffusion.cu:35-44).  No volume access.
template <bool FAST>
__device__ __forceinline__ Obs finish(const FuseParams& p, const V3 Pc, float iz, float fx, float fy, const Corners& c)
{
    Obs o;
    o.ok = false;
    o.val = 0.f;
    o.w = 0.f;
    float md, costheta, w;
    if constexpr (FAST) {
        md = lerp_f(lerp_f(c.c00.w, c.c01.w, fx), lerp_f(c.c10.w, c.c11.w, fx), fy);
        const float nx = lerp_f(lerp_f(c.c00.x, c.c01.x, fx), lerp_f(c.c10.x, c.c11.x, fx), fy);
        const float ny = lerp_f(lerp_f(c.c00.y, c.c01.y, fx), lerp_f(c.c10.y, c.c11.y, fx), fy);
        const float nz = lerp_f(lerp_f(c.c00.z, c.c01.z, fx), lerp_f(c.c10.z, c.c11.z, fx), fy);
        const float dotn = __builtin_fmaf(nz, Pc.z, __builtin_fmaf(ny, Pc.y, nx * Pc.x));
        const float len2 = __builtin_fmaf(Pc.z, Pc.z, __builtin_fmaf(Pc.y, Pc.y, Pc.x * Pc.x));
        costheta = -dotn * __builtin_amdgcn_rsqf(len2);
        w = costheta * iz;
    } else {
        md = lerp(lerp(c.c00.w, c.c01.w, fx), lerp(c.c10.w, c.c11.w, fx), fy);
        V3 mdn;
        mdn.x = lerp(lerp(c.c00.x, c.c01.x, fx), lerp(c.c10.x, c.c11.x, fx), fy);
        mdn.y = lerp(lerp(c.c00.y, c.c01.y, fx), lerp(c.c10.y, c.c11.y, fx), fy);
        mdn.z = lerp(lerp(c.c00.z, c.c01.z, fx), lerp(c.c10.z, c.c11.z, fx), fy);
        costheta = dot(mdn, Pc) / -length(Pc);
        w = costheta * 1.0f / Pc.z;
    }
    const float sd = costheta * (md - Pc.z);
    // unconditional: val / w are only read when ok is set, and selects are cheaper than divergent branches
    o.ok = ((int)!(sd <= -p.trunc) & (int)isfinite(md) & (int)isfinite(w) & (int)(costheta > p.mincos)) != 0;
    // fast numerics: one v_med3_f32 (sd is never NaN when the predicate holds; for a negative trunc_dist the median would
    // differ from the clamp, a case only the exact path reproduces)
    if constexpr (FAST) o.val = __builtin_amdgcn_fmed3f(sd, -p.trunc, p.trunc);
    else o.val = clampf(sd, -p.trunc, p.trunc);
    o.w = w;
    return o;
}

// Exact numerics with cheaper instruction sequences (k_sdf_fuse_tiled, bricks that pass the operand-range test):
// the same IEEE results as finish<false>, computed with
//   * one Newton-refined reciprocal of Z (`yz`, shared with the two projection quotients) for w = costheta / Z,
//   * sqrt_core for length(Pc) and rcp_nr + div_core for dot / -length (kfx_device.h),
//   * v_med3_f32 for the clamp (sd is never NaN when the predicate holds).
// The operand ranges that make div_core / sqrt_core exact are established per brick (Z, |X|, |Y| of all its voxels within
// [2^-20, 2^20]) and per launch (|fu|, |fv| in [2^-20, 2^20], mincostheta >= 2^-20, trunc > 0).  Numerators are not range
// checked: a quotient too small or too large for div_core to be exact (|dot| < 2^-100, results beyond 2^127) cannot pass
// the update predicate on either path -- costheta <= mincostheta, or a non-finite weight -- and a projection quotient
// below 2^-80 disappears in u0 + q (or leaves the sample outside the image border when u0 is that small too).
__device__ __forceinline__ Obs finish_shared(const FuseParams& p, const V3 Pc, float yz, float fx, float fy, const Corners& c)
{
    Obs o;
    const float md = lerp(lerp(c.c00.w, c.c01.w, fx), lerp(c.c10.w, c.c11.w, fx), fy);
    V3 mdn;
    mdn.x = lerp(lerp(c.c00.x, c.c01.x, fx), lerp(c.c10.x, c.c11.x, fx), fy);
    mdn.y = lerp(lerp(c.c00.y, c.c01.y, fx), lerp(c.c10.y, c.c11.y, fx), fy);
    mdn.z = lerp(lerp(c.c00.z, c.c01.z, fx), lerp(c.c10.z, c.c11.z, fx), fy);
    const float nlen = -sqrt_core(dot(Pc, Pc));
    const float costheta = div_core(dot(mdn, Pc), nlen, rcp_nr(nlen));
    const float w = div_core(costheta, Pc.z, yz); // costheta * 1.0f / Pc.z
    const float sd = costheta * (md - Pc.z);
    o.ok = ((int)!(sd <= -p.trunc) & (int)isfinite(md) & (int)isfinite(w) & (int)(costheta > p.mincos)) != 0;
    o.val = __builtin_amdgcn_fmed3f(sd, -p.trunc, p.trunc);
    o.w = w;
    return o;
}

// finish_shared with the x-differences of the bilinear cell taken from the tile (DXT kernels): d0 = c01 - c00 and d1 = c11 - c10
// are the same single roundings whoever computes them, and every voxel that samples the cell uses them, so they are formed
// once per texel while staging; lerp(a, b, fx) = a + fx * (b - a) becomes a + fx * d: two operations instead of three,
// eight fewer per voxel, same bits.
__device__ __forceinline__ Obs finish_shared_dx(const FuseParams& p, const V3 Pc, float yz, float fx, float fy, const float4 c00, const float4 d0,
                                                const float4 c10, const float4 d1)
{
    Obs o;
    const float md = lerp(c00.w + fx * d0.w, c10.w + fx * d1.w, fy);
    V3 mdn;
    mdn.x = lerp(c00.x + fx * d0.x, c10.x + fx * d1.x, fy);
    mdn.y = lerp(c00.y + fx * d0.y, c10.y + fx * d1.y, fy);
    mdn.z = lerp(c00.z + fx * d0.z, c10.z + fx * d1.z, fy);
    const float nlen = -sqrt_core(dot(Pc, Pc));
    const float costheta = div_core(dot(mdn, Pc), nlen, rcp_nr(nlen));
    const float w = div_core(costheta, Pc.z, yz); // costheta * 1.0f / Pc.z
    const float sd = costheta * (md - Pc.z);
    o.ok = ((int)!(sd <= -p.trunc) & (int)isfinite(md) & (int)isfinite(w) & (int)(costheta > p.mincos)) != 0;
    o.val = __builtin_amdgcn_fmed3f(sd, -p.trunc, p.trunc);
    o.w = w;
    return o;
}

__device__ __forceinline__ int med3_i32(int x, int lo, int hi) // clamp for lo <= hi
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
    return r;
}

// ---- cell storage policies ----------------------------------------------------------------
// CellF32: roo::SDF_t {float val; float w;} (Sdf.h:11-36).  CellF16: roo::SDF_h {half val; half w;},
// the 4-byte cell of BASELINE config C5 -- the arithmetic of the reference's commented-out half
// variant (Sdf.h:38-62): every intermediate of the running average is rounded to half
// (round-to-nearest-even) before the next operation.  ld2/st2 move two x-adjacent cells as
// (val0, w0, val1, w1); the volume is streamed nontemporally (each cell is touched once per frame;
// measured: in-place RMW sweep of 512^3 5.6 -> 6.0 TB/s).
struct CellF32 {
    static constexpr int BYTES = 8;
    __device__ static __forceinline__ float q(float x) { return x; }
    __device__ static __forceinline__ float4 ld2(const unsigned char* p)
    {
        const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
        return make_float4(t.x, t.y, t.z, t.w);
    }
    __device__ static __forceinline__ void st2(unsigned char* p, const float4 c)
    {
        v4f t;
        t.x = c.x; t.y = c.y; t.z = c.z; t.w = c.w;
        __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
    }
    __device__ static __forceinline__ float2 ld1(const unsigned char* p) { return *reinterpret_cast<const float2*>(p); }
    __device__ static __forceinline__ void st1(unsigned char* p, const float2 c) { *reinterpret_cast<float2*>(p) = c; }
};

typedef unsigned int v2u __attribute__((ext_vector_type(2)));
struct CellF16 {
    static constexpr int BYTES = 4;
    __device__ static __forceinline__ float q(float x) { return __half2float(__float2half_rn(x)); }
    __device__ static __forceinline__ float2 unpack(unsigned u)
    {
        return make_float2(__half2float(__ushort_as_half((unsigned short)(u & 0xffffu))), __half2float(__ushort_as_half((unsigned short)(u >> 16))));
    }
    __device__ static __forceinline__ unsigned pack(float val, float w)
    {
        return (unsigned)__half_as_ushort(__float2half_rn(val)) | ((unsigned)__half_as_ushort(__float2half_rn(w)) << 16);
    }
    __device__ static __forceinline__ float4 ld2(const unsigned char* p)
    {
        const v2u t = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(p));
        const float2 a = unpack(t.x), b = unpack(t.y);
        return make_float4(a.x, a.y, b.x, b.y);
    }
    __device__ static __forceinline__ void st2(unsigned char* p, const float4 c)
    {
        v2u t;
        t.x = pack(c.x, c.y);
        t.y = pack(c.z, c.w);
        __builtin_nontemporal_store(t, reinterpret_cast<v2u*>(p));
    }
    __device__ static __forceinline__ float2 ld1(const unsigned char* p) { return unpack(*reinterpret_cast<const unsigned*>(p)); }
    __device__ static __forceinline__ void st1(unsigned char* p, const float2 c) { *reinterpret_cast<unsigned*>(p) = pack(c.x, c.y); }
};

// SDF_t::operator+= then LimitWeight (Sdf.h:22-32): `o` is the new sample, (oval, ow) the stored cell.
// CELL::q rounds to the storage precision after every operation (identity for fp32 cells).
template <bool FAST, typename CELL>
__device__ __forceinline__ void accumulate(const Obs& o, float max_w, float& oval, float& ow)
{
    float val = CELL::q(o.val), w = CELL::q(o.w);
    if (ow > 0) {
        if constexpr (FAST) {
            // the running average in incremental form, old + (new - old) w / (w + ow): the same value up to rounding, and a
            // cell that is handed the value it already holds keeps it bit for bit.  (With (w val + ow oval) * rcp(w + ow) the
            // +trunc of free space crept away from trunc by the reciprocal's bias, a few 1e-8 per frame: after ~800 frames
            // no free region passed the class tables' 1e-5 test any more and the table march had nothing left to skip.)
            // (fp32 cells; the half cells keep the reference's operation order, whose every intermediate is rounded to half)
            const float ws = CELL::q(w + ow);
            if constexpr (CELL::BYTES == 8) val = __builtin_fmaf(w * __builtin_amdgcn_rcpf(ws), val - oval, oval);
            else val = CELL::q(CELL::q(__builtin_fmaf(w, val, ow * oval)) * __builtin_amdgcn_rcpf(ws));
            w = ws;
        } else {
            val = CELL::q(w * val + ow * oval);
            w = CELL::q(w + ow);
            val = CELL::q(val / w);
        }
    }
    oval = val;
    ow = CELL::q(fminf(w, max_w));
}

// the four corners from the packed texel image (k_pack_texels: copies of the same normals and depths), one 16-byte load each
__device__ __forceinline__ Corners fetch_texels(const FuseParams& p, int ix, int iy)
{
    const unsigned o = __umul24((unsigned)iy, p.tpitch) + (unsigned)ix * 16u;
    const float4* b = reinterpret_cast<const float4*>(p.tex + o);
    const float4* t = reinterpret_cast<const float4*>(p.tex + p.tpitch + o);
    Corners c;
    c.c00 = b[0]; c.c01 = b[1]; c.c10 = t[0]; c.c11 = t[1];
    return c;
}

// One voxel: projection -> bounds test -> corner fetch -> observation.
template <bool FAST, bool OFF32, bool TEX = false>
__device__ __forceinline__ Obs observe(const FuseParams& p, const V3 Pc)
{
    Obs o;
    o.ok = false;
    o.val = 0.f;
    o.w = 0.f;
    float pu, pv, iz;
    project<FAST>(p, Pc, pu, pv, iz);
    if (in_bounds(p, pu, pv)) {
        const float fix = floorf(pu), fiy = floorf(pv); // quirk Q6: floorf, then integer conversion
        const Corners c = TEX ? fetch_texels(p, (int)fix, (int)fiy) : (OFF32 ? fetch_global32(p, (int)fix, (int)fiy) : fetch_global64(p, (int)fix, (int)fiy));
        o = finish<FAST>(p, Pc, iz, pu - fix, pv - fiy, c);
    }
    return o;
}

// ---------------------------------------------------------------------------------------
// Generic kernel: global gathers, any alignment (VEC = 1: 8-byte cells for sub-volume views
// with odd x offset / pitch), any image size.  Workgroup = (64*VEC) x 4 x FUSE_ZC voxels.
// ---------------------------------------------------------------------------------------
template <int VEC, bool FAST, bool OFF32, typename CELL>
__global__ __launch_bounds__(256) void k_sdf_fuse(const FuseParams p)
{
    __shared__ float s_pz[FUSE_ZC];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x0 = (blockIdx.x * 64 + lane) * VEC;
    const int y = blockIdx.y * FUSE_ROWS + wv;
    const int zbeg = blockIdx.z * FUSE_ZC;
    const int zend = min(zbeg + FUSE_ZC, p.Z);
    if (threadIdx.x < FUSE_ZC) // VoxelPositionInUnits z, once per slice
        s_pz[threadIdx.x] = p.bmin.z + p.size.z * (float)(zbeg + (int)threadIdx.x + p.zoff) / p.d1;
    __syncthreads();
    if (x0 >= p.X || y >= p.Y) return;

    const float py = p.bmin.y + p.size.y * (float)y / p.h1;
    CamXY<FAST> cam[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) cam[v].init(p, p.bmin.x + p.size.x * (float)(x0 + v) / p.w1, py);

    unsigned char* cell = p.vptr + (size_t)zbeg * p.vimg_pitch + (size_t)y * p.vpitch + (size_t)x0 * CELL::BYTES;
    for (int z = zbeg; z < zend; ++z, cell += p.vimg_pitch) {
        const float pz = s_pz[z - zbeg];
        Obs o[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) o[v] = observe<FAST, OFF32>(p, cam[v].at(p, pz));
        if constexpr (VEC == 2) {
            if (o[0].ok || o[1].ok) {
                float4 c = CELL::ld2(cell);
                if (o[0].ok) accumulate<FAST, CELL>(o[0], p.max_w, c.x, c.y);
                if (o[1].ok) accumulate<FAST, CELL>(o[1], p.max_w, c.z, c.w);
                CELL::st2(cell, c);
            }
        } else {
            if (o[0].ok) {
                float2 c = CELL::ld1(cell);
                accumulate<FAST, CELL>(o[0], p.max_w, c.x, c.y);
                CELL::st1(cell, c);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// LDS-tiled kernel.  A workgroup owns a brick of 64 x 8 x FUSE_ZC voxels.  The perspective
// image of a z-column of voxels is a line segment, so the projections of the brick's first and
// last slice bound the pixel rectangle all of its voxels sample (given Z > 0 throughout, which
// the two end slices establish because Z is affine in z).  That rectangle, plus one texel of
// slack, is staged once in LDS as {nx, ny, nz, depth}; every bilinear lookup is then four
// ds_read_b128.  Bricks whose rectangle misses the in-bounds band of the image are skipped;
// bricks too close to the camera for the rectangle to fit `cap_px` texels, or straddling the
// camera plane, take the global-gather path (per lane, so a texel outside the staged
// rectangle can never be read from LDS).  Texel values are copies, so results are identical
// to the generic kernel in either numerics mode.
// ---------------------------------------------------------------------------------------
// Brick geometry (template): LX lanes side by side in x (2 voxels each), 64 / LX rows per wave, WY waves stacked in y,
// the remaining 4 / WY waves split the ZC slices between them.  <32, 4, 16>: 64 x 8 x 16 voxels, the default; <16, 2, 16>:
// 32 x 8 x 16 voxels (a wave = 32 voxels x 4 rows, two waves per half of the slices) for ranges where a voxel covers more
// than ~1.3 pixels -- the rectangle narrows with the brick (its width is about r (BX + 0.56 BZ), its height r (BY + 0.42 BZ)
// texels for r pixels per voxel), so it keeps fitting a tile that leaves 3-6 workgroups on a CU where the wide brick would
// need 48 KiB or fall back to global gathers.
// TRACK: besides the update, the kernel keeps the brick summary of the volume current (include/kfx.h, kfx_sdf_summary):
// per 8 x 8 x 8 cells the range of the values written this frame and whether every cell was rewritten, folded into the
// stored range by the one workgroup that owns the summary brick (64 x 8 x 16 and 32 x 8 x 16 voxel bricks are unions of
// whole summary bricks, so no atomics: DPP / permlane swaps reduce a wave's lanes, LDS the workgroup's waves).  RaycastSdf
// uses it to step through uniformly free or never-observed space without touching the volume (raycast.hip).
// Register budget (launch bounds): fast, ZU = 2: the 64 VGPRs of 8 waves per SIMD -- what the 1216-texel tile makes room
// for; without the bound hipcc's allocation moves between 64 and 78 on unrelated edits.  Bit-exact: the 80 of 6 waves; one
// wave less costs 6 % (measured when an edit took the untracked kernel from 78 to 81; the tracked one took 86 unbounded and
// fits 77 without scratch).  scripts/check_fuse_codegen.py checks both.
// DXT (bit-exact kernels): a texel is staged as {texel, difference to its right neighbour} -- 32 bytes, so `cap_px` texels take
// twice the LDS; chosen by the host for ranges whose rectangles fit half the tile (finish_shared_dx).
// NW: waves per workgroup (4; 8 for the narrow brick at large tiles -- twice the waves behind one staged rectangle where LDS,
// not registers, limits how many workgroups a CU holds: k_sdf_fuse_tiled<true, 2, CELL, 16, 2, 16, false, false, 8>).
template <bool FAST, int ZU, typename CELL, int LX = 32, int WY = 4, int ZC = FUSE_ZC, bool TRACK = false, bool DXT = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, (FAST && ZU == 2) ? 8 : (!FAST ? 6 : 1)) void k_sdf_fuse_tiled(const FuseParams p_arg, const int cap_px)
{
    static_assert(!DXT || (!FAST && !TRACK), "the difference tile belongs to the bit-exact, untracked kernels");
    // the uniforms of the per-voxel arithmetic live in vector registers (in_vgpr, kfx_device.h): an SGPR operand makes a
    // 3.3-cycle instruction a 5-cycle one
    FuseParams p = p_arg;
    // (not in the fast TRACK instantiation: its bookkeeping registers on top of the parked uniforms would cost the eighth wave
    // per SIMD -- 72 VGPRs -- which is worth more than the issue slots)
    constexpr bool PARK = !(FAST && TRACK);
    if constexpr (PARK) {
        p.T.m[2] = in_vgpr(p.T.m[2]); p.T.m[6] = in_vgpr(p.T.m[6]); p.T.m[10] = in_vgpr(p.T.m[10]);
        if constexpr (!FAST) { p.T.m[3] = in_vgpr(p.T.m[3]); p.T.m[7] = in_vgpr(p.T.m[7]); p.T.m[11] = in_vgpr(p.T.m[11]); }
        p.K.fu = in_vgpr(p.K.fu); p.K.fv = in_vgpr(p.K.fv); p.K.u0 = in_vgpr(p.K.u0); p.K.v0 = in_vgpr(p.K.v0);
    }
    constexpr int NT = 64 * NW, RW = 64 / LX, WZ = NW / WY, ZW = ZC / WZ, BY = RW * WY;
    static_assert(LX * RW == 64 && WY * WZ == NW && ZW * WZ == ZC && ZW % ZU == 0, "brick geometry");
    static_assert(NW == 4 || (NW == 8 && !TRACK), "the summary epilogue is written for four waves");
    constexpr int NG = ZW / 8, NXB = LX / 4, NZB = ZC / 8; // summary bricks: z-groups per wave, per workgroup along x and z
    static_assert(!TRACK || (BY == 8 && ZW % 8 == 0 && NG <= 2 && 8 % ZU == 0), "summary bricks are 8 x 8 x 8");
    __shared__ float s_part[TRACK ? 4 * 2 * 8 * 3 : 1];
    extern __shared__ __attribute__((aligned(16))) float4 s_tile[];
    __shared__ float s_pz[ZC];
    __shared__ float s_dmax[NW], s_cmin[NW];
    __shared__ float4 s_tz[ZC]; // exact mode: {pz, T(0,2)*pz, T(1,2)*pz, T(2,2)*pz} per slice
    // (the wave index through readfirstlane: the compiler cannot tell that tid >> 6 is wave-uniform, and without it the slice
    // loop's bounds, the loop branch and everything derived from them are computed per lane)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups go to the 8 XCDs round-robin by linear id, and the grid is 8 bricks wide at 512 voxels: blockIdx.x alone
    // would pin an x-slab of the volume to an XCD, and a scene that leaves the outer slabs half empty (the walls of
    // S_room) leaves two XCDs idle while six finish.  Rotating the x-brick by the z-brick gives every XCD every x-slab;
    // an XCD still works on one x-slab at a time (all y-bricks of a layer), so the image columns its bricks stage stay
    // in its L2 (rotating by the y-brick as well made S_full 12 % slower: every XCD then cycles through the whole image).
    // Measured, interleaved on one box (scripts/ab_xcd_swizzle.sh): S_room fast 0.438 -> 0.410 ms, S_full unchanged.
    const int bzi = (TRACK && p.z_rev) ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z;
    const int bxi = p.xcd_swizzle ? (int)((blockIdx.x + (bzi >> (p.xcd_swizzle - 1))) % gridDim.x) : (int)blockIdx.x;
    const int x0 = (bxi * LX + (lane & (LX - 1))) * 2;
    const int y = blockIdx.y * BY + (wv % WY) * RW + lane / LX;
    const int zbeg = bzi * ZC;
    const bool keep = TRACK && zbeg >= p.keep_z0 && zbeg < p.keep_z1;
    const int zend = min(zbeg + ZC, p.Z);
    const int wz0 = zbeg + (wv / WY) * ZW, wz1 = min(wz0 + ZW, zend); // this wave's slices
    const bool live = x0 < p.X && y < p.Y;
    if (tid < ZC) {
        const float pz = p.bmin.z + p.size.z * (float)(zbeg + tid + p.zoff) / p.d1;
        s_pz[tid] = pz;
        if constexpr (!FAST) s_tz[tid] = make_float4(pz, p.T.m[2] * pz, p.T.m[6] * pz, p.T.m[10] * pz);
    }
    __syncthreads();

    const float py = voxel_pos(p.bmin.y, p.size.y, y, p.h1, p.inv_h1, p.pos_div);
    CamXY<FAST> cam[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) cam[v].init(p, voxel_pos(p.bmin.x, p.size.x, x0 + v, p.w1, p.inv_w1, p.pos_div), py);

    // ---- pixel rectangle of the brick: the projections of its eight corners ----
    // A pinhole camera maps a convex body in front of it onto the convex hull of its vertices' images, so the corners of the brick --
    // of its live part: the voxels inside the extents -- bound the pixel of every sample; Z and |X|, |Y| are affine / convex in the
    // voxel index, so their extremes sit in the corners too.  Every wave evaluates the eight corners itself (lane & 7 = corner, the
    // voxel's own expressions) and reduces over eight lanes with DPP: no LDS round trip, no barrier, ~50 vector instructions where
    // projecting every lane's four end-slice voxels and reducing over the workgroup took ~170 (round 6; the rectangle decides which
    // texels are staged and which bricks are skipped, never a value: a sample that rounding puts a hair outside the corners' hull
    // falls into the one-texel slack).
    float umin, umax, vmin, vmax, zmin, cmax = 0.f;
    bool any_bad;
    {
        const int c = lane & 7;
        const int xb0 = bxi * LX * 2, yb0 = (int)blockIdx.y * BY;
        const int xc = (c & 1) ? min(xb0 + LX * 2, p.X) - 1 : xb0;
        const int yc = (c & 2) ? min(yb0 + BY, p.Y) - 1 : yb0;
        const float pzc = s_pz[(c & 4) ? (zend - 1 - zbeg) : 0];
        CamXY<FAST> cc;
        cc.init(p, voxel_pos(p.bmin.x, p.size.x, xc, p.w1, p.inv_w1, p.pos_div), voxel_pos(p.bmin.y, p.size.y, yc, p.h1, p.inv_h1, p.pos_div));
        const V3 Pc = cc.at(p, pzc);
        float pu, pv, iz;
        project<FAST>(p, Pc, pu, pv, iz);
        const bool bad = !(Pc.z > 0.f) || !(fabsf(pu) < 1e9f) || !(fabsf(pv) < 1e9f);
        const auto fmin2 = [](float a, float b) { return fminf(a, b); };
        const auto fmax2 = [](float a, float b) { return fmaxf(a, b); };
        // (every lane holds the same values now; readfirstlane tells the compiler so -- what is derived from them below, the tile's
        //  origin and width in the voxel loop's addresses included, belongs in scalar registers)
        const auto uni = [](float x) { return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(x))); };
        umin = uni(wave8_combine(pu, fmin2)); umax = uni(wave8_combine(pu, fmax2));
        vmin = uni(wave8_combine(pv, fmin2)); vmax = uni(wave8_combine(pv, fmax2));
        zmin = uni(wave8_combine(Pc.z, fmin2));
        if constexpr (!FAST) cmax = uni(wave8_combine(fmaxf(fmaxf(fabsf(Pc.x), fabsf(Pc.y)), Pc.z), fmax2));
        any_bad = __ballot(bad) != 0ull;
    }

    bool use_tile = false;
    int tx0 = 0, ty0 = 0, tw = 0, th = 0;
    if (!any_bad) {
        // every sample needs 2 <= pu < w-2 and 2 <= pv < h-2: a rectangle that misses that band
        // means no voxel of the brick can pass InBounds (workgroup-uniform exit)
        if (umax < 2.0f || !(umin < p.dwb) || vmax < 2.0f || !(vmin < p.dhb)) return;
        // bilinear cells (ix, ix+1) x (iy, iy+1) of every sample, one texel of slack per side
        const float fx0 = fmaxf(floorf(umin) - 1.f, 0.f), fx1 = fminf(floorf(umax) + 2.f, (float)(p.depth.w - 1));
        const float fy0 = fmaxf(floorf(vmin) - 1.f, 0.f), fy1 = fminf(floorf(vmax) + 2.f, (float)(p.depth.h - 1));
        tx0 = (int)fx0; ty0 = (int)fy0;
        tw = (int)fx1 - tx0 + 1; th = (int)fy1 - ty0 + 1;
        use_tile = tw > 1 && th > 1 && tw * th <= cap_px;
        // DXT: only the shared-reciprocal loop reads the {texel, difference} layout; a brick that fails its operand-range test
        // (or KFX_FUSE_EXACT_SHARED=0) gathers from global memory
        if constexpr (DXT) use_tile = use_tile && p.exact_shared && zmin >= 0x1p-20f && cmax <= 0x1p20f;
    }
    // Interior bricks: the rectangle of projections lies inside the band 2 <= pu < w - 2, 2 <= pv < h - 2 by a margin far
    // above the rounding of a projection (a voxel of an inner slice projects between the projections of its column's two
    // ends), so every sample passes InBounds and falls inside the staged rectangle with its one-texel slack: the voxel loop
    // of such a brick needs neither the four bounds compares, nor the two rectangle compares, nor the two index clamps --
    // eight slow-class instructions and as many scalar ones per voxel (fast: 139 + 40 -> 123 + 19 vector + scalar
    // instructions per voxel pair, exact: 240 + 35 -> 219 + 15).  Measured: fast 0.3310 -> 0.3293 ms, exact 0.4425 -> 0.436 ms
    // at 512^3 -- far less than the instruction count: by now neither loop is bound by issue alone.
    // (TRACK keeps the lanes beyond the volume's extents in the march, and those project anywhere: a brick cut by the extents
    // takes the checked loop)
    const bool whole = !TRACK || ((bxi + 1) * LX * 2 <= p.X && ((int)blockIdx.y + 1) * BY <= p.Y);
    const bool interior = __builtin_amdgcn_readfirstlane((int)(use_tile && whole && umin >= 2.01f && umax < p.dwb - 0.01f && vmin >= 2.01f && vmax < p.dhb - 0.01f)) != 0;
    float dmax = -__builtin_inff(); // farthest finite depth in the rectangle (fmaxf skips NaN texels)
    // Staging by LDS-DMA (gfx950 global_load_lds_dwordx4: 16 bytes per lane straight to LDS at M0 + lane * 16; no VGPR destination, no
    // ds_write) from the PACKED texel image {nx, ny, nz, depth} (p.tex).  The rectangle's rows are shared out over the waves; a wave
    // stages a row in chunks of 64 texels that land as s_tile[r * tw + c], the layout the loops below read.  Everything per chunk is
    // SCALAR: the row's address is an SGPR pair (the instruction's saddr; the lane's share is the one VGPR lane * 16), the lanes beyond
    // the row's end are switched off through exec (s_bfm: the mask of the tail chunk), the destination goes to M0 -- no vector
    // instruction at all, where the register path below spends an index division, two address computations, a 16-byte and a 4-byte
    // load, a repack, a ds_write_b128 and a maximum PER TEXEL (a first version that left addresses and predicates to hipcc kept the
    // kernel's vector instruction count where it was: profiles/r06_c3).  dmax -- the farthest depth of the rectangle, for the occlusion
    // cull -- comes from the maxima of 8 x 4 pixel blocks the packing launch leaves beside the texels (p.bmax): one look-up per thread
    // over the blocks the rectangle touches, a superset of it, so the bound is conservative (it decides which bricks are skipped,
    // never a value).  Texels are the same copies of the same pixels: same bits.
    constexpr bool DMA = KFX_FUSE_STAGE_DMA && !DXT;
    if constexpr (DMA) {
        if (use_tile) {   // (uniform)
            const unsigned lane16 = (unsigned)lane * 16u;
            const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(char*)s_tile;
            const int tail = tw & 63;                                            // texels of a row's last chunk (0: it is full)
            const unsigned long long mask_tail = tail ? ((1ull << tail) - 1ull) : ~0ull;
            const int chunks = (tw + 63) >> 6;
            for (int r = wv; r < th; r += NW) {   // (scalar loop control: wv went through readfirstlane)
                const unsigned char* rowp = p.tex + ((size_t)(ty0 + r) * p.tpitch + (size_t)tx0 * 16u);
                unsigned dst = lds0 + (unsigned)(r * tw) * 16u;
#pragma unroll 1
                for (int c = 0; c < chunks; ++c, rowp += 1024, dst += 1024u) {
                    const unsigned long long m = (c + 1 == chunks) ? mask_tail : ~0ull;
                    unsigned long long keep_exec;
                    unsigned keep_m0;
                    asm volatile("s_mov_b64 %0, exec\n\t"
                                 "s_mov_b32 %1, m0\n\t"
                                 "s_mov_b64 exec, %2\n\t"
                                 "s_mov_b32 m0, %3\n\t"
                                 "s_nop 0\n\t"
                                 "global_load_lds_dwordx4 %4, %5\n\t"
                                 "s_mov_b32 m0, %1\n\t"
                                 "s_mov_b64 exec, %0"
                                 : "=&s"(keep_exec), "=&s"(keep_m0)
                                 : "s"(m), "s"(dst), "v"(lane16), "s"(rowp)
                                 : "memory");
                }
            }
            // the farthest depth: maxima of the 8 x 4 pixel blocks the rectangle touches (NaN-free: a block without a finite depth holds -inf)
            {
                const int bx0 = tx0 >> 3, nbx = ((tx0 + tw - 1) >> 3) - bx0 + 1, by0 = ty0 >> 2, nb = nbx * (((ty0 + th - 1) >> 2) - by0 + 1);
                const float inv_nbx = 1.0f / (float)nbx;
                for (int i = tid; i < nb; i += NT) {
                    const int br = (int)(((float)i + 0.5f) * inv_nbx);   // i / nbx (i < 4096)
                    dmax = fmaxf(dmax, p.bmax[(size_t)(by0 + br) * p.bw8 + (size_t)(bx0 + i - br * nbx)]);
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off, 64));
                if (lane == 0) s_dmax[wv] = dmax;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMAs have landed; the barrier below covers the other waves'
        }
    }
    if (!DMA && use_tile) {
        // Cooperative staging, flat over the rectangle's texels (consecutive threads = consecutive texels of a row), four
        // texels per thread requested before the first is consumed: a row-by-row loop waits one L2 round trip per
        // iteration (~15 of them per wave for the 90 x 30 texel rectangles of 1280x960 depth), and nothing else runs in
        // the workgroup meanwhile.  (The tiled kernels are launched for images with 32-bit offsets only: dpitch / npitch.)
        if constexpr (FAST) {
            const int ntex = tw * th;
            const float inv_tw = 1.0f / (float)tw;
            constexpr int SU = 4;
            for (int t0 = tid; t0 < ntex; t0 += NT * SU) {
                float4 n[SU];
                float d[SU];
#pragma unroll
                for (int k = 0; k < SU; ++k) {
                    const int t = t0 + k * NT;
                    n[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    d[k] = -__builtin_inff();
                    if (t < ntex) {
                        const int r = (int)(((float)t + 0.5f) * inv_tw); // t / tw: t < 4096, the quotient is never within 0.5 / tw of an integer
                        const int c = t - r * tw;
                        const unsigned y = (unsigned)(ty0 + r), x = (unsigned)(tx0 + c);
                        n[k] = *reinterpret_cast<const float4*>(p.norm.ptr + (__umul24(y, p.npitch) + x * 16u));
                        d[k] = *reinterpret_cast<const float*>(p.depth.ptr + (__umul24(y, p.dpitch) + x * 4u));
                    }
                }
#pragma unroll
                for (int k = 0; k < SU; ++k) {
                    const int t = t0 + k * NT;
                    if (t < ntex) {
                        s_tile[t] = make_float4(n[k].x, n[k].y, n[k].z, d[k]);
                        dmax = fmaxf(dmax, d[k]);
                    }
                }
            }
        } else {
            // exact numerics: bound by instruction issue, the staging waits are covered by the other workgroups of the CU
            // (and with the flat loop in this instantiation hipcc schedules the voxel loop 11 % slower: 0.438 -> 0.486 ms)
            for (int r = wv; r < th; r += NW) {
                const float* drow = row<float>(p.depth, (size_t)(ty0 + r)) + tx0;
                const float4* nrow = row<float4>(p.norm, (size_t)(ty0 + r)) + tx0;
                for (int c = lane; c < tw; c += 64) {
                    const float4 n = nrow[c];
                    const float d = drow[c];
                    s_tile[(r * tw + c) * (DXT ? 2 : 1)] = make_float4(n.x, n.y, n.z, d);
                    dmax = fmaxf(dmax, d);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off, 64));
        if (lane == 0) s_dmax[wv] = dmax;
    }
    __syncthreads();
    // Occlusion culling of the whole brick.  An update needs costheta > mincos and
    // costheta * (md - Z) > -trunc, hence md - Z > -trunc / mincos; md is a convex combination of
    // texels of the staged rectangle, so md <= dmax, and Z >= zmin.  If even dmax - zmin lies below
    // that bound (with a relative margin far above the rounding of the per-voxel expression), no voxel
    // of the brick can change: skip it before any volume traffic or per-voxel arithmetic.
    if (use_tile && p.mincos > 0.f && p.trunc > 0.f) {
        if constexpr (NW == 4) {
            dmax = fmaxf(fmaxf(s_dmax[0], s_dmax[1]), fmaxf(s_dmax[2], s_dmax[3]));
        } else {
            dmax = s_dmax[0];
#pragma unroll
            for (int k = 1; k < NW; ++k) dmax = fmaxf(dmax, s_dmax[k]);
        }
        const float bound = -(p.trunc / p.mincos) * 1.001f;
        const float dfar = dmax + fabsf(dmax) * 1e-5f;
        if (dfar - zmin < bound) return; // also when every texel is NaN (dmax = -inf)
        // Fast kernels, beyond that.  A voxel that lies D = Z - md behind the surface it sees is updated only if mincos < costheta AND
        // costheta D < trunc, and costheta -- the interpolated normal against the voxel's own viewing direction -- is at least
        // c_min = the smallest value any texel of the rectangle gives with ITS viewing direction, less what one pixel of direction
        // can change (a bilinear sample is a convex combination of its cell's texels; a texel that is not finite makes every
        // sample of its cells NaN or infinite, which the predicate rejects, so such texels do not count).  With c = max(mincos,
        // c_min) no voxel with D >= trunc / c can change: a brick that lies that far behind the FARTHEST depth it sees is skipped
        // whole.  The test above knows c = mincos only -- ten truncation bands for the application's mincostheta = 0.1; behind a
        // wall seen face-on one band is enough, behind the room's side walls (seen at costheta ~ 0.3) three.  c_min costs one pass
        // over the staged tile and is evaluated only for bricks that lie more than a band behind everything they see.  Which
        // bricks are evaluated changes, never a value: same bits (the parity, chain and fuzz suites compare with the oracle).
        if constexpr (FAST && !DXT && CELL::BYTES == 8) {
            const float band = p.trunc * 1.001f;
            if (zmin - dfar > band && p.fuse_cull) { // (uniform)
                float c = __builtin_inff();
                const int ntex = tw * th;
                const float inv_tw = 1.0f / (float)tw;
                const float ifu = __builtin_amdgcn_rcpf(p.K.fu), ifv = __builtin_amdgcn_rcpf(p.K.fv);
                const float slack = 3.0f * fmaxf(fabsf(ifu), fabsf(ifv));   // |change of a unit direction| over one pixel, with room
                for (int t = tid; t < ntex; t += NT) {
                    const float4 q = s_tile[t];
                    const int r = (int)(((float)t + 0.5f) * inv_tw);
                    const int cc = t - r * tw;
                    const float dx = ((float)(tx0 + cc) - p.K.u0) * ifu, dy = ((float)(ty0 + r) - p.K.v0) * ifv;
                    const float rs = __builtin_amdgcn_rsqf(__builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, 1.0f)));
                    const float dotn = __builtin_fmaf(q.x, dx, __builtin_fmaf(q.y, dy, q.z));
                    const float ct = -dotn * rs - (fabsf(q.x) + fabsf(q.y) + fabsf(q.z)) * slack - 1e-4f;
                    if (isfinite(ct) && isfinite(q.w)) c = fminf(c, ct);
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) c = fminf(c, __shfl_xor(c, off, 64));
                if (lane == 0) s_cmin[wv] = c;
                __syncthreads();
                if constexpr (NW == 4) {
                    c = fminf(fminf(s_cmin[0], s_cmin[1]), fminf(s_cmin[2], s_cmin[3]));
                } else {
                    c = s_cmin[0];
#pragma unroll
                    for (int k = 1; k < NW; ++k) c = fminf(c, s_cmin[k]);
                }
                if ((zmin - dfar) * fmaxf(p.mincos, c) >= band) return;   // (c = +inf: no finite texel at all -- the test above has returned)
            }
        }
    }
    if constexpr (DXT) {   // (workgroup-uniform control flow up to here: every thread reaches the barrier)
        if (use_tile) {
            const int ntex = tw * th;
            for (int t = tid; t < ntex; t += NT) {
                const float4 a = s_tile[2 * t], b = s_tile[2 * min(t + 1, ntex - 1)]; // (the last column's difference is never read)
                s_tile[2 * t + 1] = make_float4(b.x - a.x, b.y - a.y, b.z - a.z, b.w - a.w);
            }
        }
        __syncthreads();
    }
    auto march = [&]() {
        // TRACK kernels keep every lane in the march (`live` only gates the updates): the per-group reductions below run in
        // wave-uniform control flow with all lanes present, and the lane mask stays in scalar registers throughout
        if constexpr (!TRACK) {
            if (!live) return;
        }

        // TRACK bookkeeping of the 8-slice group the wave is in: range of the cell values it stored (one v_min3 and one
        // v_max3 per cell pair) and the lanes that updated both of their cells in every slice so far (three scalar
        // instructions per pair) -- the fast kernel is co-limited by vector issue.
        // note_pair: every cell pair of every slice, in wave-uniform control flow (`ok` = the cell is updated).  note_vals: a
        // pair that was loaded, updated (one or both cells) and stored -- both cells enter the range: a cell that was not
        // updated keeps a value that belongs to the brick anyway (NaN is ignored by min3 / max3), so the range stays
        // conservative.  emit_group: the group is complete -- reduce over the lanes of each summary brick (4 neighbours in x
        // = 8 cells, every row of the wave; DPP within the 4, row / half swaps across the rows: no LDS crossbar) and leave
        // {lo, hi, every cell rewritten} per brick in s_part for the workgroup's epilogue.
        float t_mn = __builtin_inff(), t_mx = -__builtin_inff();
        unsigned long long t_all = ~0ull;
        int t_g = 0;
        auto note_pair = [&](bool ok0, bool ok1) {
            if constexpr (TRACK) {
                t_all &= __ballot(ok0) & __ballot(ok1);
            }
        };
        auto note_vals = [&](const float4& c) {
            if constexpr (TRACK) {
                t_mn = __builtin_fminf(__builtin_fminf(t_mn, c.x), c.z);
                t_mx = __builtin_fmaxf(__builtin_fmaxf(t_mx, c.x), c.z);
            }
        };
        auto emit_group = [&](int g, float mn, float mx, unsigned long long all) {
            if constexpr (TRACK) {
                const auto fmin2 = [](float a, float b) { return fminf(a, b); };
                const auto fmax2 = [](float a, float b) { return fmaxf(a, b); };
                mn = wave_xor_combine<1>(mn, fmin2); mx = wave_xor_combine<1>(mx, fmax2);
                mn = wave_xor_combine<2>(mn, fmin2); mx = wave_xor_combine<2>(mx, fmax2);
                if constexpr (LX == 16) { mn = wave_xor_combine<16>(mn, fmin2); mx = wave_xor_combine<16>(mx, fmax2); }
                mn = wave_xor_combine<32>(mn, fmin2); mx = wave_xor_combine<32>(mx, fmax2);
                if ((lane & 3) == 0 && lane < LX) {
                    // lanes of this lane's summary brick within the wave; the group's 8 slices must lie inside the launch
                    const unsigned long long brick = (LX == 32 ? 0x0000000F0000000Full : 0x000F000F000F000Full) << (lane & (LX - 1) & ~3);
                    const int full = (wz1 - wz0 >= 8 * (g + 1) && (all & brick) == brick) ? 1 : 0;
                    float* q = s_part + ((wv * 2 + g) * 8 + (lane >> 2)) * 3;
                    q[0] = mn; q[1] = mx; q[2] = __int_as_float(full);
                }
            }
        };
        auto flush = [&]() {   // before leaving: the current group, and an empty second group if the wave never reached it
            if constexpr (TRACK) {
                emit_group(t_g, t_mn, t_mx, t_all);
                if (NG == 2 && t_g == 0) emit_group(1, __builtin_inff(), -__builtin_inff(), 0ull);
            }
        };
        auto next_group = [&](int z) {
            if constexpr (TRACK && NG == 2) {
                if (z - wz0 == 8) { // uniform
                    emit_group(0, t_mn, t_mx, t_all);
                    t_mn = __builtin_inff(); t_mx = -__builtin_inff(); t_all = ~0ull; t_g = 1;
                }
            }
        };
        const bool upd = !TRACK || live;   // TRACK: a lane outside the extents observes like the others and updates nothing

        // the cell pairs of up to ZU slices.  `keep` bricks (TRACK: the planes the next sweep starts with) use ordinary loads,
        // which leave the lines in the 256 MiB memory-side cache.  Written as asm: given `keep ? plain load : nontemporal
        // load` of one address hipcc emits a single plain load for both cases.  The compiler does not know the loads are in
        // flight: the wait's "+v" operands keep the destination registers allocated and untouched until the data has arrived
        // (the idiom of RayF32::issue / finish, sampling.h); tests/test_gpu_chain.py and test_gpu_summary.py compare the
        // tracked volume with the untracked one bit for bit, at sizes where all and where a quarter of the planes take this path.
        auto load_cells = [&](float4 (&c)[ZU], const bool (&any)[ZU], const unsigned char* at) {
            if constexpr (TRACK && FAST && CELL::BYTES == 8 && ZU <= 2) {   // (the bit-exact kernels are bound by issue, not by memory)
                if (keep) { // uniform
                    v4f raw[ZU];
    #pragma unroll
                    for (int k = 0; k < ZU; ++k)
                        if (any[k]) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(raw[k]) : "v"(at + (size_t)k * p.vimg_pitch) : "memory");
                    if constexpr (ZU == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0]) : : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0]), "+v"(raw[ZU - 1]) : : "memory");
    #pragma unroll
                    for (int k = 0; k < ZU; ++k)
                        if (any[k]) c[k] = make_float4(raw[k].x, raw[k].y, raw[k].z, raw[k].w);
                    return;
                }
            }
    #pragma unroll
            for (int k = 0; k < ZU; ++k)
                if (any[k]) c[k] = CELL::ld2(at + (size_t)k * p.vimg_pitch);
        };

        // one voxel's observation, corners from the LDS tile when the cell lies inside it
        auto observe_tile = [&](int v, float pz) -> Obs {
            Obs o;
            o.ok = false;
            o.val = 0.f;
            o.w = 0.f;
            const V3 Pc = cam[v].at(p, pz);
            float pu, pv, iz;
            project<FAST>(p, Pc, pu, pv, iz);
            if (in_bounds(p, pu, pv)) {
                const float fix = floorf(pu), fiy = floorf(pv);
                const int ix = (int)fix, iy = (int)fiy;
                const unsigned rx = (unsigned)(ix - tx0), ry = (unsigned)(iy - ty0);
                Corners c;
                if (use_tile && rx < (unsigned)(tw - 1) && ry < (unsigned)(th - 1)) {
                    const float4* t = s_tile + (ry * (unsigned)tw + rx);
                    c.c00 = t[0]; c.c01 = t[1]; c.c10 = t[tw]; c.c11 = t[tw + 1];
                } else {
                    c = DMA ? fetch_texels(p, ix, iy) : fetch_global32(p, ix, iy);
                }
                o = finish<FAST>(p, Pc, iz, pu - fix, pv - fiy, c);
            }
            return o;
        };

        // ZU slices per iteration: their volume cells are requested together, so a wave keeps ZU
        // 16-byte reads per lane in flight
        unsigned char* cell = p.vptr + (size_t)wz0 * p.vimg_pitch + (size_t)y * p.vpitch + (size_t)x0 * CELL::BYTES;

        // Tiled bricks run branch-free: tile indices are clamped into the staged rectangle (always a valid LDS
        // address), every lane evaluates the observation, and the bounds / predicate results only gate the
        // update.  (Fast mode is co-limited by VALU issue and HBM -- about 75 VALU ops per voxel against 16 B --
        // and divergent early-outs cost more issue slots than they save.)  A sample that is in bounds but outside
        // the rectangle -- impossible with the one-texel slack, kept as a guard -- sends the wave through the
        // generic per-lane path for that iteration.  Values are the same expressions as the generic path.
        if constexpr (!FAST) {
            // Exact numerics, cheaper instruction sequences (finish_shared): taken when every voxel of the brick keeps the
            // operands of div_core / sqrt_core inside the range where they are the IEEE results (NaN bounds fail the test).
            if (use_tile && p.exact_shared && zmin >= 0x1p-20f && cmax <= 0x1p20f) {
                const int cxmax = in_vgpr(tw - 2), cymax = in_vgpr(th - 2), tx0v = in_vgpr(tx0), ty0v = in_vgpr(ty0);
                // both voxels of the lane in slice z; `any` = the lane has a cell pair to update
                auto observe_pair = [&](auto interior_c, int z, Obs (&o)[2]) -> bool {
                    constexpr bool INTERIOR = decltype(interior_c)::value;
                    const float4 tz = s_tz[z - zbeg];
                    bool stray = false;
    #pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        // cam[v].at(p, pz): the products T(i,2)*pz are uniform per slice and come from LDS
                        const V3 Pc = v3(cam[v].ax + tz.y + p.T.m[3], cam[v].ay + tz.z + p.T.m[7], cam[v].az + tz.w + p.T.m[11]);
                        const float yz = rcp_nr(Pc.z);
                        const float pu = p.K.u0 + div_core(p.K.fu * Pc.x, Pc.z, yz);
                        const float pv = p.K.v0 + div_core(p.K.fv * Pc.y, Pc.z, yz);
                        const float fix = floorf(pu), fiy = floorf(pv);
                        const int rx = (int)fix - tx0v, ry = (int)fiy - ty0v;
                        if constexpr (INTERIOR) { // every lane samples inside the image band and the rectangle
                            if constexpr (DXT) {
                                const float4* t = s_tile + 2 * (ry * tw + rx);
                                o[v] = finish_shared_dx(p, Pc, yz, pu - fix, pv - fiy, t[0], t[1], t[2 * tw], t[2 * tw + 1]);
                            } else {
                                const float4* t = s_tile + (ry * tw + rx);
                                Corners c;
                                c.c00 = t[0]; c.c01 = t[1]; c.c10 = t[tw]; c.c11 = t[tw + 1];
                                o[v] = finish_shared(p, Pc, yz, pu - fix, pv - fiy, c);
                            }
                        } else {
                            const bool inb = INTERIOR ? upd : (upd && in_bounds(p, pu, pv));
                            const bool inside = INTERIOR || ((unsigned)rx <= (unsigned)cxmax && (unsigned)ry <= (unsigned)cymax);
                            if constexpr (DXT) {
                                const float4* t = s_tile + 2 * (med3_i32(ry, 0, cymax) * tw + med3_i32(rx, 0, cxmax));
                                o[v] = finish_shared_dx(p, Pc, yz, pu - fix, pv - fiy, t[0], t[1], t[2 * tw], t[2 * tw + 1]);
                            } else {
                            const float4* t = s_tile + (med3_i32(ry, 0, cymax) * tw + med3_i32(rx, 0, cxmax));
                            Corners c;
                            c.c00 = t[0]; c.c01 = t[1]; c.c10 = t[tw]; c.c11 = t[tw + 1];
                            o[v] = finish_shared(p, Pc, yz, pu - fix, pv - fiy, c);
                            }
                            o[v].ok = ((int)o[v].ok & (int)inb & (int)inside) != 0;
                            stray |= ((int)inb & (int)!inside) != 0;
                        }
                    }
                    if constexpr (!INTERIOR) {
                        if (__builtin_expect(__ballot(stray) != 0ull, 0)) {
                            o[0] = observe<false, true, DMA>(p, cam[0].at(p, tz.x));
                            o[1] = observe<false, true, DMA>(p, cam[1].at(p, tz.x));
                            o[0].ok = o[0].ok && upd; o[1].ok = o[1].ok && upd;
                        }
                    }
                    return ((int)o[0].ok | (int)o[1].ok) != 0;
                };
                // (requesting the cells of slice z and consuming them after the observation of slice z + 1 -- a software
                // pipeline -- was measured and changes nothing: 0.54 ms either way, the loop is bound by instruction issue)
                // ZU slices per iteration: their cells are requested together and consumed after all ZU observations
                auto shared_loop = [&](auto interior_c) {
                for (int z = wz0; z < wz1; z += ZU, cell += (size_t)ZU * p.vimg_pitch) {
                    next_group(z);
                    Obs o[ZU][2];
                    bool any[ZU];
    #pragma unroll
                    for (int k = 0; k < ZU; ++k) {
                        any[k] = false;
                        if (z + k < wz1) { // uniform
                            any[k] = observe_pair(interior_c, z + k, o[k]);
                            note_pair(o[k][0].ok, o[k][1].ok);
                        }
                    }
                    float4 c[ZU];
                    load_cells(c, any, cell);
    #pragma unroll
                    for (int k = 0; k < ZU; ++k)
                        if (any[k]) {
                            if (o[k][0].ok) accumulate<false, CELL>(o[k][0], p.max_w, c[k].x, c[k].y);
                            if (o[k][1].ok) accumulate<false, CELL>(o[k][1], p.max_w, c[k].z, c[k].w);
                            CELL::st2(cell + (size_t)k * p.vimg_pitch, c[k]);
                            note_vals(c[k]);
                        }
                }
                };
                if (interior) shared_loop(std::true_type{});
                else shared_loop(std::false_type{});
                flush();
                return;
            }
        }
        if (use_tile) {
            const int cxmax = PARK ? in_vgpr(tw - 2) : tw - 2, cymax = PARK ? in_vgpr(th - 2) : th - 2, tx0v = PARK ? in_vgpr(tx0) : tx0,
                      ty0v = PARK ? in_vgpr(ty0) : ty0;
            auto tiled_loop = [&](auto interior_c) {
            constexpr bool INTERIOR = decltype(interior_c)::value;
            for (int z = wz0; z < wz1; z += ZU, cell += (size_t)ZU * p.vimg_pitch) {
                next_group(z);
                Obs o[ZU][2];
                bool any[ZU];
                bool stray = false;
    #pragma unroll
                for (int k = 0; k < ZU; ++k) {
                    any[k] = false;
                    if (z + k < wz1) { // uniform
                        const float pz = s_pz[z + k - zbeg];
    #pragma unroll
                        for (int v = 0; v < 2; ++v) {
                            const V3 Pc = cam[v].at(p, pz);
                            float pu, pv, iz;
                            project<FAST>(p, Pc, pu, pv, iz);
                            const float fix = floorf(pu), fiy = floorf(pv);
                            const int rx = (int)fix - tx0v, ry = (int)fiy - ty0v;
                            if constexpr (INTERIOR) { // every lane samples inside the image band and the rectangle
                                const float4* t = s_tile + (ry * tw + rx);
                                Corners c;
                                c.c00 = t[0]; c.c01 = t[1]; c.c10 = t[tw]; c.c11 = t[tw + 1];
                                o[k][v] = finish<FAST>(p, Pc, iz, pu - fix, pv - fiy, c);
                            } else {
                                const bool inb = INTERIOR ? upd : (upd && in_bounds(p, pu, pv));
                                const bool inside = INTERIOR || ((unsigned)rx <= (unsigned)cxmax && (unsigned)ry <= (unsigned)cymax);
                                const float4* t = s_tile + (med3_i32(ry, 0, cymax) * tw + med3_i32(rx, 0, cxmax)); // clamps: one v_med3_i32 each
                                Corners c;
                                c.c00 = t[0]; c.c01 = t[1]; c.c10 = t[tw]; c.c11 = t[tw + 1];
                                o[k][v] = finish<FAST>(p, Pc, iz, pu - fix, pv - fiy, c);
                                o[k][v].ok = ((int)o[k][v].ok & (int)inb & (int)inside) != 0;
                                stray |= ((int)inb & (int)!inside) != 0;
                            }
                        }
                        any[k] = ((int)o[k][0].ok | (int)o[k][1].ok) != 0;
                    }
                }
                if constexpr (!INTERIOR) {
                if (__builtin_expect(__ballot(stray) != 0ull, 0)) {
    #pragma unroll
                    for (int k = 0; k < ZU; ++k)
                        if (z + k < wz1) {
                            const float pz = s_pz[z + k - zbeg];
                            o[k][0] = observe<FAST, true, DMA>(p, cam[0].at(p, pz));
                            o[k][1] = observe<FAST, true, DMA>(p, cam[1].at(p, pz));
                            o[k][0].ok = o[k][0].ok && upd; o[k][1].ok = o[k][1].ok && upd;
                            any[k] = o[k][0].ok || o[k][1].ok;
                        }
                }
                }
    #pragma unroll
                for (int k = 0; k < ZU; ++k)
                    if (z + k < wz1) note_pair(o[k][0].ok, o[k][1].ok); // uniform
                float4 c[ZU];
                load_cells(c, any, cell);
    #pragma unroll
                for (int k = 0; k < ZU; ++k)
                    if (any[k]) {
                        if (o[k][0].ok) accumulate<FAST, CELL>(o[k][0], p.max_w, c[k].x, c[k].y);
                        if (o[k][1].ok) accumulate<FAST, CELL>(o[k][1], p.max_w, c[k].z, c[k].w);
                        CELL::st2(cell + (size_t)k * p.vimg_pitch, c[k]);
                        note_vals(c[k]);
                    }
            }
            };
            if (interior) tiled_loop(std::true_type{});
            else tiled_loop(std::false_type{});
            flush();
            return;
        }
        for (int z = wz0; z < wz1; z += ZU, cell += (size_t)ZU * p.vimg_pitch) {
            next_group(z);
            Obs o[ZU][2];
            bool any[ZU];
    #pragma unroll
            for (int k = 0; k < ZU; ++k) {
                any[k] = false;
                if (z + k < wz1) {
                    const float pz = s_pz[z + k - zbeg];
                    o[k][0] = observe_tile(0, pz);
                    o[k][1] = observe_tile(1, pz);
                    o[k][0].ok = o[k][0].ok && upd; o[k][1].ok = o[k][1].ok && upd;
                    any[k] = o[k][0].ok || o[k][1].ok;
                    note_pair(o[k][0].ok, o[k][1].ok);
                }
            }
            float4 c[ZU];
            load_cells(c, any, cell);
    #pragma unroll
            for (int k = 0; k < ZU; ++k)
                if (any[k]) {
                    if (o[k][0].ok) accumulate<FAST, CELL>(o[k][0], p.max_w, c[k].x, c[k].y);
                    if (o[k][1].ok) accumulate<FAST, CELL>(o[k][1], p.max_w, c[k].z, c[k].w);
                    CELL::st2(cell + (size_t)k * p.vimg_pitch, c[k]);
                    note_vals(c[k]);
                }
        }
        flush();
    };
    march();

    if constexpr (TRACK) {
        // ---- summary epilogue: every thread of the workgroup arrives here (the early exits above are workgroup-uniform
        // and leave the summary as it is: nothing was written) ----
        __syncthreads();
        if (tid < NXB * NZB) {
            const int xb = tid % NXB, zb = tid / NXB;
            float mn = __builtin_inff(), mx = -__builtin_inff();
            int full = 1;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) { // the waves whose slices include summary z-brick zb
                const int g = zb - (w4 / WY) * NG;
                if (g >= 0 && g < NG) {
                    const float* q = s_part + ((w4 * 2 + g) * 8 + xb) * 3;
                    mn = fminf(mn, q[0]); mx = fmaxf(mx, q[1]); full &= __float_as_int(q[2]);
                }
            }
            const bool some = mx > -__builtin_inff(); // an updated cell holds a number, and it entered the range
            const int bx = p.sum_bx0 + bxi * NXB + xb, by = p.sum_by0 + blockIdx.y, bz = p.sum_bz0 + (zbeg + p.zoff_local) / 8 + zb;
            if (some && bx * 8 < p.sum_w && by * 8 < p.sum_h && bz * 8 < p.sum_d) {
                float4* r = p.sum_R + ((size_t)bz * p.sum_nby + by) * p.sum_nbx + bx;
                // state 0: every cell has a value in [lo, hi]; 1: every cell NaN; 2: mixed / unknown
                // `full`: all 8 x 8 x 8 cells were updated (a brick cut by the volume's or the launch's extents never is: it
                // takes the merge below, which is always valid)
                if (full) { // every cell rewritten: the frame's range, no need to know the old one
                    *r = make_float4(mn, mx, __int_as_float(0), 0.f);
                } else {
                    float4 old = *r;
                    if (__float_as_int(old.z) == 1) old = make_float4(mn, mx, __int_as_float(2), 0.f);
                    else old = make_float4(fminf(old.x, mn), fmaxf(old.y, mx), old.z, 0.f);
                    *r = old;
                }
            }
        }
    }
}

// Diagnostics: how many voxels the fuse kernels would update (same predicate, no volume access).
template <bool FAST>
__global__ __launch_bounds__(256) void k_sdf_fuse_count(const FuseParams p, const int bx, const int by, const int bz,
                                                        unsigned long long* __restrict__ count)
{
    // a workgroup walks bricks of 64 x FUSE_ROWS x FUSE_ZC voxels with a grid stride and adds its total once: a single
    // counter takes ~10 ns per atomic, so one atomic per wave-brick (131 k at 512^3) made this kernel 1.6 ms
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned n = 0;
    for (int b = blockIdx.x; b < bx * by * bz; b += gridDim.x) {
        const int x = (b % bx) * 64 + lane;
        const int y = ((b / bx) % by) * FUSE_ROWS + wv;
        const int zbeg = (b / (bx * by)) * FUSE_ZC;
        const int zend = min(zbeg + FUSE_ZC, p.Z);
        if (x >= p.X || y >= p.Y) continue;
        CamXY<FAST> cam;
        cam.init(p, p.bmin.x + p.size.x * (float)x / p.w1, p.bmin.y + p.size.y * (float)y / p.h1);
        for (int z = zbeg; z < zend; ++z) {
            const float pz = p.bmin.z + p.size.z * (float)(z + p.zoff) / p.d1;
            n += observe<FAST, false>(p, cam.at(p, pz)).ok ? 1u : 0u;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64); // wave64 butterfly sum
    if (lane == 0 && n) atomicAdd(count, (unsigned long long)n);
}

// SdfReset: contiguous fill of (trunc, 0) over [ptr, RowPtr(h-1,d-1)+w) (Volume.h:343-356).
__global__ __launch_bounds__(256) void k_fill_sdf(float2* __restrict__ base, size_t n_cells, float val, float w)
{
    // one workgroup = one contiguous 16 KiB chunk, four 16-byte nontemporal stores per thread and no loop: a pure write stream
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const size_t n2 = n_cells / 2;
    v4f* b4 = reinterpret_cast<v4f*>(base);
    v4f v4;
    v4.x = val; v4.y = w; v4.z = val; v4.w = w;
    if (i + 768 < n2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(v4, b4 + i + k * 256);
    } else {
        for (size_t j = i; j < n2; j += 256) __builtin_nontemporal_store(v4, b4 + j);
    }
    if (i == 0 && (n_cells & 1)) base[n_cells - 1] = make_float2(val, w);
}
__global__ __launch_bounds__(256) void k_fill_sdf_unaligned(float2* __restrict__ base, size_t n_cells, float val, float w)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_cells; j += stride)
        base[j] = make_float2(val, w);
}

// 4-byte (fp16) cells: fill the contiguous span with one 32-bit pattern
__global__ __launch_bounds__(256) void k_fill_u32(unsigned* __restrict__ base, size_t n, unsigned pattern)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) base[j] = pattern;
}

// SdfSphere (cu_sdffusion.cu:175-195): val = |pos - c| - r, w = 1.
template <typename CELL>
__global__ __launch_bounds__(256) void k_sdf_sphere(VolView v, int X, int Y, int Z, V3 c, float r)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int z = blockIdx.z;
    if (x >= X || y >= Y || z >= Z) return;
    const V3 size = v.bmax - v.bmin;
    const V3 pos = v3(v.bmin.x + size.x * (float)x / (float)(v.w - 1), v.bmin.y + size.y * (float)y / (float)(v.h - 1),
                      v.bmin.z + size.z * (float)z / (float)(v.d - 1));
    const float dist = length(pos - c);
    CELL::st1(v.ptr + (size_t)z * v.img_pitch + (size_t)y * v.pitch + (size_t)x * CELL::BYTES, make_float2(dist - r, 1.0f));
}


// ---------------------------------------------------------------------------------------
// Colour TSDF fusion: cu_sdffusion.cu:70-138.  Besides the SDF update the kernel keeps a
// grey-level volume (BoundedVolume<float>) as the weighted running mean of the RGB image sampled at the
// voxel's projection into the colour camera (T_iw, Kimg).  Reference launch: 16x16 threads over x/y, all of z
// in a loop -- so x/y extents are (dim/16)*16 and every slice is visited.  Exact mode: IEEE
// arithmetic in the reference's order; fast mode: rcp / rsq / FMA as in the grey kernel, the colour mean with float
// reciprocals (the channel sum is interpolated once: bilinear interpolation is linear).  A lane owns one voxel column segment:
// 8-byte SDF cells and 4-byte colour cells are both contiguous across the wave.
// ---------------------------------------------------------------------------------------
struct ColorParams {
    unsigned char* cptr;        // BoundedVolume<float>
    size_t cpitch, cimg_pitch;
    Pose Ti;                    // T_iw
    Intr Ki;                    // Kimg
    ImgView img;                // Image<uchar3>
    float iwb, ihb;             // (float)img.w - 2, (float)img.h - 2
};

struct __attribute__((packed)) U3 { unsigned char x, y, z; };

// Image<uchar3>::GetBilinear<float3> (Image.h:317-334) with lerp(uchar3, uchar3, float) of sampling.h:23-30
// (integer difference, converted to float, times t, plus the first value) and the float3 lerp of
// cutil_math.h:375-378 across rows; then ConvertPixel<float,float3> (pixel_convert.h:159-165) and the
// application's "/ 255.0", a double division rounded back to float (cu_sdffusion.cu:98).
// Texels arrive packed as r | g << 8 | b << 16 (from global memory or from the LDS tile: copies either way).
// Fast numerics: the channel sum is interpolated once (bilinear interpolation is linear) and scaled by the
// float constant 1 / (3 * 255).
struct Rgb4 { unsigned b0, b1, t0, t1; };

template <bool FAST>
__device__ __forceinline__ float grey_bilinear(const Rgb4 t, float fx, float fy)
{
    if constexpr (FAST) {
        auto sum = [](unsigned u) { return (float)((u & 0xffu) + ((u >> 8) & 0xffu) + ((u >> 16) & 0xffu)); };
        const float lo = lerp_f(sum(t.b0), sum(t.b1), fx), hi = lerp_f(sum(t.t0), sum(t.t1), fx);
        return lerp_f(lo, hi, fy) * (1.0f / 765.0f);
    } else {
        auto ch = [](unsigned u, int k) { return (int)((u >> (8 * k)) & 0xffu); };
        float c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float lo = (float)ch(t.b0, k) + fx * (float)(ch(t.b1, k) - ch(t.b0, k));
            const float hi = (float)ch(t.t0, k) + fx * (float)(ch(t.t1, k) - ch(t.t0, k));
            c[k] = lo + fy * (hi - lo);
        }
        const float grey = (c[0] + c[1] + c[2]) / 3.0f;
        return (float)((double)grey / 255.0);
    }
}

__device__ __forceinline__ unsigned pack_rgb(const U3 c) { return (unsigned)c.x | ((unsigned)c.y << 8) | ((unsigned)c.z << 16); }

__device__ __forceinline__ Rgb4 fetch_rgb_global(const ColorParams& q, int ix, int iy)
{
    const U3* bl = reinterpret_cast<const U3*>(q.img.ptr + (size_t)iy * q.img.pitch) + (size_t)ix;
    const U3* tl = reinterpret_cast<const U3*>(q.img.ptr + (size_t)(iy + 1) * q.img.pitch) + (size_t)ix;
    return Rgb4{pack_rgb(bl[0]), pack_rgb(bl[1]), pack_rgb(tl[0]), pack_rgb(tl[1])};
}

// T_iw * P_w with the x/y terms hoisted: leading partial sums of the reference expression (MatUtils.h:117-125)
struct ColCam {
    float ax, ay, az;
    __device__ __forceinline__ void init(const ColorParams& q, float px, float py)
    {
        ax = q.Ti.m[0] * px + q.Ti.m[1] * py;
        ay = q.Ti.m[4] * px + q.Ti.m[5] * py;
        az = q.Ti.m[8] * px + q.Ti.m[9] * py;
    }
    __device__ __forceinline__ V3 at(const ColorParams& q, float pz) const
    {
        return v3(ax + q.Ti.m[2] * pz + q.Ti.m[3], ay + q.Ti.m[6] * pz + q.Ti.m[7], az + q.Ti.m[10] * pz + q.Ti.m[11]);
    }
};

template <bool FAST>
__device__ __forceinline__ void project_color(const ColorParams& q, const V3 Pi, float& qu, float& qv)
{
    if constexpr (FAST) {
        const float izi = __builtin_amdgcn_rcpf(Pi.z);
        qu = __builtin_fmaf(q.Ki.fu * Pi.x, izi, q.Ki.u0);
        qv = __builtin_fmaf(q.Ki.fv * Pi.y, izi, q.Ki.v0);
    } else {
        qu = q.Ki.u0 + q.Ki.fu * Pi.x / Pi.z;
        qv = q.Ki.v0 + q.Ki.fv * Pi.y / Pi.z;
    }
}

__device__ __forceinline__ bool in_bounds_color(const ColorParams& q, float qu, float qv)
{
    return 2.0f <= qu && qu < q.iwb && 2.0f <= qv && qv < q.ihb;
}

// One voxel through global memory: both bounds tests, the observation and the grey level (cu_sdffusion.cu:84-99)
template <bool FAST>
__device__ __forceinline__ Obs observe_color_global(const FuseParams& p, const ColorParams& q, const V3 Pc, const V3 Pi, float& grey)
{
    Obs o;
    o.ok = false;
    o.val = 0.f;
    o.w = 0.f;
    grey = 0.f;
    float pu, pv, iz, qu, qv;
    project<FAST>(p, Pc, pu, pv, iz);
    project_color<FAST>(q, Pi, qu, qv);
    if (in_bounds(p, pu, pv) && in_bounds_color(q, qu, qv)) {
        const float fix = floorf(pu), fiy = floorf(pv);
        o = finish<FAST>(p, Pc, iz, pu - fix, pv - fiy, fetch_global64(p, (int)fix, (int)fiy));
        if (o.ok) {
            const float gix = floorf(qu), giy = floorf(qv);
            grey = grey_bilinear<FAST>(fetch_rgb_global(q, (int)gix, (int)giy), qu - gix, qv - giy);
        }
    }
    return o;
}

// colour running mean with the weight pair of the SDF update (cu_sdffusion.cu:100-104): `w` is the new sample's
// weight, `curw` the stored weight before the update
template <bool FAST>
__device__ __forceinline__ float color_mean(float w, float grey, float cc, float curw)
{
    if constexpr (FAST) return __builtin_fmaf(w, grey, cc * curw) * __builtin_amdgcn_rcpf(w + curw);
    else return (w * grey + cc * curw) / (w + curw);
}

template <bool FAST>
__global__ __launch_bounds__(256) void k_sdf_fuse_color(const FuseParams p, const ColorParams q)
{
    __shared__ float s_pz[FUSE_ZC];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const int y = blockIdx.y * FUSE_ROWS + wv;
    const int zbeg = blockIdx.z * FUSE_ZC;
    const int zend = min(zbeg + FUSE_ZC, p.Z);
    if (threadIdx.x < FUSE_ZC) s_pz[threadIdx.x] = p.bmin.z + p.size.z * (float)(zbeg + (int)threadIdx.x) / p.d1;
    __syncthreads();
    if (x >= p.X || y >= p.Y) return;

    const float px = p.bmin.x + p.size.x * (float)x / p.w1;
    const float py = p.bmin.y + p.size.y * (float)y / p.h1;
    CamXY<FAST> cam;
    cam.init(p, px, py);
    ColCam ccam;
    ccam.init(q, px, py);

    unsigned char* cell = p.vptr + (size_t)zbeg * p.vimg_pitch + (size_t)y * p.vpitch + (size_t)x * 8;
    unsigned char* ccell = q.cptr + (size_t)zbeg * q.cimg_pitch + (size_t)y * q.cpitch + (size_t)x * 4;
    for (int z = zbeg; z < zend; ++z, cell += p.vimg_pitch, ccell += q.cimg_pitch) {
        const float pz = s_pz[z - zbeg];
        float grey;
        const Obs o = observe_color_global<FAST>(p, q, cam.at(p, pz), ccam.at(q, pz), grey);
        if (o.ok) {
            float2 cur = *reinterpret_cast<const float2*>(cell);
            const float curw = cur.y;
            accumulate<FAST, CellF32>(o, p.max_w, cur.x, cur.y);
            *reinterpret_cast<float2*>(cell) = cur;
            float* cc = reinterpret_cast<float*>(ccell);
            *cc = color_mean<FAST>(o.w, grey, *cc, curw);
        }
    }
}

// ---------------------------------------------------------------------------------------
// LDS-tiled colour fusion: the brick / rectangle scheme of k_sdf_fuse_tiled with a second rectangle, the
// brick's footprint in the RGB image, staged beside the depth / normal tile as packed r | g << 8 | b << 16
// texels.  A lane owns two x-adjacent voxels (16-byte SDF and 8-byte colour accesses).  Bricks whose
// rectangles do not fit, or that straddle either camera plane, take the global-gather path per voxel;
// texel values are copies, so results equal k_sdf_fuse_color's in either numerics mode.
// ---------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

template <bool FAST>
__global__ __launch_bounds__(256) void k_sdf_fuse_color_tiled(const FuseParams p, const ColorParams q, const int cap_px, const int cap_cpx)
{
    extern __shared__ __attribute__((aligned(16))) float4 s_tile[];
    unsigned* const s_rgb = reinterpret_cast<unsigned*>(s_tile + cap_px);
    __shared__ float s_pz[FUSE_ZC];
    __shared__ float s_box[4][9];
    __shared__ float s_dmax[4];
    __shared__ int s_bad[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware brick order, as k_sdf_fuse_tiled: the x-brick rotated by the z-brick
    const int bxi = p.xcd_swizzle ? (int)((blockIdx.x + (blockIdx.z >> (p.xcd_swizzle - 1))) % gridDim.x) : (int)blockIdx.x;
    const int x0 = (bxi * 32 + (lane & 31)) * 2;
    const int y = blockIdx.y * TB_Y + wv * 2 + (lane >> 5);
    const int zbeg = blockIdx.z * FUSE_ZC;
    const int zend = min(zbeg + FUSE_ZC, p.Z);
    const bool live = x0 < p.X && y < p.Y;
    if (tid < FUSE_ZC) s_pz[tid] = p.bmin.z + p.size.z * (float)(zbeg + tid + p.zoff) / p.d1;
    __syncthreads();

    const float py = p.bmin.y + p.size.y * (float)y / p.h1;
    CamXY<FAST> cam[2];
    ColCam ccam[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const float px = p.bmin.x + p.size.x * (float)(x0 + v) / p.w1;
        cam[v].init(p, px, py);
        ccam[v].init(q, px, py);
    }

    // ---- pixel rectangles of the brick in both images: projections of its first and last slice ----
    const float inf = __builtin_inff();
    float lo[5] = {inf, inf, inf, inf, inf};      // umin, vmin, qumin, qvmin, zmin
    float hi[4] = {-inf, -inf, -inf, -inf};       // umax, vmax, qumax, qvmax
    bool bad = false;
    if (live) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float pz = s_pz[e ? (zend - 1 - zbeg) : 0];
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const V3 Pc = cam[v].at(p, pz), Pi = ccam[v].at(q, pz);
                float pu, pv, iz, qu, qv;
                project<FAST>(p, Pc, pu, pv, iz);
                project_color<FAST>(q, Pi, qu, qv);
                bad = bad || !(Pc.z > 0.f) || !(Pi.z > 0.f) || !(fabsf(pu) < 1e9f) || !(fabsf(pv) < 1e9f) || !(fabsf(qu) < 1e9f) || !(fabsf(qv) < 1e9f);
                lo[0] = fminf(lo[0], pu); hi[0] = fmaxf(hi[0], pu);
                lo[1] = fminf(lo[1], pv); hi[1] = fmaxf(hi[1], pv);
                lo[2] = fminf(lo[2], qu); hi[2] = fmaxf(hi[2], qu);
                lo[3] = fminf(lo[3], qv); hi[3] = fmaxf(hi[3], qv);
                lo[4] = fminf(lo[4], Pc.z);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { // wave64 butterfly
#pragma unroll
        for (int k = 0; k < 5; ++k) lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
#pragma unroll
        for (int k = 0; k < 4; ++k) hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
    }
    const bool wave_bad = __ballot(bad) != 0ull;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) s_box[wv][k] = lo[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) s_box[wv][5 + k] = hi[k];
        s_bad[wv] = wave_bad ? 1 : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; ++k) lo[k] = fminf(fminf(s_box[0][k], s_box[1][k]), fminf(s_box[2][k], s_box[3][k]));
#pragma unroll
    for (int k = 0; k < 4; ++k) hi[k] = fmaxf(fmaxf(s_box[0][5 + k], s_box[1][5 + k]), fmaxf(s_box[2][5 + k], s_box[3][5 + k]));
    const bool any_bad = (s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3]) != 0;
    const float zmin = lo[4];

    bool use_tile = false;
    int tx0 = 0, ty0 = 0, tw = 0, th = 0, cx0 = 0, cy0 = 0, cw = 0, chh = 0;
    if (!any_bad) {
        // an update needs both samples inside their image's border band: a rectangle that misses it means no voxel
        // of the brick can change (workgroup-uniform exit)
        if (hi[0] < 2.0f || !(lo[0] < p.dwb) || hi[1] < 2.0f || !(lo[1] < p.dhb)) return;
        if (hi[2] < 2.0f || !(lo[2] < q.iwb) || hi[3] < 2.0f || !(lo[3] < q.ihb)) return;
        const float fx0 = fmaxf(floorf(lo[0]) - 1.f, 0.f), fx1 = fminf(floorf(hi[0]) + 2.f, (float)(p.depth.w - 1));
        const float fy0 = fmaxf(floorf(lo[1]) - 1.f, 0.f), fy1 = fminf(floorf(hi[1]) + 2.f, (float)(p.depth.h - 1));
        tx0 = (int)fx0; ty0 = (int)fy0;
        tw = (int)fx1 - tx0 + 1; th = (int)fy1 - ty0 + 1;
        const float gx0 = fmaxf(floorf(lo[2]) - 1.f, 0.f), gx1 = fminf(floorf(hi[2]) + 2.f, (float)(q.img.w - 1));
        const float gy0 = fmaxf(floorf(lo[3]) - 1.f, 0.f), gy1 = fminf(floorf(hi[3]) + 2.f, (float)(q.img.h - 1));
        cx0 = (int)gx0; cy0 = (int)gy0;
        cw = (int)gx1 - cx0 + 1; chh = (int)gy1 - cy0 + 1;
        use_tile = tw > 1 && th > 1 && tw * th <= cap_px && cw > 1 && chh > 1 && cw * chh <= cap_cpx;
    }
    float dmax = -inf;
    if (use_tile) { // cooperative, row-coalesced staging of both rectangles
        for (int r = wv; r < th; r += 4) {
            const float* drow = row<float>(p.depth, (size_t)(ty0 + r)) + tx0;
            const float4* nrow = row<float4>(p.norm, (size_t)(ty0 + r)) + tx0;
            for (int c = lane; c < tw; c += 64) {
                const float4 n = nrow[c];
                const float d = drow[c];
                s_tile[r * tw + c] = make_float4(n.x, n.y, n.z, d);
                dmax = fmaxf(dmax, d);
            }
        }
        for (int r = wv; r < chh; r += 4) {
            const U3* irow = reinterpret_cast<const U3*>(q.img.ptr + (size_t)(cy0 + r) * q.img.pitch) + cx0;
            for (int c = lane; c < cw; c += 64) s_rgb[r * cw + c] = pack_rgb(irow[c]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off, 64));
        if (lane == 0) s_dmax[wv] = dmax;
    }
    __syncthreads();
    // occlusion culling of the whole brick: see k_sdf_fuse_tiled
    if (use_tile && p.mincos > 0.f && p.trunc > 0.f) {
        dmax = fmaxf(fmaxf(s_dmax[0], s_dmax[1]), fmaxf(s_dmax[2], s_dmax[3]));
        const float bound = -(p.trunc / p.mincos) * 1.001f;
        if (dmax + fabsf(dmax) * 1e-5f - zmin < bound) return;
    }
    if (!live) return;

    unsigned char* cell = p.vptr + (size_t)zbeg * p.vimg_pitch + (size_t)y * p.vpitch + (size_t)x0 * 8;
    unsigned char* ccell = q.cptr + (size_t)zbeg * q.cimg_pitch + (size_t)y * q.cpitch + (size_t)x0 * 4;
    const int cxmax = tw - 2, cymax = th - 2, gxmax = cw - 2, gymax = chh - 2;
    for (int z = zbeg; z < zend; ++z, cell += p.vimg_pitch, ccell += q.cimg_pitch) {
        const float pz = s_pz[z - zbeg];
        Obs o[2];
        float grey[2];
        bool stray = !use_tile;
        if (use_tile) { // branch-free: indices clamped into the staged rectangles, results gate the update only
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const V3 Pc = cam[v].at(p, pz), Pi = ccam[v].at(q, pz);
                float pu, pv, iz, qu, qv;
                project<FAST>(p, Pc, pu, pv, iz);
                project_color<FAST>(q, Pi, qu, qv);
                const bool inb = in_bounds(p, pu, pv) && in_bounds_color(q, qu, qv);
                const float fix = floorf(pu), fiy = floorf(pv), gix = floorf(qu), giy = floorf(qv);
                const int rx = (int)fix - tx0, ry = (int)fiy - ty0, gx = (int)gix - cx0, gy = (int)giy - cy0;
                const bool inside = (unsigned)rx <= (unsigned)cxmax && (unsigned)ry <= (unsigned)cymax &&
                                    (unsigned)gx <= (unsigned)gxmax && (unsigned)gy <= (unsigned)gymax;
                const float4* t = s_tile + (min(max(ry, 0), cymax) * tw + min(max(rx, 0), cxmax));
                Corners c;
                c.c00 = t[0]; c.c01 = t[1]; c.c10 = t[tw]; c.c11 = t[tw + 1];
                o[v] = finish<FAST>(p, Pc, iz, pu - fix, pv - fiy, c);
                o[v].ok = ((int)o[v].ok & (int)inb & (int)inside) != 0;
                const unsigned* g = s_rgb + (min(max(gy, 0), gymax) * cw + min(max(gx, 0), gxmax));
                const Rgb4 texels{g[0], g[1], g[cw], g[cw + 1]};
                if constexpr (FAST) {
                    grey[v] = grey_bilinear<true>(texels, qu - gix, qv - giy);
                } else { // IEEE path: a float and a double division per sample -- worth a divergent skip
                    grey[v] = 0.f;
                    if (o[v].ok) grey[v] = grey_bilinear<false>(texels, qu - gix, qv - giy);
                }
                stray |= ((int)inb & (int)!inside) != 0;
            }
        }
        if (__builtin_expect(__ballot(stray) != 0ull, 0)) {
#pragma unroll
            for (int v = 0; v < 2; ++v) o[v] = observe_color_global<FAST>(p, q, cam[v].at(p, pz), ccam[v].at(q, pz), grey[v]);
        }
        if (o[0].ok || o[1].ok) {
            float4 c = CellF32::ld2(cell);
            const v2f k = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(ccell));
            float k0 = k.x, k1 = k.y;
            if (o[0].ok) {
                k0 = color_mean<FAST>(o[0].w, grey[0], k0, c.y);
                accumulate<FAST, CellF32>(o[0], p.max_w, c.x, c.y);
            }
            if (o[1].ok) {
                k1 = color_mean<FAST>(o[1].w, grey[1], k1, c.w);
                accumulate<FAST, CellF32>(o[1], p.max_w, c.z, c.w);
            }
            CellF32::st2(cell, c);
            v2f kk;
            kk.x = k0; kk.y = k1;
            __builtin_nontemporal_store(kk, reinterpret_cast<v2f*>(ccell));
        }
    }
}

// SdfReset(BoundedVolume<float>): vol.Fill(0.5) over the contiguous span (cu_sdffusion.cu:166-169, Volume.h:343-356)
__global__ __launch_bounds__(256) void k_fill_f32(float* __restrict__ base, size_t n, float v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) base[i] = v;
}

} // namespace kfx

using namespace kfx;

// ---- the packed texel image of the tiled kernels' LDS-DMA staging --------------------------------------------------------
// {nx, ny, nz, depth} per pixel of the depth image: copies of the normal map's xyz and of the depth image (what the kernels'
// register path packs texel by texel while it stages).
// ... and the maximum finite depth of every block of 8 x 4 pixels (the workgroup's tile is 64 x 4: eight blocks), -inf where a
// block has none: what the kernels bound a brick's farthest depth with (bmax: rows of bw8 floats, one row per four image rows).
__global__ __launch_bounds__(256) void k_pack_texels(const ImgView depth, const ImgView norm, unsigned char* __restrict__ tex, const size_t tpitch,
                                                     float* __restrict__ bmax, const unsigned bw8)
{
    __shared__ float s_max[4][8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int u = blockIdx.x * 64 + lane, v = blockIdx.y * 4 + wv;
    float d = -__builtin_inff();
    if (u < depth.w && v < depth.h) {
        const float4 n = row<float4>(norm, (size_t)v)[u];
        const float dd = row<float>(depth, (size_t)v)[u];
        reinterpret_cast<float4*>(tex + (size_t)v * tpitch)[u] = make_float4(n.x, n.y, n.z, dd);
        d = dd;
    }
    d = wave8_combine(fmaxf(d, -__builtin_inff()), [](float a, float b) { return fmaxf(a, b); });   // (fmaxf drops NaN)
    if ((lane & 7) == 0) s_max[wv][lane >> 3] = d;
    __syncthreads();
    if (threadIdx.x < 8)
        bmax[(size_t)blockIdx.y * bw8 + blockIdx.x * 8 + threadIdx.x] =
            fmaxf(fmaxf(s_max[0][threadIdx.x], s_max[1][threadIdx.x]), fmaxf(s_max[2][threadIdx.x], s_max[3][threadIdx.x]));
}

// the packed image's geometry for a w x h depth image: texel rows of tpitch bytes, then the block maxima
struct TexLayout { size_t tpitch, bmax_off, bytes; unsigned bw8, bh4; };
static TexLayout tex_layout(size_t w, size_t h)
{
    TexLayout t;
    t.tpitch = (w * 16 + 255) / 256 * 256;
    t.bw8 = (unsigned)((w + 63) / 64 * 8);
    t.bh4 = (unsigned)((h + 3) / 4);
    t.bmax_off = t.tpitch * h;
    t.bytes = t.bmax_off + (size_t)t.bw8 * t.bh4 * sizeof(float);
    return t;
}
size_t kfx::texel_image_bytes(size_t w, size_t h) { return tex_layout(w, h).bytes; }

// Library scratch for callers that bring depth and normals only (kfx_sdf_fuse and its siblings; kfx_frame_step brings the packed
// image its fused preprocess wrote).  One buffer per calling thread, device and stream: the pack launch and the SdfFuse launches
// that read it are enqueued on the same stream by the same thread, so the next call's pack is ordered behind this call's reads;
// another stream -- or another thread on the same stream: the tests' rank threads share the null stream -- gets its own buffer.
struct TexScratch {
    static constexpr int N = 4;
    struct Entry { void* buf; size_t bytes; hipStream_t stream; int device; unsigned long used; };
    Entry e[N] = {};
    unsigned long tick = 0;
    bool main_thread = false;
    ~TexScratch()
    {
        // threads that come and go (rank threads of the tests and of apps --transport threads) give their buffers back; the main
        // thread's go with the process (its destructor may run when the runtime is already on its way out)
        if (main_thread) return;
        for (Entry& x : e)
            if (x.buf) { (void)hipFree(x.buf); (void)hipGetLastError(); }
    }
    void* get(size_t bytes, hipStream_t stream)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        Entry* hit = nullptr;
        Entry* lru = &e[0];
        for (Entry& x : e) {
            if (x.buf && x.stream == stream && x.device == dev) { hit = &x; break; }
            if (!x.buf) { if (lru->buf) lru = &x; } else if (lru->buf && x.used < lru->used) lru = &x;
        }
        if (!hit) {
            hit = lru;
            if (hit->buf) {   // (a fifth stream: the oldest entry's stream must be done with its buffer; hipFree synchronises)
                (void)hipFree(hit->buf); (void)hipGetLastError();
                hit->buf = nullptr; hit->bytes = 0;
            }
            hit->stream = stream; hit->device = dev;
        }
        if (hit->bytes < bytes) {
            if (hit->buf) { (void)hipFree(hit->buf); (void)hipGetLastError(); hit->buf = nullptr; hit->bytes = 0; }
            if (hipMalloc(&hit->buf, bytes) != hipSuccess) { (void)hipGetLastError(); hit->buf = nullptr; return nullptr; }
            hit->bytes = bytes;
        }
        hit->used = ++tick;
        return hit->buf;
    }
};
static void* tex_scratch(size_t bytes, hipStream_t stream)
{
    thread_local TexScratch pool;
    static const long main_tid = (long)getpid();
    pool.main_thread = (long)syscall(SYS_gettid) == main_tid;
    return pool.get(bytes, stream);
}

static int check_volume(const kfx_volume* vol, size_t cell = 8)
{
    if (!vol || !vol->ptr) return set_error(KFX_E_NULL, "volume is null");
    if (vol->w == 0 || vol->h == 0 || vol->d == 0 || vol->w > 65535 || vol->h > 65535 || vol->d > 65535)
        return set_error(KFX_E_SHAPE, "volume dimensions");
    if (vol->pitch < vol->w * cell || vol->img_pitch < vol->pitch * (vol->h - 1) + vol->w * cell)
        return set_error(KFX_E_SHAPE, "volume pitch smaller than a row / slice");
    if (((uintptr_t)vol->ptr | vol->pitch | vol->img_pitch) & (cell - 1)) return set_error(KFX_E_ALIGN, "volume not aligned to its cell size");
    return 0;
}

static VolView vol_view(const kfx_volume* vol)
{
    VolView v;
    v.ptr = (unsigned char*)vol->ptr;
    v.pitch = vol->pitch;
    v.img_pitch = vol->img_pitch;
    v.w = (int)vol->w;
    v.h = (int)vol->h;
    v.d = (int)vol->d;
    v.bmin = V3{vol->boxmin[0], vol->boxmin[1], vol->boxmin[2]};
    v.bmax = V3{vol->boxmax[0], vol->boxmax[1], vol->boxmax[2]};
    return v;
}

// Fills the kernel parameters; *small_images tells whether the 32-bit image offsets are usable.
// true when div_uniform(size * i, dim - 1) equals size * i / (dim - 1) for every i of the axis (host arithmetic: fmaf and the division
// are correctly rounded, as on the device); the last answer per axis is remembered
static bool pos_div_verified(int axis, float size, int dim)
{
    struct Memo { float size; int dim; bool ok; };
    static thread_local Memo memo[2] = {{0.0f, 0, false}, {0.0f, 0, false}};
    Memo& m = memo[axis];
    if (m.dim == dim && memcmp(&m.size, &size, sizeof(float)) == 0) return m.ok;
    const float n1 = (float)(dim - 1), inv = 1.0f / n1;
    bool ok = div_uniform_safe_host(size) && div_uniform_safe_host(size * n1);
    for (int i = 0; ok && i < dim; ++i) {
        const float a = size * (float)i;
        const float q0 = a * inv;
        const float q = fmaf(fmaf(-n1, q0, a), inv, q0), d = a / n1;
        ok = memcmp(&q, &d, sizeof(float)) == 0;
    }
    m = {size, dim, ok};
    return ok;
}

static int fuse_params(FuseParams& p, bool* small_images, const kfx_volume* vol, const kfx_image* depth,
                       const kfx_image* norm, const float T_cw[12], const float K[4], float trunc_dist, float max_w,
                       float mincostheta, unsigned flags, size_t cell = 8, const kfx_slab* slab = nullptr)
{
    if (int e = check_volume(vol, cell)) return e;
    if (!depth || !norm || !depth->ptr || !norm->ptr || !T_cw || !K) return set_error(KFX_E_NULL, "SdfFuse: null argument");
    if (depth->w < 4 || depth->h < 4 || norm->w < depth->w || norm->h < depth->h)
        return set_error(KFX_E_SHAPE, "SdfFuse: depth/normal image dimensions");
    if (depth->pitch < depth->w * 4 || norm->pitch < depth->w * 16) return set_error(KFX_E_SHAPE, "SdfFuse: image pitch");
    if ((((uintptr_t)depth->ptr | depth->pitch) & 3) || (((uintptr_t)norm->ptr | norm->pitch) & 15))
        return set_error(KFX_E_ALIGN, "SdfFuse: image alignment");
    p.vptr = (unsigned char*)vol->ptr;
    p.vpitch = vol->pitch;
    p.vimg_pitch = vol->img_pitch;
    const bool full = (flags & KFX_FUSE_FULL_EXTENT) != 0;
    p.X = full ? (int)vol->w : (int)(vol->w / 8) * 8;
    p.Y = full ? (int)vol->h : (int)(vol->h / 8) * 8;
    p.Z = full ? (int)vol->d : (int)(vol->d / 8) * 8;
    if (!full && (flags & KFX_FUSE_SLAB_EXTENT) && slab) {
        // the reference's extents on the WHOLE volume (quirk Q1): x / y as above, z = the local planes below (full_d / 8) * 8
        const size_t zlim = (slab->full_d / 8) * 8;
        const size_t z_end = slab->z_offset + vol->d < zlim ? slab->z_offset + vol->d : zlim;
        p.Z = z_end > slab->z_offset ? (int)(z_end - slab->z_offset) : 0;
    }
    p.w1 = (float)(vol->w - 1);
    p.h1 = (float)(vol->h - 1);
    p.d1 = (float)(vol->d - 1);
    p.zoff = 0;
    p.bmin = V3{vol->boxmin[0], vol->boxmin[1], vol->boxmin[2]};
    p.size = V3{vol->boxmax[0] - vol->boxmin[0], vol->boxmax[1] - vol->boxmin[1], vol->boxmax[2] - vol->boxmin[2]};
    if (slab) {
        // the local volume is planes [z_offset, z_offset + d) of a volume with full_d planes spanning
        // [full_zmin, full_zmax]: voxel positions are computed with the FULL volume's expression, so a
        // slab integrates bit-identically to the same planes of the monolithic volume
        if (slab->full_d < 2 || slab->z_offset + vol->d > slab->full_d) return set_error(KFX_E_SHAPE, "SdfFuse: slab outside the full volume");
        p.d1 = (float)(slab->full_d - 1);
        p.zoff = (int)slab->z_offset;
        p.bmin.z = slab->full_zmin;
        p.size.z = slab->full_zmax - slab->full_zmin;
    }
    for (int i = 0; i < 12; ++i) p.T.m[i] = T_cw[i];
    p.K = Intr{K[0], K[1], K[2], K[3]};
    p.depth = ImgView{(const unsigned char*)depth->ptr, depth->pitch, (int)depth->w, (int)depth->h};
    p.norm = ImgView{(const unsigned char*)norm->ptr, norm->pitch, (int)norm->w, (int)norm->h};
    p.dwb = (float)depth->w - 2.0f;
    p.dhb = (float)depth->h - 2.0f;
    p.trunc = trunc_dist;
    p.max_w = max_w;
    p.mincos = mincostheta;
    // 32-bit image offsets (saddr loads, 24-bit multiplies) need images below these limits
    *small_images = depth->pitch * depth->h < (1ull << 31) && norm->pitch * norm->h < (1ull << 31) &&
                    depth->pitch < (1u << 24) && norm->pitch < (1u << 24) && depth->h < (1u << 24);
    p.dpitch = *small_images ? (unsigned)depth->pitch : 0u;
    p.npitch = *small_images ? (unsigned)norm->pitch : 0u;
    p.sum_R = nullptr;
    p.sum_nbx = p.sum_nby = p.sum_bx0 = p.sum_by0 = p.sum_bz0 = p.sum_w = p.sum_h = p.sum_d = p.zoff_local = 0;
    p.tex = nullptr;
    p.tpitch = 0;
    p.bmax = nullptr;
    p.bw8 = 0;
    // div_uniform is taken only where the host has compared it with the IEEE division for EVERY index of the axis (cached per axis:
    // a volume's extents do not change between frames), so the positions are the reference's by construction
    p.inv_w1 = 1.0f / p.w1;
    p.inv_h1 = 1.0f / p.h1;
    static const int pos_div_env = [] { const char* e = getenv("KFX_FUSE_POS_DIV"); return e ? atoi(e) : 1; }();
    p.pos_div = (pos_div_env && vol->w >= 2 && vol->h >= 2 && pos_div_verified(0, p.size.x, (int)vol->w) && pos_div_verified(1, p.size.y, (int)vol->h)) ? 1 : 0;
    static const int swizzle_env = [] { const char* e = getenv("KFX_FUSE_XCD_SWIZZLE"); const int v = e ? atoi(e) : 1; return v < 0 ? 0 : (v > 8 ? 8 : v); }();
    p.xcd_swizzle = swizzle_env;
    static const int cull_env = [] { const char* e = getenv("KFX_FUSE_CULL"); return e ? atoi(e) : 1; }();
    p.fuse_cull = cull_env;
    // launch-wide half of the operand-range test of the exact kernel's shared-reciprocal arithmetic (finish_shared);
    // KFX_FUSE_EXACT_SHARED=0 keeps hipcc's own division / square-root expansions (A/B, and the parity suite runs both)
    static const int shared_env = [] { const char* e = getenv("KFX_FUSE_EXACT_SHARED"); return e ? atoi(e) : 1; }();
    const float afu = fabsf(p.K.fu), afv = fabsf(p.K.fv);
    p.z_rev = 0; p.keep_z0 = 0; p.keep_z1 = 0;
    p.exact_shared = shared_env && afu >= 0x1p-20f && afu <= 0x1p20f && afv >= 0x1p-20f && afv <= 0x1p20f &&
                     mincostheta >= 0x1p-20f && mincostheta < __builtin_inff() && trunc_dist > 0.f && trunc_dist < __builtin_inff();
    return 0;
}

// Pixels per voxel, r = f * voxel / Z of camera (T, K), at the centre of local planes [z0, z1) (0 when that point is
// behind the camera)
static float px_per_voxel(const FuseParams& p, const Pose& T, const Intr& K, int z0, int z1)
{
    const float cx = p.bmin.x + 0.5f * p.size.x, cy = p.bmin.y + 0.5f * p.size.y;
    const float cz = p.bmin.z + p.size.z * (0.5f * (float)(z0 + z1 - 1) + (float)p.zoff) / p.d1;
    const float Zc = T.m[8] * cx + T.m[9] * cy + T.m[10] * cz + T.m[11];
    if (!(Zc > 0.f)) return 0.f;
    const float voxel = fmaxf(p.size.x / p.w1, p.size.y / p.h1);
    return fmaxf(fabsf(K.fu), fabsf(K.fv)) * voxel / Zc;
}

// LDS tile capacity (texels) of the 64 x 8 x 16 brick for the bricks of local planes [z0, z1): grows with r (see fuse_launch)
static int tile_cap(const FuseParams& p, const Pose& T, const Intr& K, int z0, int z1, bool small_ok = false)
{
    const float r = px_per_voxel(p, T, K, z0, z1);
    // far ranges (small_ok: the fast kernel): the rectangle, about (73 r + 5) x (14.7 r + 5) texels at the image corner, fits
    // 1216 texels (19 KiB: 8 workgroups per CU instead of 6, and the fast kernel's 64 VGPRs allow 8 waves per SIMD) --
    // 512^3, 640x480, frame loop: S_full (r <= 0.54) fast 0.351 -> 0.333 ms, S_room (r = 1.05 ... 0.58: planes beyond
    // 2.6 m) 0.286 -> 0.283 ms; with 1216 texels everywhere S_room loses (0.291 ms: the near bricks gather from global
    // memory).  The bit-exact kernel holds 78 VGPRs = 6 waves per SIMD whatever the tile and only pays for the extra launch
    // boundary (S_room 0.391 -> 0.405 ms), so it keeps 1536.
    static const float r_small = [] { const char* e = getenv("KFX_FUSE_R_SMALL"); return e ? (float)atof(e) : 0.85f; }();
    if (small_ok && !(r > r_small)) return 1216;
    if (!(r > 1.3f)) return 1536;
    const float want = 1536.f * (r / 1.05f) * (r / 1.05f);
    const int c = want >= 3072.f ? 3072 : ((int)want + 511) / 512 * 512;
    return c < 1536 ? 1536 : c;
}

// Brick geometry and tile capacity for local planes [z0, z1).  Up to r = 1.3 pixels per voxel: the 64 x 8 x 16 brick with
// 1536 texels (24 KiB, 6 workgroups per CU).  Beyond: the 32 x 8 x 16 brick, whose rectangle -- about r (32 + 0.56 * 16) + 5
// by r (8 + 0.42 * 16) + 5 texels for a brick at the edge of a 60 x 45 degree field of view -- is given the smallest of the
// capacities that leave 8 / 6 / 5 / 4 / 3 workgroups on a CU (160 KiB LDS, ~1 KiB of statics per workgroup).  Measured at
// 512^3, 1280x960, 2-4 m (r = 2.2 ... 1.1; scripts/c3_brick_ab.py, fast / exact): 64 x 8 x 16 everywhere 0.591 / 0.799 ms,
// 32 x 8 x 16 everywhere 0.537 / 0.732 ms, 32 x 8 x 8 0.586 / 0.749 ms (its staging and rectangle prologue are amortised
// over half the slices); at 640x480 (r <= 1.05) the narrow brick costs 0-3 %.
// dxt: the bit-exact kernel with {texel, x-difference} tiles (finish_shared_dx): 32 bytes per texel, so it is used where the
// rectangle -- about (73 r + 5) x (14.7 r + 5) texels -- fits 768 texels, the 24 KiB that keep six workgroups on a CU
// (r <= 0.64; KFX_FUSE_DXT=0 switches it off, a positive value sets the limit in hundredths).
struct TilePlan { int small_brick, cap, dxt; };
static TilePlan tile_plan(const FuseParams& p, const Pose& T, const Intr& K, int z0, int z1, int brick_env, bool fast)
{
    const float r = px_per_voxel(p, T, K, z0, z1);
    const int small_brick = brick_env < 0 ? (r > 1.3f ? 1 : 0) : (brick_env != 0);
    static const float r_dxt = [] { const char* e = getenv("KFX_FUSE_DXT"); return e ? 0.01f * (float)atoi(e) : 0.64f; }();
    if (!small_brick && !fast && r_dxt > 0.f && r > 0.f && !(r > r_dxt)) return TilePlan{0, 768, 1};
    if (!small_brick) return TilePlan{0, tile_cap(p, T, K, z0, z1, fast), 0};
    // three quarters of the worst-case rectangle: most bricks are nearer the optical axis than the image corner, and a
    // brick that does not fit still works (it gathers from global memory); measured at 1280x960 with one capacity for the
    // whole volume: 1984 texels 0.525 ms, 2496 0.543 ms, 3328 0.577 ms, per-range worst case 0.538 ms (fast mode)
    const float want = 0.75f * (r * 41.f + 5.f) * (r * 14.7f + 5.f);
    const int caps[] = {1216, 1600, 1984, 2496, 3328};
    for (int c : caps)
        if (want <= (float)c) return TilePlan{small_brick, c, 0};
    return TilePlan{small_brick, 3328, 0};
}

template <typename CELL>
static int fuse_launch(const kfx_volume* vol, const kfx_image* depth, const kfx_image* norm, const float T_cw[12],
                       const float K[4], float trunc_dist, float max_w, float mincostheta, unsigned flags, kfx_stream stream,
                       const kfx_slab* slab = nullptr, kfx_sdf_summary* summary = nullptr, const kfx_image* texels = nullptr)
{
    FuseParams p;
    bool small_images = false;
    if (int e = fuse_params(p, &small_images, vol, depth, norm, T_cw, K, trunc_dist, max_w, mincostheta, flags, CELL::BYTES, slab)) return e;
    p.sum_R = nullptr;
    p.zoff_local = 0;
    bool track = false;
    if (summary) {
        int ox, oy, oz;
        if (int e = summary_view_offset(summary, vol, &ox, &oy, &oz)) return e;
        track = (ox % 8 == 0) && (oy % 8 == 0) && (oz % 8 == 0);
        p.sum_R = summary->R;
        p.sum_nbx = summary->nbx; p.sum_nby = summary->nby;
        p.sum_bx0 = ox / 8; p.sum_by0 = oy / 8; p.sum_bz0 = oz / 8;
        p.sum_w = summary->w; p.sum_h = summary->h; p.sum_d = summary->d;
    }
    if (p.X == 0 || p.Y == 0 || p.Z == 0) return 0; // reference launches an empty grid
    // two cells per lane need an even extent and a pointer / pitches aligned to the cell pair
    const bool vec2 = (p.X % 2 == 0) && ((((uintptr_t)vol->ptr | vol->pitch | vol->img_pitch) & (2 * CELL::BYTES - 1)) == 0);
    const bool fast = math_mode() == KFX_MATH_FAST;
    hipStream_t s = (hipStream_t)stream;
    // tuning / A-B knobs (read once): KFX_FUSE_TILED=0 forces the global-gather kernel,
    // KFX_FUSE_CAP sets the LDS tile capacity in texels (16 B each, at most 3968)
    static const int tiled = [] { const char* e = getenv("KFX_FUSE_TILED"); return e ? atoi(e) : 1; }();
    static const int cap_env = [] { const char* e = getenv("KFX_FUSE_CAP"); const int v = e ? atoi(e) : 0; return v <= 0 ? 0 : (v < 64 ? 64 : (v > 3968 ? 3968 : v)); }();  // <= 62 KiB: dynamic + static LDS stay below the 64 KiB launch limit
    if (summary && !(track && tiled && vec2 && small_images && CELL::BYTES == 8)) {
        // this launch cannot keep the summary current (unaligned view, untiled kernel): nothing is known afterwards
        if (int e = kfx_sdf_summary_invalidate(summary, stream)) return e;
        track = false;
    }
    if (summary) summary->c_dirty = 1;
    if (tiled && vec2 && small_images) {
        // slices per iteration: 2 in fast mode (memory-bound: more reads in flight), 4 where the large LDS tile leaves
        // only 3 workgroups per CU (1280x960 at 512^3: 0.568 -> 0.538 ms; at 6 workgroups per CU 4 is slower), 1 in exact
        // mode (VALU-bound)
        static const int zu_env = [] { const char* e = getenv("KFX_FUSE_ZU"); return e ? atoi(e) : 0; }();
        // LDS tile capacity per z-range.  A brick's pixel rectangle grows with the pixels-per-voxel ratio
        // r = f * voxel / Z: 1536 texels (24 KiB, 6 workgroups per CU) hold it up to r ~ 1.3; beyond that more
        // bricks would fall back to global gathers, so the capacity grows with r^2 up to 3072 texels (48 KiB,
        // 3 workgroups per CU) -- measured at 512^3, 1280x960, 2-4 m: 0.80 ms (1536) / 0.62 ms (3072), while at
        // r < 1.2 the larger tile only costs occupancy (0.39 -> 0.51 ms).  r is evaluated at the centre of each
        // 64-slice range and ranges with equal capacity share a launch.
        // Beyond r = 1.3 the brick narrows to 32 x 8 x 16 voxels instead (tile_plan; KFX_FUSE_BRICK=0 / 1 forces the wide /
        // narrow brick): its rectangle fits 25-52 KiB up to r ~ 2.2, where the wide brick needs the full 48 KiB from
        // r ~ 1.5 on and above r ~ 1.9 does not fit at all (those bricks gathered from global memory).
        static const int brick_env = [] { const char* e = getenv("KFX_FUSE_BRICK"); return e ? atoi(e) : -1; }();
        auto plan_for = [&](int z0, int z1) -> TilePlan {
            TilePlan t = tile_plan(p, p.T, p.K, z0, z1, brick_env, fast);
            if (track || CELL::BYTES != 8) { // the difference tile exists for the untracked fp32-cell kernel only
                if (t.dxt) t = TilePlan{0, tile_cap(p, p.T, p.K, z0, z1, fast), 0};
            }
            if (cap_env) { t.cap = cap_env; t.dxt = 0; }
            return t;
        };
        const int zstep = 64;
        // the difference tile only where it costs no extra launch: every range of the view must want it (S_room at 512^3,
        // whose far third qualifies, lost 4 % to the third launch boundary; S_full gains 3 %)
        bool all_dxt = true;
        for (int z = 0; z < p.Z; z += zstep) all_dxt = all_dxt && plan_for(z, z + zstep < p.Z ? z + zstep : p.Z).dxt != 0;
        auto plan_of = [&](int za, int zb) -> TilePlan {
            TilePlan t = plan_for(za, zb);
            if (t.dxt && !all_dxt) { t = TilePlan{0, tile_cap(p, p.T, p.K, za, zb, fast), 0}; if (cap_env) t.cap = cap_env; }
            return t;
        };
        // Tracked launches sweep the planes in serpentine order -- every other launch from the far end -- and read the last
        // KFX_FUSE_KEEP_MB (default 256) of a sweep with ordinary loads: those planes stay in the 256 MiB memory-side cache
        // and the next sweep starts on them.  (The volume is otherwise streamed nontemporally, which leaves nothing behind;
        // untracked launches belong to loops whose plain march reads ~500 MB of the volume in between and evicts the tail.)
        static const int keep_mb = [] { const char* e = getenv("KFX_FUSE_KEEP_MB"); return e ? atoi(e) : 256; }();
        const int rev = (track && keep_mb > 0) ? (int)(summary->sweeps++ & 1u) : 0;
        int keep_lo = 0, keep_hi = 0;   // planes read with ordinary loads (this view's local coordinates)
        if (track && keep_mb > 0) {
            const size_t plane = (size_t)p.vimg_pitch;
            int n = (int)(((size_t)keep_mb << 20) / (plane ? plane : 1));
            n = n / FUSE_ZC * FUSE_ZC;
            if (n > p.Z) n = p.Z;
            if (rev) { keep_lo = 0; keep_hi = n; } else { keep_lo = p.Z - n; keep_hi = p.Z; }
        }
        struct Range { int z0, z1; TilePlan plan; };
        Range ranges[64];
        int n_ranges = 0;
        for (int z0 = 0; z0 < p.Z;) {
            int z1 = z0 + zstep < p.Z ? z0 + zstep : p.Z;
            const TilePlan plan = plan_of(z0, z1);
            while (z1 < p.Z) { // extend over following ranges that want the same brick and capacity
                const int z2 = z1 + zstep < p.Z ? z1 + zstep : p.Z;
                const TilePlan nxt = plan_of(z1, z2);
                if (nxt.cap != plan.cap || nxt.small_brick != plan.small_brick || nxt.dxt != plan.dxt) break;
                z1 = z2;
            }
            if (n_ranges == 64) { ranges[63].z1 = p.Z; break; } // (never: a range is at least 64 planes and plans change a few times at most)
            ranges[n_ranges++] = Range{z0, z1, plan};
            z0 = z1;
        }
        // The packed texel image the kernels stage by LDS-DMA (k_sdf_fuse_tiled): the caller's (kfx_frame_step: written by its fused
        // preprocess), else packed here into the library's scratch by one small launch (2-3 us at 640x480) -- unless every range
        // runs the difference-tile kernel, which stages through registers.
        bool any_plain = false;
        for (int ri = 0; ri < n_ranges; ++ri) any_plain = any_plain || !ranges[ri].plan.dxt;
        if (KFX_FUSE_STAGE_DMA && any_plain) {
            // (the caller's packed image -- kfx::texel_image_bytes(w, h) bytes laid out by tex_layout, written by the fused preprocess --
            //  is recognised by its geometry: w, h and pitch of the depth image's layout)
            const TexLayout tl = tex_layout(depth->w, depth->h);
            if (texels && texels->ptr && texels->w == depth->w && texels->h == depth->h && texels->pitch == tl.tpitch && !((uintptr_t)texels->ptr & 15)) {
                p.tex = (const unsigned char*)texels->ptr;
            } else {
                void* buf = tex_scratch(tl.bytes, s);   // (tpitch < 2^24: small_images bounds the normal map's pitch, which is at least as long)
                if (!buf) return set_error((int)hipErrorOutOfMemory, "SdfFuse: no device memory for the packed texel image");
                hipLaunchKernelGGL(k_pack_texels, dim3(ceil_div((int)depth->w, 64), ceil_div((int)depth->h, 4)), dim3(256), 0, s, p.depth, p.norm,
                                   (unsigned char*)buf, tl.tpitch, reinterpret_cast<float*>((unsigned char*)buf + tl.bmax_off), tl.bw8);
                p.tex = (const unsigned char*)buf;
            }
            p.tpitch = (unsigned)tl.tpitch;
            p.bmax = reinterpret_cast<const float*>(p.tex + tl.bmax_off);
            p.bw8 = tl.bw8;
        }
        for (int ri = 0; ri < n_ranges; ++ri) {
            const Range& rg = ranges[rev ? n_ranges - 1 - ri : ri];
            const int z0 = rg.z0, z1 = rg.z1;
            const TilePlan plan = rg.plan;
            const int cap_px = plan.cap;
            FuseParams q = p;
            q.z_rev = rev;
            q.keep_z0 = keep_lo - z0; q.keep_z1 = keep_hi - z0;
            q.vptr = p.vptr + (size_t)z0 * p.vimg_pitch;
            q.zoff = p.zoff + z0;
            q.zoff_local = z0;
            q.Z = z1 - z0;
            const size_t lds = (size_t)cap_px * sizeof(float4) * (plan.dxt ? 2 : 1);
            if constexpr (CELL::BYTES == 8) {
                if (track) { // the same kernels with the summary epilogue (ZU = 2 fast, 1 exact)
                    const dim3 gw(ceil_div(q.X, TB_X), ceil_div(q.Y, TB_Y), ceil_div(q.Z, FUSE_ZC)), gn(ceil_div(q.X, 32), ceil_div(q.Y, 8), ceil_div(q.Z, 16));
                    if (plan.small_brick && fast) hipLaunchKernelGGL((k_sdf_fuse_tiled<true, 2, CELL, 16, 2, 16, true>), gn, dim3(256), lds, s, q, cap_px);
                    else if (plan.small_brick) hipLaunchKernelGGL((k_sdf_fuse_tiled<false, 1, CELL, 16, 2, 16, true>), gn, dim3(256), lds, s, q, cap_px);
                    else if (fast) hipLaunchKernelGGL((k_sdf_fuse_tiled<true, 2, CELL, 32, 4, FUSE_ZC, true>), gw, dim3(256), lds, s, q, cap_px);
                    else hipLaunchKernelGGL((k_sdf_fuse_tiled<false, 1, CELL, 32, 4, FUSE_ZC, true>), gw, dim3(256), lds, s, q, cap_px);
                    continue;
                }
            }
            if (plan.small_brick) {
                dim3 grid(ceil_div(q.X, 32), ceil_div(q.Y, 8), ceil_div(q.Z, 16));
                const int zu = zu_env ? zu_env : (fast ? (cap_px > 2560 ? 4 : 2) : 1);
                // Tiles above 2048 texels (32 KiB: at most four workgroups on a CU) get eight waves per workgroup instead of four --
                // twice the waves behind one staged rectangle; C3 / S_room 0.4182 -> 0.4237 of peak, interleaved A/B; below that
                // size the shorter waves' prologues cost more than the residency buys (-1 %).  KFX_FUSE_NW8=<texels> moves the
                // threshold (0: never).  Same bits.
                static const int nw8_from = [] { const char* e = getenv("KFX_FUSE_NW8"); return e ? atoi(e) : 2048; }();
                if (fast && nw8_from > 0 && cap_px > nw8_from) {
                    hipLaunchKernelGGL((k_sdf_fuse_tiled<true, 2, CELL, 16, 2, 16, false, false, 8>), grid, dim3(512), lds, s, q, cap_px);
                    continue;
                }
                if (fast && zu == 4) hipLaunchKernelGGL((k_sdf_fuse_tiled<true, 4, CELL, 16, 2, 16>), grid, dim3(256), lds, s, q, cap_px);
                else if (fast && zu == 2) hipLaunchKernelGGL((k_sdf_fuse_tiled<true, 2, CELL, 16, 2, 16>), grid, dim3(256), lds, s, q, cap_px);
                else if (fast) hipLaunchKernelGGL((k_sdf_fuse_tiled<true, 1, CELL, 16, 2, 16>), grid, dim3(256), lds, s, q, cap_px);
                else hipLaunchKernelGGL((k_sdf_fuse_tiled<false, 1, CELL, 16, 2, 16>), grid, dim3(256), lds, s, q, cap_px);
            } else {
                dim3 grid(ceil_div(q.X, TB_X), ceil_div(q.Y, TB_Y), ceil_div(q.Z, FUSE_ZC));
                const int zu = zu_env ? zu_env : (fast ? (cap_px > 2560 ? 4 : 2) : 1);
                if (fast && zu == 4) hipLaunchKernelGGL((k_sdf_fuse_tiled<true, 4, CELL>), grid, dim3(256), lds, s, q, cap_px);
                else if (fast && zu == 2) hipLaunchKernelGGL((k_sdf_fuse_tiled<true, 2, CELL>), grid, dim3(256), lds, s, q, cap_px);
                else if (fast) hipLaunchKernelGGL((k_sdf_fuse_tiled<true, 1, CELL>), grid, dim3(256), lds, s, q, cap_px);
                else if (zu == 2) hipLaunchKernelGGL((k_sdf_fuse_tiled<false, 2, CELL>), grid, dim3(256), lds, s, q, cap_px);
                else if (plan.dxt) hipLaunchKernelGGL((k_sdf_fuse_tiled<false, 1, CELL, 32, 4, FUSE_ZC, false, true>), grid, dim3(256), lds, s, q, cap_px);
                else hipLaunchKernelGGL((k_sdf_fuse_tiled<false, 1, CELL>), grid, dim3(256), lds, s, q, cap_px);
            }
        }
    } else if (vec2) {
        dim3 grid(ceil_div(p.X, 128), ceil_div(p.Y, FUSE_ROWS), ceil_div(p.Z, FUSE_ZC));
        if (fast && small_images) hipLaunchKernelGGL((k_sdf_fuse<2, true, true, CELL>), grid, dim3(256), 0, s, p);
        else if (fast) hipLaunchKernelGGL((k_sdf_fuse<2, true, false, CELL>), grid, dim3(256), 0, s, p);
        else if (small_images) hipLaunchKernelGGL((k_sdf_fuse<2, false, true, CELL>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_sdf_fuse<2, false, false, CELL>), grid, dim3(256), 0, s, p);
    } else {
        dim3 grid(ceil_div(p.X, 64), ceil_div(p.Y, FUSE_ROWS), ceil_div(p.Z, FUSE_ZC));
        if (fast && small_images) hipLaunchKernelGGL((k_sdf_fuse<1, true, true, CELL>), grid, dim3(256), 0, s, p);
        else if (fast) hipLaunchKernelGGL((k_sdf_fuse<1, true, false, CELL>), grid, dim3(256), 0, s, p);
        else if (small_images) hipLaunchKernelGGL((k_sdf_fuse<1, false, true, CELL>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_sdf_fuse<1, false, false, CELL>), grid, dim3(256), 0, s, p);
    }
    return check_launch("kfx_sdf_fuse");
}

extern "C" int kfx_sdf_fuse(const kfx_volume* vol, const kfx_image* depth, const kfx_image* norm,
                            const float T_cw[12], const float K[4], float trunc_dist, float max_w,
                            float mincostheta, unsigned flags, kfx_stream stream)
{
    return fuse_launch<CellF32>(vol, depth, norm, T_cw, K, trunc_dist, max_w, mincostheta, flags, stream);
}

// kfx_sdf_fuse / kfx_sdf_fuse_tracked (summary != null) with the packed texel image brought by the caller (kfx_frame_step)
int kfx::sdf_fuse_texels(const kfx_volume* vol, kfx_sdf_summary* summary, const kfx_image* depth, const kfx_image* norm, const kfx_image* texels,
                         const float T_cw[12], const float K[4], float trunc_dist, float max_w, float mincostheta, unsigned flags, kfx_stream stream)
{
    return fuse_launch<CellF32>(vol, depth, norm, T_cw, K, trunc_dist, max_w, mincostheta, flags, stream, nullptr, summary, texels);
}

// kfx_sdf_fuse_slab with the packed texel image brought by the caller (kfx_slab_frame_step)
int kfx::sdf_fuse_slab_texels(const kfx_volume* vol, const kfx_slab* slab, const kfx_image* depth, const kfx_image* norm, const kfx_image* texels,
                              const float T_cw[12], const float K[4], float trunc_dist, float max_w, float mincostheta, unsigned flags, kfx_stream stream)
{
    if (!slab) return set_error(KFX_E_NULL, "kfx_sdf_fuse_slab: null slab");
    return fuse_launch<CellF32>(vol, depth, norm, T_cw, K, trunc_dist, max_w, mincostheta, flags, stream, slab, nullptr, texels);
}

extern "C" int kfx_sdf_fuse_tracked(const kfx_volume* vol, kfx_sdf_summary* summary, const kfx_image* depth, const kfx_image* norm,
                                    const float T_cw[12], const float K[4], float trunc_dist, float max_w, float mincostheta,
                                    unsigned flags, kfx_stream stream)
{
    if (!summary) return set_error(KFX_E_NULL, "kfx_sdf_fuse_tracked: null summary");
    return fuse_launch<CellF32>(vol, depth, norm, T_cw, K, trunc_dist, max_w, mincostheta, flags, stream, nullptr, summary);
}

extern "C" int kfx_sdf_fuse_slab(const kfx_volume* vol, const kfx_slab* slab, const kfx_image* depth, const kfx_image* norm,
                                 const float T_cw[12], const float K[4], float trunc_dist, float max_w,
                                 float mincostheta, unsigned flags, kfx_stream stream)
{
    if (!slab) return set_error(KFX_E_NULL, "kfx_sdf_fuse_slab: null slab");
    return fuse_launch<CellF32>(vol, depth, norm, T_cw, K, trunc_dist, max_w, mincostheta, flags, stream, slab);
}

extern "C" int kfx_sdf_fuse_slab_h(const kfx_volume* vol, const kfx_slab* slab, const kfx_image* depth, const kfx_image* norm,
                                   const float T_cw[12], const float K[4], float trunc_dist, float max_w,
                                   float mincostheta, unsigned flags, kfx_stream stream)
{
    if (!slab) return set_error(KFX_E_NULL, "kfx_sdf_fuse_slab_h: null slab");
    return fuse_launch<CellF16>(vol, depth, norm, T_cw, K, trunc_dist, max_w, mincostheta, flags, stream, slab);
}

extern "C" int kfx_sdf_fuse_h(const kfx_volume* vol, const kfx_image* depth, const kfx_image* norm,
                              const float T_cw[12], const float K[4], float trunc_dist, float max_w,
                              float mincostheta, unsigned flags, kfx_stream stream)
{
    return fuse_launch<CellF16>(vol, depth, norm, T_cw, K, trunc_dist, max_w, mincostheta, flags, stream);
}

extern "C" int kfx_sdf_fuse_count(const kfx_volume* vol, const kfx_image* depth, const kfx_image* norm,
                                  const float T_cw[12], const float K[4], float trunc_dist, float mincostheta,
                                  unsigned flags, unsigned long long* d_count, kfx_stream stream)
{
    if (!d_count) return set_error(KFX_E_NULL, "kfx_sdf_fuse_count: null counter");
    FuseParams p;
    bool small_images = false;
    if (int e = fuse_params(p, &small_images, vol, depth, norm, T_cw, K, trunc_dist, 0.f, mincostheta, flags)) return e;
    if (p.X == 0 || p.Y == 0 || p.Z == 0) return 0;
    const int bx = ceil_div(p.X, 64), by = ceil_div(p.Y, FUSE_ROWS), bz = ceil_div(p.Z, FUSE_ZC);
    const long long total = (long long)bx * by * bz;
    if (total > 0x7fffffffLL) return set_error(KFX_E_RANGE, "kfx_sdf_fuse_count: volume too large");
    dim3 grid((unsigned)(total < 4096 ? total : 4096));
    if (math_mode() == KFX_MATH_FAST) hipLaunchKernelGGL(k_sdf_fuse_count<true>, grid, dim3(256), 0, (hipStream_t)stream, p, bx, by, bz, d_count);
    else hipLaunchKernelGGL(k_sdf_fuse_count<false>, grid, dim3(256), 0, (hipStream_t)stream, p, bx, by, bz, d_count);
    return check_launch("kfx_sdf_fuse_count");
}

extern "C" int kfx_sdf_reset(const kfx_volume* vol, float trunc_dist, kfx_stream stream)
{
    if (int e = check_volume(vol)) return e;
    const size_t span_bytes = (vol->d - 1) * vol->img_pitch + (vol->h - 1) * vol->pitch + vol->w * 8;
    const size_t n = span_bytes / 8;
    const int blocks = (int)std::min<size_t>((n / 2 + 255) / 256 + 1, 256 * 32);
    hipStream_t s = (hipStream_t)stream;
    if (((uintptr_t)vol->ptr & 15) == 0)
        hipLaunchKernelGGL(k_fill_sdf, dim3((unsigned)((n / 2 + 1023) / 1024 + 1)), dim3(256), 0, s, (float2*)vol->ptr, n, trunc_dist, 0.0f);
    else
        hipLaunchKernelGGL(k_fill_sdf_unaligned, dim3(blocks), dim3(256), 0, s, (float2*)vol->ptr, n, trunc_dist, 0.0f);
    return check_launch("kfx_sdf_reset");
}

extern "C" int kfx_sdf_sphere(const kfx_volume* vol, const float center[3], float r, kfx_stream stream)
{
    if (int e = check_volume(vol)) return e;
    if (!center) return set_error(KFX_E_NULL, "SdfSphere: null center");
    const int X = (int)(vol->w / 8) * 8, Y = (int)(vol->h / 8) * 8, Z = (int)(vol->d / 8) * 8;
    if (X == 0 || Y == 0 || Z == 0) return 0;
    dim3 grid(ceil_div(X, 64), ceil_div(Y, 4), Z);
    hipLaunchKernelGGL(k_sdf_sphere<CellF32>, grid, dim3(256), 0, (hipStream_t)stream, vol_view(vol), X, Y, Z,
                       V3{center[0], center[1], center[2]}, r);
    return check_launch("kfx_sdf_sphere");
}

extern "C" int kfx_sdf_reset_h(const kfx_volume* vol, float trunc_dist, kfx_stream stream)
{
    if (int e = check_volume(vol, 4)) return e;
    const size_t span_bytes = (vol->d - 1) * vol->img_pitch + (vol->h - 1) * vol->pitch + vol->w * 4;
    const size_t n = span_bytes / 4;
    const unsigned pattern = (unsigned)__half_as_ushort(__float2half_rn(trunc_dist)); // {val = trunc, w = 0}
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_fill_u32, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned*)vol->ptr, n, pattern);
    return check_launch("kfx_sdf_reset_h");
}

extern "C" int kfx_sdf_sphere_h(const kfx_volume* vol, const float center[3], float r, kfx_stream stream)
{
    if (int e = check_volume(vol, 4)) return e;
    if (!center) return set_error(KFX_E_NULL, "SdfSphere: null center");
    const int X = (int)(vol->w / 8) * 8, Y = (int)(vol->h / 8) * 8, Z = (int)(vol->d / 8) * 8;
    if (X == 0 || Y == 0 || Z == 0) return 0;
    dim3 grid(ceil_div(X, 64), ceil_div(Y, 4), Z);
    hipLaunchKernelGGL(k_sdf_sphere<CellF16>, grid, dim3(256), 0, (hipStream_t)stream, vol_view(vol), X, Y, Z,
                       V3{center[0], center[1], center[2]}, r);
    return check_launch("kfx_sdf_sphere_h");
}

// SdfFuse(vol, colorVol, depth, norm, T_cw, K, img, T_iw, Kimg, trunc_dist, max_w, mincostheta) (cu_sdffusion.cu:120-138)
extern "C" int kfx_sdf_fuse_color(const kfx_volume* vol, const kfx_volume* colorvol, const kfx_image* depth, const kfx_image* norm,
                                  const float T_cw[12], const float K[4], const kfx_image* img, const float T_iw[12], const float Kimg[4],
                                  float trunc_dist, float max_w, float mincostheta, unsigned flags, kfx_stream stream)
{
    FuseParams p;
    bool small_images = false;
    if (int e = fuse_params(p, &small_images, vol, depth, norm, T_cw, K, trunc_dist, max_w, mincostheta, flags | KFX_FUSE_FULL_EXTENT)) return e;
    if (int e = check_volume(colorvol, 4)) return e;
    if (!img || !img->ptr || !T_iw || !Kimg) return set_error(KFX_E_NULL, "SdfFuse(colour): null argument");
    if (colorvol->w < vol->w || colorvol->h < vol->h || colorvol->d < vol->d) return set_error(KFX_E_SHAPE, "SdfFuse(colour): colour volume smaller than the SDF volume");
    if (img->w < 4 || img->h < 4 || img->pitch < img->w * 3) return set_error(KFX_E_SHAPE, "SdfFuse(colour): rgb image dimensions");
    if (!(flags & KFX_FUSE_FULL_EXTENT)) { // the reference's 16x16 launch over x / y, all of z (cu_sdffusion.cu:132-135)
        p.X = (int)(vol->w / 16) * 16;
        p.Y = (int)(vol->h / 16) * 16;
    }
    if (p.X == 0 || p.Y == 0 || p.Z == 0) return 0;
    ColorParams q;
    q.cptr = (unsigned char*)colorvol->ptr;
    q.cpitch = colorvol->pitch;
    q.cimg_pitch = colorvol->img_pitch;
    for (int i = 0; i < 12; ++i) q.Ti.m[i] = T_iw[i];
    q.Ki = Intr{Kimg[0], Kimg[1], Kimg[2], Kimg[3]};
    q.img = ImgView{(const unsigned char*)img->ptr, img->pitch, (int)img->w, (int)img->h};
    q.iwb = (float)img->w - 2.0f;
    q.ihb = (float)img->h - 2.0f;
    const bool fast = math_mode() == KFX_MATH_FAST;
    hipStream_t s = (hipStream_t)stream;
    static const int tiled = [] { const char* e = getenv("KFX_FUSE_TILED"); return e ? atoi(e) : 1; }();
    // two voxels per lane: even extent, 16-byte aligned SDF rows, 8-byte aligned colour rows
    const bool vec2 = (p.X % 2 == 0) && ((((uintptr_t)vol->ptr | vol->pitch | vol->img_pitch) & 15) == 0) &&
                      ((((uintptr_t)colorvol->ptr | colorvol->pitch | colorvol->img_pitch) & 7) == 0);
    if (tiled && vec2) {
        // per z-range LDS capacities as in fuse_launch: the depth / normal tile (16 B texels) by the depth camera's
        // pixels-per-voxel ratio, the RGB tile (4 B texels, a third more room) by the colour camera's
        auto rgb_cap = [&](int a, int b) { const int c = tile_cap(p, q.Ti, q.Ki, a, b) * 4 / 3; return c > 3584 ? 3584 : c; }; // 48 + 14 KiB + statics < 64 KiB
        const int zstep = 64;
        int z0 = 0;
        while (z0 < p.Z) {
            int z1 = z0 + zstep < p.Z ? z0 + zstep : p.Z;
            const int cap_px = tile_cap(p, p.T, p.K, z0, z1), cap_cpx = rgb_cap(z0, z1);
            while (z1 < p.Z) {
                const int z2 = z1 + zstep < p.Z ? z1 + zstep : p.Z;
                if (tile_cap(p, p.T, p.K, z1, z2) != cap_px || rgb_cap(z1, z2) != cap_cpx) break;
                z1 = z2;
            }
            FuseParams pp = p;
            ColorParams qq = q;
            pp.vptr = p.vptr + (size_t)z0 * p.vimg_pitch;
            qq.cptr = q.cptr + (size_t)z0 * q.cimg_pitch;
            pp.zoff = z0;
            pp.Z = z1 - z0;
            dim3 grid(ceil_div(pp.X, TB_X), ceil_div(pp.Y, TB_Y), ceil_div(pp.Z, FUSE_ZC));
            const size_t lds = (size_t)cap_px * sizeof(float4) + (size_t)cap_cpx * sizeof(unsigned);
            if (fast) hipLaunchKernelGGL(k_sdf_fuse_color_tiled<true>, grid, dim3(256), lds, s, pp, qq, cap_px, cap_cpx);
            else hipLaunchKernelGGL(k_sdf_fuse_color_tiled<false>, grid, dim3(256), lds, s, pp, qq, cap_px, cap_cpx);
            z0 = z1;
        }
    } else {
        dim3 grid(ceil_div(p.X, 64), ceil_div(p.Y, FUSE_ROWS), ceil_div(p.Z, FUSE_ZC));
        if (fast) hipLaunchKernelGGL(k_sdf_fuse_color<true>, grid, dim3(256), 0, s, p, q);
        else hipLaunchKernelGGL(k_sdf_fuse_color<false>, grid, dim3(256), 0, s, p, q);
    }
    return check_launch("kfx_sdf_fuse_color");
}

// SdfReset(BoundedVolume<float>) (cu_sdffusion.cu:166-169): every cell of the span, padding included, = 0.5
extern "C" int kfx_color_reset(const kfx_volume* colorvol, kfx_stream stream)
{
    if (int e = check_volume(colorvol, 4)) return e;
    const size_t n = ((colorvol->d - 1) * colorvol->img_pitch + (colorvol->h - 1) * colorvol->pitch + colorvol->w * 4) / 4;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_fill_f32, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)colorvol->ptr, n, 0.5f);
    return check_launch("kfx_color_reset");
}
