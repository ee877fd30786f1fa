// sampling.h -- the volume samplers shared by the ray-march (raycast.hip) and the mesh extraction (mesh.hip):
// BoundedVolume::GetUnitsTrilinearClamped and GetUnitsBackwardDiffDxDyDz (reference BoundedVolume.h:93-106 ->
// Volume.h:224-295) over the cell readers RayF32 / RayF16 / RayC32.  GEOM is any parameter block with the
// members {VolView vol; V3 size, dims1, hi2[, voxel]; V3 inv_size; int fastdiv, off32;} (RayParams, ColorGeom, MeshParams):
// inv_size / fastdiv enable the division shortcut of kfx_device.h (div_uniform), off32 says that every cell of the
// volume lies within 4 GiB of its base, so addresses are a uniform base plus a 32-bit lane offset (the saddr form of
// global_load: no 64-bit address arithmetic per lane).
#pragma once

#include <cstdlib>

#include <hip/hip_fp16.h>

#include "kfx_device.h"

namespace kfx {

// ---- cell readers: RayF32 = roo::SDF_t {float val; float w;}, RayF16 = roo::SDF_h {half val; half w;} ----
struct __attribute__((aligned(8))) Pair { float v0, w0, v1, w1; }; // fp32 cells x and x+1 of one row
struct RayF32 {
    static constexpr int BYTES = 8;
    // values of cells x and x+1 of the row starting at byte address `row`
    __device__ static __forceinline__ float2 pair(const unsigned char* row, int x)
    {
        const Pair c = *reinterpret_cast<const Pair*>(row + (size_t)x * 8);
        return make_float2(c.v0, c.v1);
    }
    __device__ static __forceinline__ float val(const unsigned char* row, int x) { return *reinterpret_cast<const float*>(row + (size_t)x * 8); }
    // The four row pairs of a trilinear sample as four 16-byte loads.  Written as asm because the compiler narrows a
    // 16-byte struct load to the two dwords that are used ({val, w, val, w}: the weights are not), which doubles
    // the number of requests the L1 (TCP) has to look up (rocprofv3, S_room: 96.6 M -> 38.4 M TCP accesses per launch,
    // 0.190 -> 0.173 ms).  o0..o3 are byte offsets from `base`.
    __device__ static __forceinline__ void pair4(const unsigned char* base, size_t o0, size_t o1, size_t o2, size_t o3,
                                                 float2& c0, float2& c1, float2& c2, float2& c3)
    {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 a, b, c, d;
        asm volatile("global_load_dwordx4 %0, %4, off\n\t"
                     "global_load_dwordx4 %1, %5, off\n\t"
                     "global_load_dwordx4 %2, %6, off\n\t"
                     "global_load_dwordx4 %3, %7, off\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                     : "v"(base + o0), "v"(base + o1), "v"(base + o2), "v"(base + o3)
                     : "memory");
        c0 = make_float2(a.x, a.z); c1 = make_float2(b.x, b.z); c2 = make_float2(c.x, c.z); c3 = make_float2(d.x, d.z);
    }
    // the same with 32-bit offsets from a wave-uniform base (SGPR pair)
    __device__ static __forceinline__ void pair4_off32(const unsigned char* base, unsigned o0, unsigned o1, unsigned o2, unsigned o3,
                                                       float2& c0, float2& c1, float2& c2, float2& c3)
    {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 a, b, c, d;
        // s_nop 4: `base` may have just been written by a VALU instruction (v_readlane of a spilled SGPR, v_readfirstlane);
        // a VMEM instruction that reads such an SGPR needs five wait states, and the compiler's hazard recogniser does not look
        // inside an asm block (found as a memory fault in k_raycast_sdf_levels_classes: the first load used a stale base)
        asm volatile("s_nop 4\n\t"
                     "global_load_dwordx4 %0, %4, %8\n\t"
                     "global_load_dwordx4 %1, %5, %8\n\t"
                     "global_load_dwordx4 %2, %6, %8\n\t"
                     "global_load_dwordx4 %3, %7, %8\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                     : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(base)
                     : "memory");
        c0 = make_float2(a.x, a.z); c1 = make_float2(b.x, b.z); c2 = make_float2(c.x, c.z); c3 = make_float2(d.x, d.z);
    }
    // The same loads split in two: issue() requests the four row pairs and returns at once, finish() waits for them.  Between
    // the two the caller may run code that touches neither global memory nor the four result vectors (the class-table march
    // runs its LDS lookups for the other lanes of the wave there): the compiler does not know the loads are in flight, the
    // "+v" operands of finish() keep the vectors allocated and untouched until the wait has passed.
    typedef float f4v __attribute__((ext_vector_type(4)));
    struct InFlight { f4v a, b, c, d; };
    __device__ static __forceinline__ void issue_off32(InFlight& f, const unsigned char* base, unsigned o0, unsigned o1, unsigned o2, unsigned o3)
    {
        asm volatile("s_nop 4\n\t"
                     "global_load_dwordx4 %0, %4, %8\n\t"
                     "global_load_dwordx4 %1, %5, %8\n\t"
                     "global_load_dwordx4 %2, %6, %8\n\t"
                     "global_load_dwordx4 %3, %7, %8"
                     : "=&v"(f.a), "=&v"(f.b), "=&v"(f.c), "=&v"(f.d)
                     : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(base)
                     : "memory");
    }
    __device__ static __forceinline__ void issue(InFlight& f, const unsigned char* base, size_t o0, size_t o1, size_t o2, size_t o3)
    {
        asm volatile("global_load_dwordx4 %0, %4, off\n\t"
                     "global_load_dwordx4 %1, %5, off\n\t"
                     "global_load_dwordx4 %2, %6, off\n\t"
                     "global_load_dwordx4 %3, %7, off"
                     : "=&v"(f.a), "=&v"(f.b), "=&v"(f.c), "=&v"(f.d)
                     : "v"(base + o0), "v"(base + o1), "v"(base + o2), "v"(base + o3)
                     : "memory");
    }
    __device__ static __forceinline__ void finish(InFlight& f, float2& c0, float2& c1, float2& c2, float2& c3)
    {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(f.a), "+v"(f.b), "+v"(f.c), "+v"(f.d) : : "memory");
        c0 = make_float2(f.a.x, f.a.z); c1 = make_float2(f.b.x, f.b.z); c2 = make_float2(f.c.x, f.c.z); c3 = make_float2(f.d.x, f.d.z);
    }
};
struct __attribute__((aligned(4))) PairH { unsigned a, b; };
struct RayF16 {
    static constexpr int BYTES = 4;
    __device__ static __forceinline__ float h(unsigned u) { return __half2float(__ushort_as_half((unsigned short)(u & 0xffffu))); }
    __device__ static __forceinline__ float2 pair(const unsigned char* row, int x)
    {
        const PairH c = *reinterpret_cast<const PairH*>(row + (size_t)x * 4);
        return make_float2(h(c.a), h(c.b));
    }
    __device__ static __forceinline__ float val(const unsigned char* row, int x) { return h(*reinterpret_cast<const unsigned*>(row + (size_t)x * 4)); }
    __device__ static __forceinline__ void pair4(const unsigned char* base, size_t o0, size_t o1, size_t o2, size_t o3,
                                                 float2& c0, float2& c1, float2& c2, float2& c3)
    {
        c0 = pair(base + o0, 0); c1 = pair(base + o1, 0); c2 = pair(base + o2, 0); c3 = pair(base + o3, 0);
    }
    __device__ static __forceinline__ void pair4_off32(const unsigned char* base, unsigned o0, unsigned o1, unsigned o2, unsigned o3,
                                                       float2& c0, float2& c1, float2& c2, float2& c3)
    {
        c0 = pair(base + o0, 0); c1 = pair(base + o1, 0); c2 = pair(base + o2, 0); c3 = pair(base + o3, 0);
    }
};
__device__ __forceinline__ const unsigned char* rowp(const VolView& v, int y, int z)
{
    return v.ptr + (size_t)z * v.img_pitch + (size_t)y * v.pitch;
}

// grey-level cells of a BoundedVolume<float> (colour raycast, cu_raycast.cu:119-189)
struct __attribute__((packed, aligned(4))) PairC { float a, b; };
struct RayC32 {
    static constexpr int BYTES = 4;
    __device__ static __forceinline__ float2 pair(const unsigned char* row, int x)
    {
        const PairC c = *reinterpret_cast<const PairC*>(row + (size_t)x * 4);
        return make_float2(c.a, c.b);
    }
    __device__ static __forceinline__ float val(const unsigned char* row, int x) { return *reinterpret_cast<const float*>(row + (size_t)x * 4); }
    __device__ static __forceinline__ void pair4(const unsigned char* base, size_t o0, size_t o1, size_t o2, size_t o3,
                                                 float2& c0, float2& c1, float2& c2, float2& c3)
    {
        c0 = pair(base + o0, 0); c1 = pair(base + o1, 0); c2 = pair(base + o2, 0); c3 = pair(base + o3, 0);
    }
    __device__ static __forceinline__ void pair4_off32(const unsigned char* base, unsigned o0, unsigned o1, unsigned o2, unsigned o3,
                                                       float2& c0, float2& c1, float2& c2, float2& c3)
    {
        c0 = pair(base + o0, 0); c1 = pair(base + o1, 0); c2 = pair(base + o2, 0); c3 = pair(base + o3, 0);
    }
};
// geometry of a second volume sampled with trilinear<>() (same member names as RayParams)
struct ColorGeom {
    VolView vol;
    V3 size, dims1, hi2;
    V3 inv_size;
    int fastdiv, off32;
};

// BoundedVolume::GetUnitsTrilinearClamped -> Volume::GetFractionalTrilinearClamped
// (pos_w - bbox.Min()) / bbox.Size(): the division by the launch-uniform box size through div_uniform when the
// operands are in its range (bit-identical quotient), the hardware division otherwise
template <typename GEOM>
__device__ __forceinline__ V3 box_fraction(const GEOM& p, const V3 pos_w)
{
    const V3 d = pos_w - p.vol.bmin;
    const float hi = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z)), lo = fminf(fminf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
    // Both paths give the IEEE quotients, so the choice may be made per WAVE: one scalar branch instead of a divergent one
    // (exec-mask bookkeeping for a path that is taken once in a million steps: a component within 2^-40 of the box's face).
    if (__builtin_expect(p.fastdiv && __ballot(!(hi < 0x1p40f && lo > 0x1p-40f)) == 0ull, 1))
        return v3(div_uniform(d.x, p.size.x, p.inv_size.x), div_uniform(d.y, p.size.y, p.inv_size.y), div_uniform(d.z, p.size.z, p.inv_size.z));
    return div_cw(d, p.size);
}

// base cell (clamped to [0, dim - 2], Volume.h:229-234) and unclamped fractions of a trilinear sample
struct CellPos { int ix, iy, iz; float fx, fy, fz; };
template <typename GEOM>
__device__ __forceinline__ CellPos cell_of(const GEOM& p, const V3 pos_w)
{
    const V3 pos_v = box_fraction(p, pos_w);
    const V3 pf = v3(pos_v.x * p.dims1.x, pos_v.y * p.dims1.y, pos_v.z * p.dims1.z);
    // max(min(dim - 2, floor(pf)), 0) as one v_med3_f32 per axis (0 <= dim - 2: the median IS the clamp; for a NaN coordinate
    // the median picks 0 where the reference's fminf / fmaxf pick dim - 2, and the sample is NaN either way: its fraction is),
    // and the clamped value, an integer below 2^24, serves as (float)ix: no conversion back
    const float cx = __builtin_amdgcn_fmed3f(floorf(pf.x), 0.f, p.hi2.x), cy = __builtin_amdgcn_fmed3f(floorf(pf.y), 0.f, p.hi2.y),
                cz = __builtin_amdgcn_fmed3f(floorf(pf.z), 0.f, p.hi2.z);
    CellPos c;
    c.ix = (int)cx; c.iy = (int)cy; c.iz = (int)cz;
    c.fx = pf.x - cx; c.fy = pf.y - cy; c.fz = pf.z - cz;
    return c;
}

// the march's step for a sample (cu_raycast.cu:77-80): max(sdf, min_delta) in front of the surface, trunc behind it or for a
// NaN sample.  fmaxf() costs a canonicalising v_max per operand; with sdf > 0 already known a compare-and-select is the same value.
__device__ __forceinline__ float march_step(const float sdf, const float min_delta, const float trunc)
{
    return sdf > 0 ? (min_delta > sdf ? min_delta : sdf) : trunc;
}

// the eight cells of the sample and their blend (Volume.h:236-250)
template <typename CELL, typename GEOM>
__device__ __forceinline__ float trilinear_at(const GEOM& p, const CellPos& c)
{
    const int ix = c.ix, iy = c.iy, iz = c.iz;
    const float fx = c.fx, fy = c.fy, fz = c.fz;
    float2 c00, c10, c01, c11;
    if (p.off32) { // launch-uniform
        const unsigned pitch = (unsigned)p.vol.pitch, img = (unsigned)p.vol.img_pitch;
        const unsigned o = (unsigned)iz * img + (unsigned)iy * pitch + (unsigned)ix * CELL::BYTES;
        CELL::pair4_off32(p.vol.ptr, o, o + pitch, o + img, o + img + pitch, c00, c10, c01, c11);
    } else {
        const size_t o = (size_t)iz * p.vol.img_pitch + (size_t)iy * p.vol.pitch + (size_t)ix * CELL::BYTES;
        CELL::pair4(p.vol.ptr, o, o + p.vol.pitch, o + p.vol.img_pitch, o + p.vol.img_pitch + p.vol.pitch, c00, c10, c01, c11);
    }
    return lerp(lerp(lerp(c00.x, c00.y, fx), lerp(c10.x, c10.y, fx), fy),
                lerp(lerp(c01.x, c01.y, fx), lerp(c11.x, c11.y, fx), fy), fz);
}

// trilinear_at() in two halves (fp32 cells): request the eight cells, and -- later -- wait for them and blend
template <typename GEOM>
__device__ __forceinline__ void trilinear_issue(RayF32::InFlight& f, const GEOM& p, const CellPos& c)
{
    if (p.off32) { // launch-uniform
        const unsigned pitch = (unsigned)p.vol.pitch, img = (unsigned)p.vol.img_pitch;
        const unsigned o = (unsigned)c.iz * img + (unsigned)c.iy * pitch + (unsigned)c.ix * 8u;
        RayF32::issue_off32(f, p.vol.ptr, o, o + pitch, o + img, o + img + pitch);
    } else {
        const size_t o = (size_t)c.iz * p.vol.img_pitch + (size_t)c.iy * p.vol.pitch + (size_t)c.ix * 8;
        RayF32::issue(f, p.vol.ptr, o, o + p.vol.pitch, o + p.vol.img_pitch, o + p.vol.img_pitch + p.vol.pitch);
    }
}
__device__ __forceinline__ float trilinear_finish(RayF32::InFlight& f, const CellPos& c)
{
    float2 c00, c10, c01, c11;
    RayF32::finish(f, c00, c10, c01, c11);
    return lerp(lerp(lerp(c00.x, c00.y, c.fx), lerp(c10.x, c10.y, c.fx), c.fy),
                lerp(lerp(c01.x, c01.y, c.fx), lerp(c11.x, c11.y, c.fx), c.fy), c.fz);
}

template <typename CELL, typename GEOM>
__device__ __forceinline__ float trilinear(const GEOM& p, const V3 pos_w)
{
    return trilinear_at<CELL>(p, cell_of(p, pos_w));
}

// BoundedVolume::GetUnitsBackwardDiffDxDyDz -> Volume::GetFractionalBackwardDiffDxDyDz.
// Corner (cx,cy,cz) gradient = v(c) - v(c - e_axis); the 8 corners need the 20 cells of
// {-1,0,1}^3 (relative to the clamped base) that have at most one coordinate equal to -1.
template <typename CELL, typename GEOM>
__device__ __forceinline__ V3 gradient(const GEOM& p, const V3 pos_w)
{
    const V3 pos_v = div_cw(pos_w - p.vol.bmin, p.size);
    const V3 pf = v3(pos_v.x * p.dims1.x, pos_v.y * p.dims1.y, pos_v.z * p.dims1.z);
    const int ix = (int)fmaxf(fminf(p.hi2.x, floorf(pf.x)), 1.f);
    const int iy = (int)fmaxf(fminf(p.hi2.y, floorf(pf.y)), 1.f);
    const int iz = (int)fmaxf(fminf(p.hi2.z, floorf(pf.z)), 1.f);
    const float fx = pf.x - (float)ix, fy = pf.y - (float)iy, fz = pf.z - (float)iz;
    const VolView& v = p.vol;
    // c[dz][dy][dx] for dx,dy,dz in {0,1}; mx/my/mz = the cells one step back along x/y/z.
    float c[2][2][2], mx[2][2], my[2][2], mz[2][2];
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const unsigned char* r = rowp(v, iy + dy, iz + dz);
            mx[dz][dy] = CELL::val(r, ix - 1);
            const float2 cc = CELL::pair(r, ix);
            c[dz][dy][0] = cc.x;
            c[dz][dy][1] = cc.y;
        }
#pragma unroll
    for (int dz = 0; dz < 2; ++dz) {
        const float2 cc = CELL::pair(rowp(v, iy - 1, iz + dz), ix);
        my[dz][0] = cc.x;
        my[dz][1] = cc.y;
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const float2 cc = CELL::pair(rowp(v, iy + dy, iz - 1), ix);
        mz[dy][0] = cc.x;
        mz[dy][1] = cc.y;
    }
    V3 g[2][2][2];
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const float v0 = c[dz][dy][dx];
                const float bx = dx ? c[dz][dy][0] : mx[dz][dy];
                const float by = dy ? c[dz][0][dx] : my[dz][dx];
                const float bz = dz ? c[0][dy][dx] : mz[dy][dx];
                g[dz][dy][dx] = v3(v0 - bx, v0 - by, v0 - bz);
            }
    const V3 deriv = lerp(lerp(lerp(g[0][0][0], g[0][0][1], fx), lerp(g[0][1][0], g[0][1][1], fx), fy),
                          lerp(lerp(g[1][0][0], g[1][0][1], fx), lerp(g[1][1][0], g[1][1][1], fx), fy), fz);
    return div_cw(deriv, p.voxel);
}

// Host side: the arithmetic / addressing shortcuts of a geometry block whose vol / size members are set (g.vol.d planes
// addressed from g.vol.ptr, which may be a virtual base: raycast_slab_launch)
template <typename GEOM>
inline void set_shortcuts(GEOM& g)
{
    g.inv_size = V3{1.0f / g.size.x, 1.0f / g.size.y, 1.0f / g.size.z};
    g.fastdiv = div_uniform_safe_host(g.size.x) && div_uniform_safe_host(g.size.y) && div_uniform_safe_host(g.size.z);
    // last byte a sampler can touch, relative to the base, below 4 GiB
    const double span = (double)(g.vol.d - 1) * (double)g.vol.img_pitch + (double)(g.vol.h - 1) * (double)g.vol.pitch + (double)g.vol.w * 16.0;
    // KFX_SAMPLER_SHORTCUTS=0 switches both shortcuts off (hardware division, 64-bit addresses: the paths volumes above
    // 4 GiB and out-of-range boxes take), so that the parity tests can run through them at small sizes
    static const int enabled = [] { const char* e = getenv("KFX_SAMPLER_SHORTCUTS"); return e ? atoi(e) : 1; }();
    g.off32 = enabled && span < 4294967296.0;
    g.fastdiv = enabled && g.fastdiv;
}

// Host side: fill the members the samplers read ({vol, size, dims1, hi2}) from a kfx_volume.  VoxelSizeUnits =
// Size / (dims - 1) with the dims converted from size_t (BoundedVolume.h:67-76) goes into `voxel` where the block has one.
template <typename GEOM>
inline void set_geometry(GEOM& g, const kfx_volume* v)
{
    g.vol.ptr = (unsigned char*)v->ptr;
    g.vol.pitch = v->pitch;
    g.vol.img_pitch = v->img_pitch;
    g.vol.w = (int)v->w;
    g.vol.h = (int)v->h;
    g.vol.d = (int)v->d;
    g.vol.bmin = V3{v->boxmin[0], v->boxmin[1], v->boxmin[2]};
    g.vol.bmax = V3{v->boxmax[0], v->boxmax[1], v->boxmax[2]};
    g.size = V3{v->boxmax[0] - v->boxmin[0], v->boxmax[1] - v->boxmin[1], v->boxmax[2] - v->boxmin[2]};
    g.dims1 = V3{(float)v->w - 1.f, (float)v->h - 1.f, (float)v->d - 1.f};
    g.hi2 = V3{(float)(v->w - 2), (float)(v->h - 2), (float)(v->d - 2)};
    set_shortcuts(g);
}
template <typename GEOM>
inline void set_voxel_size(GEOM& g, const kfx_volume* v)
{
    g.voxel = V3{g.size.x / (float)(v->w - 1), g.size.y / (float)(v->h - 1), g.size.z / (float)(v->d - 1)};
}

} // namespace kfx
