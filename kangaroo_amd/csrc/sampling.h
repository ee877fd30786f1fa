// sampling.h -- the volume samplers shared by the ray-march (raycast.hip) and the mesh extraction (mesh.hip):
// BoundedVolume::GetUnitsTrilinearClamped and GetUnitsBackwardDiffDxDyDz (reference BoundedVolume.h:93-106 ->
// Volume.h:224-295) over the cell readers RayF32 / RayF16 / RayC32.  GEOM is any parameter block with the
// members {VolView vol; V3 size, dims1, hi2[, voxel];} (RayParams, ColorGeom, MeshParams).
#pragma once

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
};
// geometry of a second volume sampled with trilinear<>() (same member names as RayParams)
struct ColorGeom {
    VolView vol;
    V3 size, dims1, hi2;
};

// BoundedVolume::GetUnitsTrilinearClamped -> Volume::GetFractionalTrilinearClamped
template <typename CELL, typename GEOM>
__device__ __forceinline__ float trilinear(const GEOM& p, const V3 pos_w)
{
    const V3 pos_v = div_cw(pos_w - p.vol.bmin, p.size);
    const V3 pf = v3(pos_v.x * p.dims1.x, pos_v.y * p.dims1.y, pos_v.z * p.dims1.z);
    const int ix = (int)fmaxf(fminf(p.hi2.x, floorf(pf.x)), 0.f);
    const int iy = (int)fmaxf(fminf(p.hi2.y, floorf(pf.y)), 0.f);
    const int iz = (int)fmaxf(fminf(p.hi2.z, floorf(pf.z)), 0.f);
    const float fx = pf.x - (float)ix, fy = pf.y - (float)iy, fz = pf.z - (float)iz;
    const unsigned char* b = rowp(p.vol, iy, iz);
    const float2 c00 = CELL::pair(b, ix);
    const float2 c10 = CELL::pair(b + p.vol.pitch, ix);
    const float2 c01 = CELL::pair(b + p.vol.img_pitch, ix);
    const float2 c11 = CELL::pair(b + p.vol.img_pitch + p.vol.pitch, ix);
    return lerp(lerp(lerp(c00.x, c00.y, fx), lerp(c10.x, c10.y, fx), fy),
                lerp(lerp(c01.x, c01.y, fx), lerp(c11.x, c11.y, fx), fy), fz);
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
}
template <typename GEOM>
inline void set_voxel_size(GEOM& g, const kfx_volume* v)
{
    g.voxel = V3{g.size.x / (float)(v->w - 1), g.size.y / (float)(v->h - 1), g.size.z / (float)(v->d - 1)};
}

} // namespace kfx
