// Volume.h -- pitched 3-D volume view / owner, roo::Volume<T, Target, Management>.
//
// Interface and layout of the reference's include/kangaroo/Volume.h:21-370:
// {size_t pitch; T* ptr; size_t w; size_t h; size_t img_pitch; size_t d;} = 48 bytes; x fastest,
// rows `pitch` bytes apart, z-slices `img_pitch` bytes apart.  Trilinear sampling clamps only the
// integer cell, never the fraction (quirk Q4); the gradient is the trilinear blend of the eight
// corners' backward differences with the base cell clamped to [1, dim-2].
#pragma once

#include <cassert>

#include <kangaroo/Image.h>

namespace roo
{

template<typename T, typename Target = TargetDevice, typename Management = DontManage>
struct Volume
{
    KANGAROO_HD ~Volume() { Management::template Cleanup<T, Target>(ptr); }

    template<typename TargetFrom, typename ManagementFrom>
    KANGAROO_HD Volume(const Volume<T, TargetFrom, ManagementFrom>& v)
        : pitch(v.pitch), ptr(v.ptr), w(v.w), h(v.h), img_pitch(v.img_pitch), d(v.d)
    {
        AssignmentCheck<Management, Target, TargetFrom>();
    }
    Volume() : pitch(0), ptr(0), w(0), h(0), img_pitch(0), d(0) {}
    Volume(unsigned int width, unsigned int height, unsigned int depth) : w(width), h(height), d(depth)
    {
        Management::AllocateCheck();
        Target::template AllocatePitchedMem<T>(&ptr, &pitch, &img_pitch, w, h, d);
    }
    KANGAROO_HD Volume(T* p, size_t width, size_t height, size_t depth)
        : pitch(sizeof(T) * width), ptr(p), w(width), h(height), img_pitch(sizeof(T) * width * height), d(depth) {}
    KANGAROO_HD Volume(T* p, size_t width, size_t height, size_t depth, size_t pitch_bytes)
        : pitch(pitch_bytes), ptr(p), w(width), h(height), img_pitch(pitch_bytes * height), d(depth) {}
    KANGAROO_HD Volume(T* p, size_t width, size_t height, size_t depth, size_t pitch_bytes, size_t img_pitch_bytes)
        : pitch(pitch_bytes), ptr(p), w(width), h(height), img_pitch(img_pitch_bytes), d(depth) {}

    // ---- copies: one 2-D copy of h*d rows, as the reference does (Volume.h:83-93) ----
    template<typename TargetFrom, typename ManagementFrom>
    void CopyFrom(const Volume<T, TargetFrom, ManagementFrom>& v)
    {
        assert(w == v.w && h == v.h);
        assert(img_pitch == pitch * h && v.img_pitch == v.pitch * v.h);
        const int st = kfx_memcpy_2d(ptr, pitch, v.ptr, v.pitch, std::min(v.w, w) * sizeof(T), h * std::min(v.d, d),
                                     TargetCopyKind<Target, TargetFrom>(), 0);
        if (st != 0) throw HipException("Unable to copy volume", st);
    }
    template<typename DT> void MemcpyFromHost(DT* hptr, size_t hpitch)
    {
        const int st = kfx_memcpy_2d((void*)ptr, pitch, hptr, hpitch, w * sizeof(T), h * d, 1, 0);
        if (st != 0) throw HipException("Unable to copy volume from host", st);
    }
    template<typename DT> void MemcpyFromHost(DT* hptr) { MemcpyFromHost(hptr, w * sizeof(T)); }

    // ---- element access --------------------------------------------------------------
    KANGAROO_HD T* ImagePtr(size_t z) { return (T*)((unsigned char*)(ptr) + z * img_pitch); }
    KANGAROO_HD const T* ImagePtr(size_t z) const { return (const T*)((const unsigned char*)(ptr) + z * img_pitch); }
    KANGAROO_HD T* RowPtr(size_t y, size_t z) { return (T*)((unsigned char*)(ptr) + z * img_pitch + y * pitch); }
    KANGAROO_HD const T* RowPtr(size_t y, size_t z) const { return (const T*)((const unsigned char*)(ptr) + z * img_pitch + y * pitch); }
    KANGAROO_HD T& operator()(size_t x, size_t y, size_t z) { return RowPtr(y, z)[x]; }
    KANGAROO_HD const T& operator()(size_t x, size_t y, size_t z) const { return RowPtr(y, z)[x]; }
    KANGAROO_HD T& operator[](size_t ix) { return ptr[ix]; }
    KANGAROO_HD const T& operator[](size_t ix) const { return ptr[ix]; }
    KANGAROO_HD T& Get(int x, int y, int z) { return RowPtr(y, z)[x]; }
    KANGAROO_HD const T& Get(int x, int y, int z) const { return RowPtr(y, z)[x]; }
    KANGAROO_HD T& Get(int3 p) { return RowPtr(p.y, p.z)[p.x]; }
    KANGAROO_HD const T& Get(int3 p) const { return RowPtr(p.y, p.z)[p.x]; }

    // ---- interpolated access, `pos` in [0,1]^3 over the voxel grid ---------------------
    KANGAROO_HD float GetFractionalTrilinearClamped(float3 pos) const
    {
        const float3 pf = make_float3(pos.x * (w - 1.f), pos.y * (h - 1.f), pos.z * (d - 1.f));
        const int ix = fmaxf(fminf(w - 2, floorf(pf.x)), 0);
        const int iy = fmaxf(fminf(h - 2, floorf(pf.y)), 0);
        const int iz = fmaxf(fminf(d - 2, floorf(pf.z)), 0);
        const float fx = pf.x - ix, fy = pf.y - iy, fz = pf.z - iz;
        const float v000 = Get(ix, iy, iz), v100 = Get(ix + 1, iy, iz), v010 = Get(ix, iy + 1, iz), v110 = Get(ix + 1, iy + 1, iz);
        const float v001 = Get(ix, iy, iz + 1), v101 = Get(ix + 1, iy, iz + 1), v011 = Get(ix, iy + 1, iz + 1), v111 = Get(ix + 1, iy + 1, iz + 1);
        return lerp(lerp(lerp(v000, v100, fx), lerp(v010, v110, fx), fy), lerp(lerp(v001, v101, fx), lerp(v011, v111, fx), fy), fz);
    }

    KANGAROO_HD float3 GetBackwardDiffDxDyDz(int x, int y, int z) const
    {
        const float v0 = Get(x, y, z);
        return make_float3(v0 - Get(x - 1, y, z), v0 - Get(x, y - 1, z), v0 - Get(x, y, z - 1));
    }

    KANGAROO_HD float3 GetFractionalBackwardDiffDxDyDz(float3 pos) const
    {
        const float3 pf = make_float3(pos.x * (w - 1.f), pos.y * (h - 1.f), pos.z * (d - 1.f));
        const int ix = fmaxf(fminf(w - 2, floorf(pf.x)), 1);
        const int iy = fmaxf(fminf(h - 2, floorf(pf.y)), 1);
        const int iz = fmaxf(fminf(d - 2, floorf(pf.z)), 1);
        const float fx = pf.x - ix, fy = pf.y - iy, fz = pf.z - iz;
        float3 g[2][2][2];
        for (int k = 0; k < 2; ++k)
            for (int j = 0; j < 2; ++j)
                for (int i = 0; i < 2; ++i) g[k][j][i] = GetBackwardDiffDxDyDz(ix + i, iy + j, iz + k);
        return lerp(lerp(lerp(g[0][0][0], g[0][0][1], fx), lerp(g[0][1][0], g[0][1][1], fx), fy),
                    lerp(lerp(g[1][0][0], g[1][0][1], fx), lerp(g[1][1][0], g[1][1][1], fx), fy), fz);
    }

    // ---- sub-views -----------------------------------------------------------------------
    KANGAROO_HD Volume<T, Target, DontManage> SubVolume(int3 start, int3 size)
    {
        return Volume<T, Target, DontManage>(&Get(start), size.x, size.y, size.z, pitch, img_pitch);
    }
    KANGAROO_HD Image<T, Target, DontManage> ImageXY(size_t z)
    {
        assert(z < d);
        return Image<T, Target, DontManage>(ImagePtr(z), w, h, pitch);
    }
    KANGAROO_HD Image<T, Target, DontManage> ImageXZ(size_t y)
    {
        assert(y < h);
        return Image<T, Target, DontManage>(RowPtr(y, 0), w, d, img_pitch);
    }
    KANGAROO_HD uint3 Voxels() const { return make_uint3(w, h, d); }

    // ---- members (public, as in the reference) ----------------------------------------------
    size_t pitch;
    T* ptr;
    size_t w;
    size_t h;
    size_t img_pitch;
    size_t d;
};

}
