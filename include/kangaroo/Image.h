// Image.h -- pitched 2-D image view / owner, roo::Image<T, Target, Management>.
//
// Interface and memory layout of the reference's include/kangaroo/Image.h:43-621:
// {size_t pitch; T* ptr; size_t w; size_t h;} = 32 bytes, byte pitch, public members, non-owning
// by default.  An Image is handed to the C ABI (include/kfx.h) by address: kfx_image has the same
// layout (see Image::abi()).  Copies go through kfx_memcpy_2d.
#pragma once

#include <algorithm>
#include <cassert>
#include <cstddef>

#include <kfx.h>
#include <kangaroo/Memory.h>
#include <kangaroo/VecMath.h>

namespace roo
{

template<typename T, typename Target = TargetDevice, typename Management = DontManage>
struct Image
{
    KANGAROO_HD ~Image() { Management::template Cleanup<T, Target>(ptr); }

    // ---- construction ----------------------------------------------------------
    KANGAROO_HD Image(const Image<T, Target, Management>& img) : pitch(img.pitch), ptr(img.ptr), w(img.w), h(img.h)
    {
        AssignmentCheck<Management, Target, Target>();
    }
    // view of another container of the same memory space (owner -> DontManage view)
    template<typename TargetFrom, typename ManagementFrom>
    KANGAROO_HD Image(const Image<T, TargetFrom, ManagementFrom>& img) : pitch(img.pitch), ptr(img.ptr), w(img.w), h(img.h)
    {
        AssignmentCheck<Management, Target, TargetFrom>();
    }
    Image() : pitch(0), ptr(0), w(0), h(0) {}
    // allocating constructor: only valid with Management = Manage
    Image(unsigned int width, unsigned int height) : w(width), h(height)
    {
        Management::AllocateCheck();
        Target::template AllocatePitchedMem<T>(&ptr, &pitch, w, h);
    }
    KANGAROO_HD Image(T* p) : pitch(0), ptr(p), w(0), h(0) {}
    KANGAROO_HD Image(T* p, size_t width) : pitch(sizeof(T) * width), ptr(p), w(width), h(0) {}
    KANGAROO_HD Image(T* p, size_t width, size_t height) : pitch(sizeof(T) * width), ptr(p), w(width), h(height) {}
    KANGAROO_HD Image(T* p, size_t width, size_t height, size_t pitch_bytes) : pitch(pitch_bytes), ptr(p), w(width), h(height) {}

    Image(Image<T, Target, Management>&& img) : pitch(img.pitch), ptr(img.ptr), w(img.w), h(img.h) { img.ptr = 0; }
    void operator=(Image<T, Target, Management>&& img)
    {
        pitch = img.pitch; ptr = img.ptr; w = img.w; h = img.h;
        img.ptr = 0;
    }

    // ---- the C-ABI view (same 32 bytes) ---------------------------------------------
    const kfx_image* abi() const
    {
        static_assert(sizeof(Image<T, Target, Management>) == sizeof(kfx_image), "roo::Image must mirror kfx_image");
        return reinterpret_cast<const kfx_image*>(this);
    }

    // ---- dimensions ------------------------------------------------------------------
    KANGAROO_HD size_t Width() const { return w; }
    KANGAROO_HD size_t Height() const { return h; }
    KANGAROO_HD size_t Area() const { return w * h; }
    KANGAROO_HD bool IsValid() const { return ptr != 0; }

    // ---- copies (blocking, default stream -- like cudaMemcpy2D in the reference) -----
    template<typename TargetFrom, typename ManagementFrom>
    void CopyFrom(const Image<T, TargetFrom, ManagementFrom>& img)
    {
        const int st = kfx_memcpy_2d(ptr, pitch, img.ptr, img.pitch, std::min(img.w, w) * sizeof(T), std::min(img.h, h),
                                     TargetCopyKind<Target, TargetFrom>(), 0);
        if (st != 0) throw HipException("Unable to copy image", st);
    }
    template<typename DT> void MemcpyFromHost(DT* hptr, size_t hpitch)
    {
        const int st = kfx_memcpy_2d((void*)ptr, pitch, hptr, hpitch, w * sizeof(T), h, TargetCopyKind<Target, TargetHost>(), 0);
        if (st != 0) throw HipException("Unable to copy in MemcpyFromHost", st);
    }
    template<typename DT> void MemcpyFromHost(DT* hptr) { MemcpyFromHost(hptr, w * sizeof(T)); }
    template<typename DT> void MemcpyToHost(DT* hptr, size_t hpitch) const
    {
        const int st = kfx_memcpy_2d(hptr, hpitch, (const void*)ptr, pitch, w * sizeof(T), h, TargetCopyKind<TargetHost, Target>(), 0);
        if (st != 0) throw HipException("Unable to copy in MemcpyToHost", st);
    }
    template<typename DT> void MemcpyToHost(DT* hptr) const { MemcpyToHost(hptr, w * sizeof(T)); }

    KANGAROO_HD void Swap(Image<T, Target, Management>& img)
    {
        const Image<T, Target, DontManage> t(ptr, w, h, pitch);
        pitch = img.pitch; ptr = img.ptr; w = img.w; h = img.h;
        img.pitch = t.pitch; img.ptr = t.ptr; img.w = t.w; img.h = t.h;
    }

    // ---- element access (host pointers on the host, device pointers in kernels) ------
    KANGAROO_HD T* RowPtr(size_t y) { return (T*)((unsigned char*)(ptr) + y * pitch); }
    KANGAROO_HD const T* RowPtr(size_t y) const { return (const T*)((const unsigned char*)(ptr) + y * pitch); }
    KANGAROO_HD T& operator()(size_t x, size_t y) { return RowPtr(y)[x]; }
    KANGAROO_HD const T& operator()(size_t x, size_t y) const { return RowPtr(y)[x]; }
    KANGAROO_HD T& operator[](size_t ix) { return ptr[ix]; }
    KANGAROO_HD const T& operator[](size_t ix) const { return ptr[ix]; }
    KANGAROO_HD const T& Get(int x, int y) const { return RowPtr(y)[x]; }

    KANGAROO_HD bool InBounds(int x, int y) const { return 0 <= x && x < (int)w && 0 <= y && y < (int)h; }
    KANGAROO_HD bool InBounds(float x, float y, float border) const
    {
        return border <= x && x < (w - border) && border <= y && y < (h - border);
    }
    KANGAROO_HD bool InBounds(const float2& p, float border) const { return InBounds(p.x, p.y, border); }

    KANGAROO_HD const T& GetWithClampedRange(int x, int y) const
    {
        x = clamp(x, 0, (int)w - 1);
        y = clamp(y, 0, (int)h - 1);
        return RowPtr(y)[x];
    }

    // bilinear sample; the row index goes floorf -> float -> size_t, the column (size_t)ix (quirk Q6)
    template<typename TR> KANGAROO_HD TR GetBilinear(float u, float v) const
    {
        const float ix = floorf(u), iy = floorf(v);
        const float fx = u - ix, fy = v - iy;
        const T* bl = RowPtr((size_t)iy) + (size_t)ix;
        const T* tl = RowPtr((size_t)(iy + 1)) + (size_t)ix;
        return lerp(lerp(bl[0], bl[1], fx), lerp(tl[0], tl[1], fx), fy);
    }
    template<typename TR> KANGAROO_HD TR GetBilinear(const float2& p) const { return GetBilinear<TR>(p.x, p.y); }
    KANGAROO_HD T GetBilinear(const float2& p) const { return GetBilinear<T>(p.x, p.y); }
    KANGAROO_HD T GetNearestNeighbour(float u, float v) const { return Get((int)(u + 0.5f), (int)(v + 0.5f)); }
    KANGAROO_HD T GetNearestNeighbour(const float2& p) const { return GetNearestNeighbour(p.x, p.y); }

    // ---- sub-views: same pitch, offset pointer -----------------------------------------
    KANGAROO_HD Image<T, Target, DontManage> SubImage(size_t x, size_t y, size_t width, size_t height) const
    {
        assert((x + width) <= w && (y + height) <= h);
        return Image<T, Target, DontManage>(const_cast<T*>(RowPtr(y)) + x, width, height, pitch);
    }
    KANGAROO_HD Image<T, Target, DontManage> SubImage(int width, int height) const
    {
        assert((size_t)width <= w && (size_t)height <= h);
        return Image<T, Target, DontManage>(ptr, width, height, pitch);
    }
    KANGAROO_HD Image<T, Target, DontManage> Row(int y) const { return SubImage(0, y, w, 1); }
    KANGAROO_HD Image<T, Target, DontManage> Col(int x) const { return SubImage(x, 0, 1, h); }

    // reinterpret this image's storage as a packed / aligned image of another type
    template<typename TP> KANGAROO_HD Image<TP, Target, DontManage> PackedImage(size_t width, size_t height)
    {
        assert(width * height * sizeof(TP) <= h * pitch);
        return Image<TP, Target, DontManage>((TP*)ptr, width, height, width * sizeof(TP));
    }
    template<typename TP> KANGAROO_HD Image<TP, Target, DontManage> AlignedImage(size_t width, size_t height, size_t align_bytes = 16)
    {
        const size_t wbytes = width * sizeof(TP);
        const size_t npitch = (wbytes % align_bytes) == 0 ? wbytes : align_bytes * (1 + wbytes / align_bytes);
        assert(npitch * height <= h * pitch);
        return Image<TP, Target, DontManage>((TP*)ptr, width, height, npitch);
    }

    // ---- members (public, as in the reference) ---------------------------------------------
    size_t pitch;
    T* ptr;
    size_t w;
    size_t h;
};

}
