// Memory.h -- allocation targets and ownership policies of the roo:: containers.
//
// Mirrors the interface of the reference's include/kangaroo/Memory.h:15-179 (TargetHost /
// TargetDevice allocators, Manage / DontManage cleanup policies, AssignmentCheck, TargetCopyKind)
// on top of the C ABI in include/kfx.h: device memory comes from kfx_alloc_pitched (rows padded
// to 256 B for gfx950 coalescing), host memory from kfx_alloc_host (page-locked).
#pragma once

#include <cstddef>
#include <exception>
#include <sstream>
#include <string>

#include <kfx.h>
#include <kangaroo/platform.h>

namespace roo
{

// Thrown on allocation / copy failure (reference: CudaException, Memory.h:15-30).
struct HipException : public std::exception
{
    HipException(const std::string& what, int status = 0) : mStatus(status)
    {
        std::stringstream ss;
        ss << "HipException: " << what;
        if (status != 0) ss << " -- " << kfx_error_name(status) << " (" << status << "): " << kfx_last_error_string();
        mWhat = ss.str();
    }
    virtual ~HipException() throw() {}
    virtual const char* what() const throw() { return mWhat.c_str(); }
    std::string mWhat;
    int mStatus;
};

// Page-locked host memory; pitch == width (reference Memory.h:32-57).
struct TargetHost
{
    template<typename T> static void AllocatePitchedMem(T** hostPtr, size_t* pitch, size_t w, size_t h)
    {
        *pitch = w * sizeof(T);
        const int st = kfx_alloc_host((void**)hostPtr, *pitch * h);
        if (st != 0) throw HipException("Unable to allocate page-locked host memory", st);
    }
    template<typename T> static void AllocatePitchedMem(T** hostPtr, size_t* pitch, size_t* img_pitch, size_t w, size_t h, size_t d)
    {
        *pitch = w * sizeof(T);
        *img_pitch = *pitch * h;
        const int st = kfx_alloc_host((void**)hostPtr, *pitch * h * d);
        if (st != 0) throw HipException("Unable to allocate page-locked host memory", st);
    }
    template<typename T> static void DeallocatePitchedMem(T* hostPtr) { kfx_free_host((void*)hostPtr); }
};

// HBM; a volume is allocated as h*d pitched rows so that img_pitch = pitch*h (reference Memory.h:59-84).
struct TargetDevice
{
    template<typename T> static void AllocatePitchedMem(T** devPtr, size_t* pitch, size_t w, size_t h)
    {
        const int st = kfx_alloc_pitched((void**)devPtr, pitch, w * sizeof(T), h);
        if (st != 0) throw HipException("Unable to allocate pitched device memory", st);
    }
    template<typename T> static void AllocatePitchedMem(T** devPtr, size_t* pitch, size_t* img_pitch, size_t w, size_t h, size_t d)
    {
        const int st = kfx_alloc_pitched((void**)devPtr, pitch, w * sizeof(T), h * d);
        if (st != 0) throw HipException("Unable to allocate pitched device memory", st);
        *img_pitch = *pitch * h;
    }
    template<typename T> static void DeallocatePitchedMem(T* devPtr) { kfx_free((void*)devPtr); }
};

// kind argument of kfx_memcpy_2d (0 h->h, 1 h->d, 2 d->h, 3 d->d, 4 default); reference Memory.h:115-121
template<typename TargetTo, typename TargetFrom> inline int TargetCopyKind() { return 4; }
template<> inline int TargetCopyKind<TargetHost, TargetHost>() { return 0; }
template<> inline int TargetCopyKind<TargetDevice, TargetHost>() { return 1; }
template<> inline int TargetCopyKind<TargetHost, TargetDevice>() { return 2; }
template<> inline int TargetCopyKind<TargetDevice, TargetDevice>() { return 3; }

// Owning policy: the container frees its memory on destruction (reference Memory.h:135-150).
struct Manage
{
    static void AllocateCheck() {}
    template<typename T, typename Target> static void Cleanup(T* ptr)
    {
        if (ptr) Target::template DeallocatePitchedMem<T>(ptr);
    }
};

// Non-owning view policy, the default; allocating through it is an error (reference Memory.h:152-164).
struct DontManage
{
    static void AllocateCheck() { throw HipException("Image that doesn't own data should not call this constructor"); }
    template<typename T, typename Target> KANGAROO_HD static void Cleanup(T*) {}
};

// Only non-owning views may be copy-constructed, and only within one memory space: any other
// combination has no definition and fails to link (reference Memory.h:166-179).
template<typename ManagementTo, typename TargetTo, typename TargetFrom> KANGAROO_HD void AssignmentCheck();
template<> KANGAROO_HD inline void AssignmentCheck<DontManage, TargetDevice, TargetDevice>() {}
template<> KANGAROO_HD inline void AssignmentCheck<DontManage, TargetHost, TargetHost>() {}

}
