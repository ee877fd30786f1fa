// capi.hip -- the non-kernel half of the C ABI: error reporting, the pitched device
// allocator (roo::TargetDevice, reference Memory.h:59-84) and 2-D copies
// (Image::CopyFrom / MemcpyFromHost / MemcpyToHost, reference Image.h:174-213).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "kfx_device.h"

namespace kfx {

static thread_local char g_err[256] = "no error";

int set_error(int code, const char* what)
{
    if (code > 0)
        snprintf(g_err, sizeof(g_err), "%s: %s (hipError %d)", what, hipGetErrorString((hipError_t)code), code);
    else
        snprintf(g_err, sizeof(g_err), "%s (%s)", what, kfx_error_name(code));
    return code;
}

// Same point the reference checks (GpuCheckErrors -> cudaGetLastError, launch_utils.h:29-47),
// but the status is returned instead of exit(-1).
int check_launch(const char* what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error((int)e, what);
    return 0;
}

static std::atomic<int> g_math{[] {
    const char* e = getenv("KFX_MATH");
    return (e && (e[0] == 'f' || e[0] == 'F' || e[0] == '1')) ? KFX_MATH_FAST : KFX_MATH_EXACT;
}()};
int math_mode() { return g_math.load(std::memory_order_relaxed); }

} // namespace kfx

using namespace kfx;

extern "C" int kfx_set_math_mode(int mode)
{
    if (mode != KFX_MATH_EXACT && mode != KFX_MATH_FAST) return set_error(KFX_E_RANGE, "kfx_set_math_mode: unknown mode");
    return g_math.exchange(mode);
}
extern "C" int kfx_get_math_mode(void) { return math_mode(); }

extern "C" const char* kfx_last_error_string(void) { return g_err; }

extern "C" const char* kfx_error_name(int code)
{
    if (code == 0) return "ok";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    switch (code) {
    case KFX_E_NULL: return "KFX_E_NULL";
    case KFX_E_SHAPE: return "KFX_E_SHAPE";
    case KFX_E_ALIGN: return "KFX_E_ALIGN";
    case KFX_E_RANGE: return "KFX_E_RANGE";
    case KFX_E_NODEVICE: return "KFX_E_NODEVICE";
    default: return "KFX_E_UNKNOWN";
    }
}

extern "C" int kfx_version(void) { return KFX_VERSION_MAJOR * 100 + KFX_VERSION_MINOR; }

extern "C" int kfx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// one process (or host thread) per GPU: select the device the calling thread's allocations and launches go to
extern "C" int kfx_set_device(int device)
{
    const hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return set_error((int)e, "kfx_set_device");
    }
    return 0;
}

// Rows are padded to 256 B so that every row (and every z-slice, img_pitch = pitch*h)
// starts on a boundary that keeps 16-byte-per-lane wave accesses (1 KiB per instruction)
// and 128-B cache lines aligned.  hipMalloc itself returns >= 256-B aligned blocks.
extern "C" int kfx_alloc_pitched(void** dev_ptr, size_t* pitch, size_t width_bytes, size_t rows)
{
    if (!dev_ptr || !pitch) return set_error(KFX_E_NULL, "kfx_alloc_pitched: null out pointer");
    if (width_bytes == 0 || rows == 0) return set_error(KFX_E_SHAPE, "kfx_alloc_pitched: empty allocation");
    const size_t p = (width_bytes + 255) & ~(size_t)255;
    if (p < width_bytes || rows > ((size_t)-1) / p) return set_error(KFX_E_RANGE, "kfx_alloc_pitched: size overflow");
    void* d = nullptr;
    const hipError_t e = hipMalloc(&d, p * rows);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *dev_ptr = nullptr;
        *pitch = 0;
        return set_error((int)e, "kfx_alloc_pitched: hipMalloc");
    }
    *dev_ptr = d;
    *pitch = p;
    return 0;
}

extern "C" int kfx_free(void* dev_ptr)
{
    if (!dev_ptr) return 0;
    const hipError_t e = hipFree(dev_ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return set_error((int)e, "kfx_free: hipFree");
    }
    return 0;
}

extern "C" int kfx_alloc_host(void** host_ptr, size_t bytes)
{
    if (!host_ptr) return set_error(KFX_E_NULL, "kfx_alloc_host: null out pointer");
    if (bytes == 0) return set_error(KFX_E_SHAPE, "kfx_alloc_host: empty allocation");
    void* h = nullptr;
    const hipError_t e = hipHostMalloc(&h, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *host_ptr = nullptr;
        return set_error((int)e, "kfx_alloc_host: hipHostMalloc");
    }
    *host_ptr = h;
    return 0;
}

extern "C" int kfx_free_host(void* host_ptr)
{
    if (!host_ptr) return 0;
    const hipError_t e = hipHostFree(host_ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return set_error((int)e, "kfx_free_host: hipHostFree");
    }
    return 0;
}

extern "C" int kfx_memcpy_2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes,
                             size_t rows, int kind, kfx_stream stream)
{
    if (width_bytes == 0 || rows == 0) return 0;
    if (!dst || !src) return set_error(KFX_E_NULL, "kfx_memcpy_2d: null pointer");
    if (dpitch < width_bytes || spitch < width_bytes) return set_error(KFX_E_SHAPE, "kfx_memcpy_2d: pitch < width");
    hipMemcpyKind k;
    switch (kind) {
    case 0: k = hipMemcpyHostToHost; break;
    case 1: k = hipMemcpyHostToDevice; break;
    case 2: k = hipMemcpyDeviceToHost; break;
    case 3: k = hipMemcpyDeviceToDevice; break;
    case 4: k = hipMemcpyDefault; break;
    default: return set_error(KFX_E_RANGE, "kfx_memcpy_2d: kind");
    }
    hipError_t e;
    if (stream)
        e = hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, rows, k, (hipStream_t)stream);
    else
        e = hipMemcpy2D(dst, dpitch, src, spitch, width_bytes, rows, k); // blocking, like cudaMemcpy2D in Image.h:178
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return set_error((int)e, "kfx_memcpy_2d");
    }
    return 0;
}

extern "C" int kfx_stream_synchronize(kfx_stream stream)
{
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return set_error((int)e, "kfx_stream_synchronize");
    }
    return 0;
}
