// composite.hip -- per-pixel glue of the multi-GPU raycast composite (kangaroo_amd/pipeline.py): every
// rank ray-casts its own Z-slab; the nearest hit over all slabs wins.  Two collectives (RCCL, issued by
// the host through torch.distributed) carry the data; these kernels pack / select / unpack around them so
// that a frame costs three small launches instead of a dozen elementwise tensor ops.
//   key     = (depth bits << 8) | rank     positive floats order like their bit patterns; misses = +inf
//   all_reduce(MIN, key)                   -> winner rank and its depth, per pixel
//   payload = winner ? {n.x, n.y, n.z, shade} : 0           all_reduce(SUM, payload)   (n.w = hit ? 1 : 0 comes out of the key)
// No reference counterpart (the reference is single-GPU).
#include "kfx_device.h"

namespace kfx {

struct CompParams {
    unsigned char *dptr, *nptr, *iptr;
    size_t dpitch, npitch, ipitch;
    long long* key;   // w*h, dense
    float* payload;   // w*h*4, dense (KFX_COMPOSITE_PAYLOAD floats per pixel)
    int w, h, rank;
};

__global__ __launch_bounds__(256) void k_composite_pack(const CompParams p)
{
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= p.w || v >= p.h) return;
    const float d = reinterpret_cast<const float*>(p.dptr + (size_t)v * p.dpitch)[u];
    const float dd = isfinite(d) ? d : __builtin_inff();
    p.key[(size_t)v * p.w + u] = ((long long)__float_as_uint(dd) << 8) | (long long)p.rank;
}

// after the MIN all-reduce of `key`: keep this rank's normal / shade only where it won
__global__ __launch_bounds__(256) void k_composite_select(const CompParams p)
{
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= p.w || v >= p.h) return;
    const size_t i = (size_t)v * p.w + u;
    const long long k = p.key[i];
    const float d = reinterpret_cast<const float*>(p.dptr + (size_t)v * p.dpitch)[u];
    const bool mine = isfinite(d) && (int)(k & 0xff) == p.rank && (unsigned)(k >> 8) == __float_as_uint(d);
    float4 n = make_float4(0.f, 0.f, 0.f, 0.f);
    float s = 0.f;
    if (mine) {
        n = reinterpret_cast<const float4*>(p.nptr + (size_t)v * p.npitch)[u];
        s = reinterpret_cast<const float*>(p.iptr + (size_t)v * p.ipitch)[u];
    }
    reinterpret_cast<float4*>(p.payload)[i] = make_float4(n.x, n.y, n.z, s); // n.w of a hit is 1 (Q8): it travels as the key's hit bit
}

// after the SUM all-reduce of `payload`: write the composite images (depth comes back out of the key)
__global__ __launch_bounds__(256) void k_composite_unpack(const CompParams p)
{
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= p.w || v >= p.h) return;
    const size_t i = (size_t)v * p.w + u;
    const unsigned bits = (unsigned)(p.key[i] >> 8);
    const bool hit = bits < 0x7f800000u;
    const float4 o = reinterpret_cast<const float4*>(p.payload)[i];
    reinterpret_cast<float*>(p.dptr + (size_t)v * p.dpitch)[u] = hit ? __uint_as_float(bits) : __builtin_nanf("");
    reinterpret_cast<float4*>(p.nptr + (size_t)v * p.npitch)[u] = make_float4(o.x, o.y, o.z, hit ? 1.0f : 0.0f);
    reinterpret_cast<float*>(p.iptr + (size_t)v * p.ipitch)[u] = o.w;
}

} // namespace kfx

using namespace kfx;

static int comp_params(CompParams& p, const kfx_image* depth, const kfx_image* norm, const kfx_image* img, long long* key,
                       float* payload, int rank)
{
    if (!depth || !norm || !img || !depth->ptr || !norm->ptr || !img->ptr || !key) return set_error(KFX_E_NULL, "composite: null argument");
    if (norm->w < depth->w || norm->h < depth->h || img->w < depth->w || img->h < depth->h) return set_error(KFX_E_SHAPE, "composite: image sizes");
    if (rank < 0 || rank > 255) return set_error(KFX_E_RANGE, "composite: rank must fit 8 bits");
    if ((((uintptr_t)norm->ptr | norm->pitch) & 15) || (((uintptr_t)depth->ptr | depth->pitch | (uintptr_t)img->ptr | img->pitch) & 3) ||
        ((uintptr_t)key & 7) || ((uintptr_t)payload & 15))
        return set_error(KFX_E_ALIGN, "composite: alignment");
    p = CompParams{(unsigned char*)depth->ptr, (unsigned char*)norm->ptr, (unsigned char*)img->ptr, depth->pitch, norm->pitch, img->pitch,
                   key, payload, (int)depth->w, (int)depth->h, rank};
    return 0;
}

extern "C" int kfx_composite_pack(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, long long* key, int rank, kfx_stream stream)
{
    CompParams p;
    if (int e = comp_params(p, depth, norm, img, key, nullptr, rank)) return e;
    if (p.w == 0 || p.h == 0) return 0;
    hipLaunchKernelGGL(k_composite_pack, dim3(ceil_div(p.w, 64), ceil_div(p.h, 4)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_composite_pack");
}

extern "C" int kfx_composite_select(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, const long long* key,
                                    float* payload, int rank, kfx_stream stream)
{
    CompParams p;
    if (!payload) return set_error(KFX_E_NULL, "composite: null payload");
    if (int e = comp_params(p, depth, norm, img, const_cast<long long*>(key), payload, rank)) return e;
    if (p.w == 0 || p.h == 0) return 0;
    hipLaunchKernelGGL(k_composite_select, dim3(ceil_div(p.w, 64), ceil_div(p.h, 4)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_composite_select");
}

extern "C" int kfx_composite_unpack(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, const long long* key,
                                    const float* payload, kfx_stream stream)
{
    CompParams p;
    if (!payload) return set_error(KFX_E_NULL, "composite: null payload");
    if (int e = comp_params(p, depth, norm, img, const_cast<long long*>(key), const_cast<float*>(payload), 0)) return e;
    if (p.w == 0 || p.h == 0) return 0;
    hipLaunchKernelGGL(k_composite_unpack, dim3(ceil_div(p.w, 64), ceil_div(p.h, 4)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_composite_unpack");
}
