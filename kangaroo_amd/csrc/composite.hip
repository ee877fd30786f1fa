// composite.hip -- per-pixel glue of the multi-GPU raycast composite (kangaroo_amd/pipeline.py): every
// rank ray-casts its own Z-slab; the nearest hit over all slabs wins.  Two collectives (RCCL, issued by
// the host through torch.distributed) carry the data; these kernels pack / select / unpack around them so
// that a frame costs three small launches instead of a dozen elementwise tensor ops.
//   key     = (depth bits << 8) | rank     positive floats order like their bit patterns; misses = +inf
//   all_reduce(MIN, key)                   -> winner rank and its depth, per pixel
//   payload = winner ? {n.x, n.y, n.z, shade} : 0           all_reduce(SUM, payload)   (n.w = hit ? 1 : 0 comes out of the key)
//
// The direct-send variant (kfx_composite_strips_*): xGMI is a full mesh of point-to-point links, so instead of two ring
// all-reduces over whole images every rank OWNS one strip of the image (pixels [j S, (j + 1) S) of the row-major image,
// S = kfx_composite_strip_pixels):
//   pack:    send[j][c][q] = plane c of pixel j S + q, c = {depth or +inf, n.x, n.y, n.z, shade}    then all-to-all (strip j -> rank j)
//   merge:   of the `world` copies of this rank's strip the nearest depth wins, the lowest rank on ties   then all-gather of the strips
//   unpack:  the merged strips back into the depth / normal / shade images
// 20 bytes per pixel cross each link once in each phase (7/8 of 6.1 MB per rank and phase at 640 x 480 on 8 GPUs, spread over
// seven links at once) against 2 x 7/8 x (2.4 + 4.9) MB in 14 dependent ring steps; same winner per pixel as the key's minimum.
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

struct StripParams {
    unsigned char *dptr, *nptr, *iptr;
    size_t dpitch, npitch, ipitch;
    float* buf;      // strip j: KFX_COMPOSITE_STRIP_PLANES planes of S floats at buf + j * stride
    size_t stride;
    int w, h, world;
    unsigned S;
};

__global__ __launch_bounds__(256) void k_strips_pack(const StripParams p)
{
    const unsigned q = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (q >= p.S) return;
    const size_t pix = (size_t)j * p.S + q;
    float d = __builtin_inff();
    float4 n = make_float4(0.f, 0.f, 0.f, 0.f);
    float s = 0.f;
    if (pix < (size_t)p.w * p.h) {
        const int v = (int)(pix / p.w), u = (int)(pix - (size_t)v * p.w);
        const float dd = reinterpret_cast<const float*>(p.dptr + (size_t)v * p.dpitch)[u];
        if (isfinite(dd)) {
            d = dd;
            n = reinterpret_cast<const float4*>(p.nptr + (size_t)v * p.npitch)[u];
            s = reinterpret_cast<const float*>(p.iptr + (size_t)v * p.ipitch)[u];
        }
    }
    float* o = p.buf + (size_t)j * p.stride + q;
    o[0] = d; o[p.S] = n.x; o[2 * (size_t)p.S] = n.y; o[3 * (size_t)p.S] = n.z; o[4 * (size_t)p.S] = s;
}

// in + r * stride: [5][S], rank r's copy of this rank's strip; out: [5][S]
__global__ __launch_bounds__(256) void k_strips_merge(const float* __restrict__ in, float* __restrict__ out, unsigned S, size_t stride, int world)
{
    const unsigned q = blockIdx.x * 256 + threadIdx.x;
    if (q >= S) return;
    unsigned best = 0x7f800000u;   // +inf: no rank hit
    int br = -1;
    for (int r = 0; r < world; ++r) {
        const unsigned b = __float_as_uint(in[(size_t)r * stride + q]);   // depths are positive: they order like their bits
        if (b < best) { best = b; br = r; }
    }
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (br >= 0)
        for (int c = 0; c < 4; ++c) v[c] = in[(size_t)br * stride + (size_t)(1 + c) * S + q];
    out[q] = __uint_as_float(best);
    for (int c = 0; c < 4; ++c) out[(size_t)(1 + c) * S + q] = v[c];
}

__global__ __launch_bounds__(256) void k_strips_unpack(const StripParams p)
{
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= p.w || v >= p.h) return;
    const size_t pix = (size_t)v * p.w + u;
    const unsigned j = (unsigned)(pix / p.S), q = (unsigned)(pix - (size_t)j * p.S);
    const float* o = p.buf + (size_t)j * p.stride + q;
    const float d = o[0];
    const bool hit = __float_as_uint(d) < 0x7f800000u;
    reinterpret_cast<float*>(p.dptr + (size_t)v * p.dpitch)[u] = hit ? d : __builtin_nanf("");
    reinterpret_cast<float4*>(p.nptr + (size_t)v * p.npitch)[u] = make_float4(o[p.S], o[2 * (size_t)p.S], o[3 * (size_t)p.S], hit ? 1.0f : 0.0f);
    reinterpret_cast<float*>(p.iptr + (size_t)v * p.ipitch)[u] = o[4 * (size_t)p.S];
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

// ---- direct-send variant ----
extern "C" size_t kfx_composite_strip_pixels(size_t w, size_t h, int world)
{
    if (world < 1) return 0;
    const size_t per = (w * h + (size_t)world - 1) / (size_t)world;
    return (per + 63) / 64 * 64;   // 256-byte planes
}

static int strip_params(StripParams& p, const kfx_image* depth, const kfx_image* norm, const kfx_image* img, float* buf, size_t stride, int world)
{
    if (!depth || !norm || !img || !depth->ptr || !norm->ptr || !img->ptr || !buf) return set_error(KFX_E_NULL, "composite strips: null argument");
    if (norm->w < depth->w || norm->h < depth->h || img->w < depth->w || img->h < depth->h) return set_error(KFX_E_SHAPE, "composite strips: image sizes");
    if (world < 1 || world > 256) return set_error(KFX_E_RANGE, "composite strips: world in [1, 256]");
    if ((((uintptr_t)norm->ptr | norm->pitch) & 15) || (((uintptr_t)depth->ptr | depth->pitch | (uintptr_t)img->ptr | img->pitch | (uintptr_t)buf) & 3))
        return set_error(KFX_E_ALIGN, "composite strips: alignment");
    const size_t S = kfx_composite_strip_pixels(depth->w, depth->h, world);
    if (S > 0x7fffffffull) return set_error(KFX_E_RANGE, "composite strips: image too large");
    if (stride && stride < KFX_COMPOSITE_STRIP_PLANES * S) return set_error(KFX_E_SHAPE, "composite strips: rank stride smaller than a strip");
    p = StripParams{(unsigned char*)depth->ptr, (unsigned char*)norm->ptr, (unsigned char*)img->ptr, depth->pitch, norm->pitch, img->pitch,
                    buf, stride ? stride : KFX_COMPOSITE_STRIP_PLANES * S, (int)depth->w, (int)depth->h, world, (unsigned)S};
    return 0;
}

extern "C" int kfx_composite_strips_pack(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, float* send, size_t rank_stride, int world,
                                         kfx_stream stream)
{
    StripParams p;
    if (int e = strip_params(p, depth, norm, img, send, rank_stride, world)) return e;
    if (p.S == 0) return 0;
    hipLaunchKernelGGL(k_strips_pack, dim3(ceil_div(p.S, 256u), world), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_composite_strips_pack");
}

extern "C" int kfx_composite_strips_merge(const float* recv, float* merged, size_t strip_pixels, size_t rank_stride, int world, kfx_stream stream)
{
    if (!recv || !merged) return set_error(KFX_E_NULL, "kfx_composite_strips_merge: null argument");
    if (world < 1 || world > 256 || strip_pixels > 0x7fffffffull) return set_error(KFX_E_RANGE, "kfx_composite_strips_merge: world / strip size");
    if (((uintptr_t)recv | (uintptr_t)merged) & 3) return set_error(KFX_E_ALIGN, "kfx_composite_strips_merge: alignment");
    if (rank_stride && rank_stride < KFX_COMPOSITE_STRIP_PLANES * strip_pixels) return set_error(KFX_E_SHAPE, "kfx_composite_strips_merge: rank stride smaller than a strip");
    if (strip_pixels == 0) return 0;
    hipLaunchKernelGGL(k_strips_merge, dim3(ceil_div((unsigned)strip_pixels, 256u)), dim3(256), 0, (hipStream_t)stream, recv, merged, (unsigned)strip_pixels,
                       rank_stride ? rank_stride : KFX_COMPOSITE_STRIP_PLANES * strip_pixels, world);
    return check_launch("kfx_composite_strips_merge");
}

extern "C" int kfx_composite_strips_unpack(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, const float* strips, size_t rank_stride,
                                           int world, kfx_stream stream)
{
    StripParams p;
    if (int e = strip_params(p, depth, norm, img, const_cast<float*>(strips), rank_stride, world)) return e;
    if (p.w == 0 || p.h == 0) return 0;
    hipLaunchKernelGGL(k_strips_unpack, dim3(ceil_div(p.w, 64), ceil_div(p.h, 4)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_composite_strips_unpack");
}
