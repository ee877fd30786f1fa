// debug.hip -- measurement aids, built into their own library (libkfx_debug.so, include/kfx_debug.h; it links against
// libkfx.so and is not part of the drop-in boundary): in-place read-modify-write sweeps of a TSDF volume with no
// arithmetic (the HBM ceiling of the SdfFuse access pattern on this chip), exhaustive checks of the kernels' arithmetic
// shortcuts against the hardware operations, and copies of a brick summary's tables for the tests.
#include "kfx_device.h"
#include "sampling.h"
#include "../../include/kfx_debug.h"

namespace kfx {

// variant 0: linear grid-stride float4 sweep over the contiguous span
__global__ __launch_bounds__(256) void k_rmw_linear(float4* __restrict__ base, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 c = base[i];
        c.y += 1.0f;
        c.w += 1.0f;
        base[i] = c;
    }
}

// variant 1: the tiled fuse kernel's mapping (64 x 8 x ZC brick, 2 voxels per lane, z-march)
template <int ZC>
__global__ __launch_bounds__(256) void k_rmw_brick(unsigned char* vptr, size_t pitch, size_t img_pitch, int X, int Y, int Z)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int x0 = (blockIdx.x * 32 + (lane & 31)) * 2;
    const int y = blockIdx.y * 8 + wv * 2 + (lane >> 5);
    const int zbeg = blockIdx.z * ZC, zend = min(zbeg + ZC, Z);
    if (x0 >= X || y >= Y) return;
    unsigned char* cell = vptr + (size_t)zbeg * img_pitch + (size_t)y * pitch + (size_t)x0 * 8;
    for (int z = zbeg; z < zend; ++z, cell += img_pitch) {
        float4 c = *reinterpret_cast<const float4*>(cell);
        c.y += 1.0f;
        c.w += 1.0f;
        *reinterpret_cast<float4*>(cell) = c;
    }
}

// variant 2: the first kernels' mapping (128 x 4 x ZC brick: one wave = one 1 KiB row segment)
template <int ZC>
__global__ __launch_bounds__(256) void k_rmw_rows(unsigned char* vptr, size_t pitch, size_t img_pitch, int X, int Y, int Z)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x0 = (blockIdx.x * 64 + lane) * 2;
    const int y = blockIdx.y * 4 + wv;
    const int zbeg = blockIdx.z * ZC, zend = min(zbeg + ZC, Z);
    if (x0 >= X || y >= Y) return;
    unsigned char* cell = vptr + (size_t)zbeg * img_pitch + (size_t)y * pitch + (size_t)x0 * 8;
    for (int z = zbeg; z < zend; ++z, cell += img_pitch) {
        float4 c = *reinterpret_cast<const float4*>(cell);
        c.y += 1.0f;
        c.w += 1.0f;
        *reinterpret_cast<float4*>(cell) = c;
    }
}

// variant 3: z-march, but each workgroup walks the WHOLE z range of its (x,y) footprint
__global__ __launch_bounds__(256) void k_rmw_column(unsigned char* vptr, size_t pitch, size_t img_pitch, int X, int Y, int Z)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x0 = (blockIdx.x * 64 + lane) * 2;
    const int y = blockIdx.y * 4 + wv;
    if (x0 >= X || y >= Y) return;
    unsigned char* cell = vptr + (size_t)y * pitch + (size_t)x0 * 8;
    for (int z = 0; z < Z; ++z, cell += img_pitch) {
        float4 c = *reinterpret_cast<const float4*>(cell);
        c.y += 1.0f;
        c.w += 1.0f;
        *reinterpret_cast<float4*>(cell) = c;
    }
}

// generic: WX waves side by side along x (each 128 voxels = 1 KiB), WY rows, ZC slices, optional
// nontemporal accesses; WX*WY = 4 waves
typedef float v4f_dbg __attribute__((ext_vector_type(4)));
template <int WX, int WY, int ZC, bool NT>
__global__ __launch_bounds__(256) void k_rmw_gen(unsigned char* vptr, size_t pitch, size_t img_pitch, int X, int Y, int Z)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x0 = ((blockIdx.x * WX + (wv % WX)) * 64 + lane) * 2;
    const int y = blockIdx.y * WY + (wv / WX);
    const int zbeg = blockIdx.z * ZC, zend = min(zbeg + ZC, Z);
    if (x0 >= X || y >= Y) return;
    unsigned char* cell = vptr + (size_t)zbeg * img_pitch + (size_t)y * pitch + (size_t)x0 * 8;
    for (int z = zbeg; z < zend; ++z, cell += img_pitch) {
        v4f_dbg c;
        if constexpr (NT) c = __builtin_nontemporal_load(reinterpret_cast<const v4f_dbg*>(cell));
        else c = *reinterpret_cast<const v4f_dbg*>(cell);
        c.y += 1.0f;
        c.w += 1.0f;
        if constexpr (NT) __builtin_nontemporal_store(c, reinterpret_cast<v4f_dbg*>(cell));
        else *reinterpret_cast<v4f_dbg*>(cell) = c;
    }
}

// exhaustive check of div_uniform against the hardware division: every numerator bit pattern for one divisor
__global__ __launch_bounds__(256) void k_div_uniform_check(float b, float inv, unsigned long long* __restrict__ out)
{
    unsigned long long bad = 0, tested = 0;
    const unsigned stride = gridDim.x * blockDim.x;
    unsigned bits = blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned it = 0; it < (1u << 16); ++it, bits += stride) { // grid = 2^16 threads: 2^16 iterations cover 2^32 patterns
        const float a = __uint_as_float(bits);
        if (!div_uniform_safe(a)) continue;
        const float want = a / b;
        const float got = div_uniform(a, b, inv);
        tested += 1;
        bad += (__float_as_uint(want) != __float_as_uint(got)) ? 1 : 0;
    }
    atomicAdd(out, bad);
    atomicAdd(out + 1, tested);
}

// div_core / rcp_nr against the hardware IEEE division.  Thread t sweeps divisor significands (every 2^23 of them
// across the grid) in each of a set of binades and pairs each with numerators from a counter-based generator
// (splitmix-style hash): random significands in random binades of [2^-60, 2^60], plus the awkward ones (all ones, one ulp
// above / below a power of two) and exact zeros.  A signed-zero quotient counts as equal to the other zero.
__device__ __forceinline__ unsigned hash32(unsigned long long x)
{
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return (unsigned)((x ^ (x >> 31)) >> 16);
}
__global__ __launch_bounds__(256) void k_div_core_check(unsigned seed, int per_divisor, unsigned long long* __restrict__ out)
{
    const int bexp[] = {-40, -20, -3, -1, 0, 1, 2, 10, 20, 40};
    unsigned long long bad = 0, tested = 0;
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (unsigned m = tid; m < (1u << 23); m += stride) {
        for (int be = 0; be < 10; ++be) {
            const unsigned bbits = ((unsigned)(127 + bexp[be]) << 23) | m | ((m & 1u) << 31); // alternate signs
            const float b = __uint_as_float(bbits);
            const float y = rcp_nr(b);
            for (int k = 0; k < per_divisor; ++k) {
                const unsigned h = hash32(((unsigned long long)seed << 40) ^ ((unsigned long long)m << 8) ^ (unsigned)(be * 64 + k));
                unsigned am = h & 0x7fffffu;
                const int sel = (h >> 23) & 15;
                if (sel == 0) am = 0x7fffffu;
                else if (sel == 1) am = 0u;
                else if (sel == 2) am = 1u;
                else if (sel == 3) am = 0x7ffffeu;
                const int ae = (int)((h >> 27) % 121u) - 60;
                unsigned abits = ((unsigned)(127 + ae) << 23) | am | (h & 0x80000000u);
                if (k == 0) abits = h & 0x80000000u; // +-0
                const float a = __uint_as_float(abits);
                const float want = a / b;
                const float got = div_core(a, b, y);
                tested += 1;
                const bool same = __float_as_uint(want) == __float_as_uint(got) || (want == 0.0f && got == 0.0f);
                bad += same ? 0 : 1;
            }
        }
    }
    atomicAdd(out, bad);
    atomicAdd(out + 1, tested);
}

// sqrt_core against sqrtf for every float in [2^-80, 2^80]
__global__ __launch_bounds__(256) void k_sqrt_core_check(unsigned long long* __restrict__ out)
{
    unsigned long long bad = 0, tested = 0;
    const unsigned lo = (unsigned)(127 - 80) << 23, hi = (unsigned)(127 + 80) << 23;
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned long long bits = (unsigned long long)lo + blockIdx.x * blockDim.x + threadIdx.x; bits <= hi; bits += stride) {
        const float x = __uint_as_float((unsigned)bits);
        tested += 1;
        bad += (__float_as_uint(sqrtf(x)) != __float_as_uint(sqrt_core(x))) ? 1 : 0;
    }
    atomicAdd(out, bad);
    atomicAdd(out + 1, tested);
}

// wave_xor_combine (DPP quad permutes, v_permlane16_swap / v_permlane32_swap) against __shfl_xor for the four lane
// distances, min and max, on pseudo-random lane values: out[0] += mismatching lanes, out[1] += lanes tested
__global__ __launch_bounds__(256) void k_wave_xor_check(unsigned seed, unsigned long long* __restrict__ out)
{
    unsigned s = seed ^ (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    unsigned long long bad = 0, tested = 0;
    const auto fmin2 = [](float a, float b) { return fminf(a, b); };
    const auto fmax2 = [](float a, float b) { return fmaxf(a, b); };
    for (int it = 0; it < 64; ++it) {
        s = s * 1664525u + 1013904223u;
        const float x = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
        const float r[8] = {wave_xor_combine<1>(x, fmin2), wave_xor_combine<2>(x, fmin2), wave_xor_combine<16>(x, fmin2), wave_xor_combine<32>(x, fmin2),
                            wave_xor_combine<1>(x, fmax2), wave_xor_combine<2>(x, fmax2), wave_xor_combine<16>(x, fmax2), wave_xor_combine<32>(x, fmax2)};
        const int off[4] = {1, 2, 16, 32};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float o = __shfl_xor(x, off[k], 64);
            bad += (__float_as_uint(r[k]) != __float_as_uint(fminf(x, o))) ? 1 : 0;
            bad += (__float_as_uint(r[4 + k]) != __float_as_uint(fmaxf(x, o))) ? 1 : 0;
            tested += 2;
        }
    }
    atomicAdd(out, bad);
    atomicAdd(out + 1, tested);
}

} // namespace kfx

using namespace kfx;

extern "C" int kfx_debug_wave_xor_check(unsigned seed, unsigned long long* d_out, kfx_stream stream)
{
    if (!d_out) return set_error(KFX_E_NULL, "kfx_debug_wave_xor_check");
    hipLaunchKernelGGL(k_wave_xor_check, dim3(64), dim3(256), 0, (hipStream_t)stream, seed, d_out);
    return check_launch("kfx_debug_wave_xor_check");
}

extern "C" int kfx_debug_div_core_check(unsigned seed, int per_divisor, unsigned long long* d_out, kfx_stream stream)
{
    if (!d_out) return set_error(KFX_E_NULL, "kfx_debug_div_core_check");
    if (per_divisor < 1 || per_divisor > 64) return set_error(KFX_E_RANGE, "kfx_debug_div_core_check: per_divisor in [1, 64]");
    hipLaunchKernelGGL(k_div_core_check, dim3(4096), dim3(256), 0, (hipStream_t)stream, seed, per_divisor, d_out);
    return check_launch("kfx_debug_div_core_check");
}

extern "C" int kfx_debug_sqrt_core_check(unsigned long long* d_out, kfx_stream stream)
{
    if (!d_out) return set_error(KFX_E_NULL, "kfx_debug_sqrt_core_check");
    hipLaunchKernelGGL(k_sqrt_core_check, dim3(4096), dim3(256), 0, (hipStream_t)stream, d_out);
    return check_launch("kfx_debug_sqrt_core_check");
}

extern "C" int kfx_debug_div_uniform_check(float b, unsigned long long* d_out, kfx_stream stream)
{
    if (!d_out) return set_error(KFX_E_NULL, "kfx_debug_div_uniform_check");
    if (!div_uniform_safe_host(b)) return set_error(KFX_E_RANGE, "kfx_debug_div_uniform_check: divisor outside [2^-40, 2^40]");
    hipLaunchKernelGGL(k_div_uniform_check, dim3(256), dim3(256), 0, (hipStream_t)stream, b, 1.0f / b, d_out);
    return check_launch("kfx_debug_div_uniform_check");
}

extern "C" int kfx_debug_rmw(const kfx_volume* vol, int variant, kfx_stream stream)
{
    if (!vol || !vol->ptr) return set_error(KFX_E_NULL, "kfx_debug_rmw");
    hipStream_t s = (hipStream_t)stream;
    const int X = (int)vol->w, Y = (int)vol->h, Z = (int)vol->d;
    unsigned char* p = (unsigned char*)vol->ptr;
    switch (variant) {
    case 0: {
        const size_t n4 = ((vol->d - 1) * vol->img_pitch + (vol->h - 1) * vol->pitch + vol->w * 8) / 16;
        hipLaunchKernelGGL(k_rmw_linear, dim3(256 * 8), dim3(256), 0, s, (float4*)p, n4);
        break;
    }
    case 1: hipLaunchKernelGGL(k_rmw_brick<16>, dim3(ceil_div(X, 64), ceil_div(Y, 8), ceil_div(Z, 16)), dim3(256), 0, s, p, vol->pitch, vol->img_pitch, X, Y, Z); break;
    case 2: hipLaunchKernelGGL(k_rmw_rows<16>, dim3(ceil_div(X, 128), ceil_div(Y, 4), ceil_div(Z, 16)), dim3(256), 0, s, p, vol->pitch, vol->img_pitch, X, Y, Z); break;
    case 3: hipLaunchKernelGGL(k_rmw_column, dim3(ceil_div(X, 128), ceil_div(Y, 4), 1), dim3(256), 0, s, p, vol->pitch, vol->img_pitch, X, Y, Z); break;
    case 4: hipLaunchKernelGGL(k_rmw_brick<64>, dim3(ceil_div(X, 64), ceil_div(Y, 8), ceil_div(Z, 64)), dim3(256), 0, s, p, vol->pitch, vol->img_pitch, X, Y, Z); break;
    case 5: hipLaunchKernelGGL(k_rmw_rows<4>, dim3(ceil_div(X, 128), ceil_div(Y, 4), ceil_div(Z, 4)), dim3(256), 0, s, p, vol->pitch, vol->img_pitch, X, Y, Z); break;
#define GEN(ID, WX, WY, ZC, NT) \
    case ID: hipLaunchKernelGGL((k_rmw_gen<WX, WY, ZC, NT>), dim3(ceil_div(X, 128 * WX), ceil_div(Y, WY), ceil_div(Z, ZC)), dim3(256), 0, s, p, vol->pitch, vol->img_pitch, X, Y, Z); break;
    GEN(10, 4, 1, 16, false)
    GEN(11, 2, 2, 16, false)
    GEN(12, 1, 4, 16, false)
    GEN(13, 1, 4, 16, true)
    GEN(14, 4, 1, 16, true)
    GEN(15, 1, 4, 1, false)
    GEN(16, 1, 4, 64, false)
    GEN(17, 4, 1, 4, false)
    GEN(18, 2, 2, 16, true)
#undef GEN
    default: return set_error(KFX_E_RANGE, "kfx_debug_rmw: variant");
    }
    return check_launch("kfx_debug_rmw");
}

// test / diagnostics aid (include/kfx_debug.h): copies of R (float4 per brick) and of the class tables built for (tol, vref,
// fine_shift) into caller buffers
extern "C" int kfx_debug_summary_export(kfx_sdf_summary* s, float tol, float vref, int fine_shift, void* R_out, void* C_out, int dims_out[12],
                                        kfx_stream stream)
{
    if (!s || !dims_out) return set_error(KFX_E_NULL, "kfx_debug_summary_export: null argument");
    if (fine_shift < 3 || fine_shift > 5) return set_error(KFX_E_RANGE, "kfx_debug_summary_export: fine_shift in [3, 5]");
    ClassView cv;
    summary_class_layout(s, fine_shift, cv);
    dims_out[0] = s->nbx; dims_out[1] = s->nby; dims_out[2] = s->nbz;
    dims_out[3] = cv.fine.first; dims_out[4] = cv.fine.rw; dims_out[5] = cv.fine.ny;
    dims_out[6] = cv.coarse.first; dims_out[7] = cv.coarse.rw; dims_out[8] = cv.coarse.ny;
    dims_out[9] = cv.words; dims_out[10] = s->n_coarse; dims_out[11] = (s->h_skippable && s->builds) ? ((volatile int*)s->h_skippable)[(s->builds - 1) % KFX_SUMMARY_RING] : -2;
    const size_t n = (size_t)s->nbx * s->nby * s->nbz;
    hipStream_t st = (hipStream_t)stream;
    if (R_out && hipMemcpyAsync(R_out, s->R, n * sizeof(float4), hipMemcpyDeviceToDevice, st) != hipSuccess) return set_error(KFX_E_RANGE, "kfx_debug_summary_export: copy");
    if (C_out) {
        if (int e = summary_classes_prepare(s, tol, vref, fine_shift, st)) return e;
        if (hipMemcpyAsync(C_out, s->C, (size_t)cv.words * sizeof(unsigned), hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return set_error(KFX_E_RANGE, "kfx_debug_summary_export: copy");
        dims_out[11] = (s->h_skippable && s->builds) ? ((volatile int*)s->h_skippable)[(s->builds - 1) % KFX_SUMMARY_RING] : -2;   // (the build has finished: its count is published)
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// kfx_debug_march_probe: what a step of the ray-march costs a wave depending on where its cells come from -- the experiment
// behind the LDS-slab march decision (DESIGN.md section 9, round-3 verdict item 4).  Every wave marches a 32 x 2 pixel tile
// of rays through the volume along +z, one voxel per step, each step's position depending on the previous sample's value
// (a real dependent chain, as in cu_raycast.cu:58-81): r voxels per pixel, a small slope in x and y.
//   mode 0: the plain march's sample -- four 16-byte global loads per lane and step (sampling.h), the wave waits for them;
//   mode 1: all cells from an LDS box staged once (z wraps inside the box): the pure cost of a step without memory latency;
//   mode 2: the workgroup (4 waves, 32 x 8 pixels) re-stages its box of S + 1 planes every S steps: loads, barrier, march, barrier;
//   mode 3: the same with the next box's loads issued before the current box is marched (registers as the second buffer).
// out[wave] = cycles (s_memtime) between the first and the last step of the wave; the caller divides by `steps`.
// ---------------------------------------------------------------------------------------------------------------------
struct ProbeParams {
    const unsigned char* base;
    unsigned pitch, img;
    int W, H, D;
    int steps, S;
    float r, sx, sy;
    unsigned long long* out;
};
constexpr int PROBE_BX = 56, PROBE_BY = 20, PROBE_MAXZ = 5, PROBE_REGS = 12;   // box of a 32 x 8 pixel workgroup tile: cells along x, y; planes

template <int MODE>
__global__ __launch_bounds__(256) void k_march_probe(const ProbeParams p)
{
    __shared__ float s_box[PROBE_MAXZ * PROBE_BY * PROBE_BX];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // the workgroup's tile: its own region of the volume (regions of different workgroups do not overlap)
    const int tiles_x = max(1, (p.W - 8) / (PROBE_BX + 8)), tile = blockIdx.x;
    const float cx = 4.f + (float)((tile % tiles_x) * (PROBE_BX + 8)), cy = 4.f + (float)(((tile / tiles_x) * (PROBE_BY + 4)) % max(1, p.H - PROBE_BY - 8));
    const float u = (float)(lane & 31), v = (float)(wv * 2 + (lane >> 5));
    float x = cx + u * p.r, y = cy + v * p.r, z = 1.0f;
    float acc = 0.f;
    const int S = p.S;
    // box origin in cells for the current stage (x / y fixed for the probe: the slopes are small enough for the box's margin)
    const int bx0 = (int)cx - 1, by0 = (int)cy - 1;
    // the box's rows as 16-byte loads (two cells each), round-robin over the threads; fully unrolled so that reg[] stays in
    // registers and all of a thread's loads are in flight together
    auto stage_loads = [&](int z0, float4 (&reg)[PROBE_REGS], int& n) {
        const int pairs_per_row = PROBE_BX / 2, total = pairs_per_row * PROBE_BY * (S + 1);
        n = 0;
#pragma unroll
        for (int k = 0; k < PROBE_REGS; ++k) {
            const int i = tid + k * 256;
            if (i < total) {
                const int row = i / pairs_per_row, px = i % pairs_per_row;
                const int zz = row / PROBE_BY, yy = row % PROBE_BY;
                reg[k] = *reinterpret_cast<const float4*>(p.base + (size_t)min(z0 + zz, p.D - 1) * p.img + (size_t)(by0 + yy) * p.pitch + (size_t)(bx0 + 2 * px) * 8);
            }
        }
    };
    auto stage_store = [&](const float4 (&reg)[PROBE_REGS], int) {
        const int pairs_per_row = PROBE_BX / 2, total = pairs_per_row * PROBE_BY * (S + 1);
#pragma unroll
        for (int k = 0; k < PROBE_REGS; ++k) {
            const int i = tid + k * 256;
            if (i < total) {
                const int row = i / pairs_per_row, px = i % pairs_per_row;
                *reinterpret_cast<float2*>(&s_box[row * PROBE_BX + 2 * px]) = make_float2(reg[k].x, reg[k].z);
            }
        }
    };
    float4 reg[PROBE_REGS];
    int nreg = 0;
    int box_z0 = 1;
    if constexpr (MODE >= 1) {
        stage_loads(box_z0, reg, nreg);
        stage_store(reg, nreg);
        __syncthreads();
        if constexpr (MODE == 3) stage_loads(box_z0 + S, reg, nreg);   // the next box, in flight while this one is marched
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int k = 0; k < p.steps; ++k) {
        const float fxx = floorf(x), fyy = floorf(y), fzz = floorf(z);
        const int ix = (int)fxx, iy = (int)fyy, iz = (int)fzz;
        const float fx = x - fxx, fy = y - fyy, fz = z - fzz;
        float sdf;
        if constexpr (MODE == 0) {
            const unsigned o = (unsigned)min(iz, p.D - 2) * p.img + (unsigned)iy * p.pitch + (unsigned)ix * 8u;
            float2 c00, c10, c01, c11;
            RayF32::pair4_off32(p.base, o, o + p.pitch, o + p.img, o + p.img + p.pitch, c00, c10, c01, c11);
            sdf = lerp(lerp(lerp(c00.x, c00.y, fx), lerp(c10.x, c10.y, fx), fy), lerp(lerp(c01.x, c01.y, fx), lerp(c11.x, c11.y, fx), fy), fz);
        } else {
            const int lz = (MODE == 1) ? (iz - 1) % S : (iz - box_z0);
            const float* q = s_box + ((lz * PROBE_BY) + (iy - by0)) * PROBE_BX + (ix - bx0);
            const float a0 = q[0], a1 = q[1], b0 = q[PROBE_BX], b1 = q[PROBE_BX + 1];
            const float* q1 = q + PROBE_BY * PROBE_BX;
            const float c0 = q1[0], c1 = q1[1], d0 = q1[PROBE_BX], d1 = q1[PROBE_BX + 1];
            sdf = lerp(lerp(lerp(a0, a1, fx), lerp(b0, b1, fx), fy), lerp(lerp(c0, c1, fx), lerp(d0, d1, fx), fy), fz);
        }
        acc += sdf;
        const float step = 1.0f + sdf * 0.0f;   // the step depends on the sample (not folded: IEEE semantics are on)
        z += step;
        x = cx + u * p.r + p.sx * (z - 1.0f) * 0.0f + (MODE == 0 ? p.sx * (z - 1.0f) : 0.f);
        y = cy + v * p.r + (MODE == 0 ? p.sy * (z - 1.0f) : 0.f);
        if constexpr (MODE >= 2) {
            if ((k + 1) % S == 0) {   // every ray of the workgroup has left the box: the next S + 1 planes
                box_z0 += S;
                if constexpr (MODE == 2) stage_loads(box_z0, reg, nreg);
                __syncthreads();          // everybody is done reading the old box
                stage_store(reg, nreg);
                __syncthreads();
                if constexpr (MODE == 3) stage_loads(box_z0 + S, reg, nreg);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) p.out[(size_t)blockIdx.x * 4 + wv] = t1 - t0;
    if (acc == -1.2345f) p.out[0] = 0;   // keep the chain alive
}

extern "C" int kfx_debug_march_probe(const kfx_volume* vol, int mode, int steps, int S, int workgroups, float r, unsigned long long* d_out, kfx_stream stream)
{
    using namespace kfx;
    if (!vol || !vol->ptr || !d_out) return set_error(KFX_E_NULL, "kfx_debug_march_probe: null argument");
    if (mode < 0 || mode > 3 || steps < 1 || S < 1 || S + 1 > PROBE_MAXZ || workgroups < 1) return set_error(KFX_E_RANGE, "kfx_debug_march_probe: parameters");
    if ((int)vol->d < steps + S + 4 || vol->w < 2 * PROBE_BX || vol->h < 2 * PROBE_BY) return set_error(KFX_E_SHAPE, "kfx_debug_march_probe: volume too small for the march");
    if ((double)vol->img_pitch * (double)vol->d >= 4294967296.0) return set_error(KFX_E_RANGE, "kfx_debug_march_probe: volumes below 4 GiB");
    // a workgroup's rays span 31 r cells in x and 7 r in y from its tile's corner (+ the mode-0 drift of 0.05 / 0.02 cells per
    // step): modes 1-3 read them from the 56 x 20-cell LDS box (corner at -1, the sample's +1 cell), mode 0 from the volume
    if (!(r > 0.f) || !(r <= 1.6f)) return set_error(KFX_E_RANGE, "kfx_debug_march_probe: r in (0, 1.6]");
    if (mode == 0) {
        const int tiles_x = (int)vol->w - 8 > PROBE_BX + 8 ? ((int)vol->w - 8) / (PROBE_BX + 8) : 1;
        const float x_far = 4.f + (float)((tiles_x - 1) * (PROBE_BX + 8)) + 31.f * r + 0.05f * (float)steps + 2.f;
        const float y_far = 4.f + (float)((int)vol->h - PROBE_BY - 8 > 1 ? (int)vol->h - PROBE_BY - 9 : 0) + 7.f * r + 0.02f * (float)steps + 2.f;
        if (!(x_far < (float)vol->w) || !(y_far < (float)vol->h)) return set_error(KFX_E_SHAPE, "kfx_debug_march_probe: the drifting rays of mode 0 would leave the volume");
    }
    ProbeParams p;
    p.base = (const unsigned char*)vol->ptr; p.pitch = (unsigned)vol->pitch; p.img = (unsigned)vol->img_pitch;
    p.W = (int)vol->w; p.H = (int)vol->h; p.D = (int)vol->d;
    p.steps = steps; p.S = S; p.r = r; p.sx = 0.05f; p.sy = 0.02f; p.out = d_out;
    const dim3 grid(workgroups), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(k_march_probe<0>, grid, block, 0, s, p);
    else if (mode == 1) hipLaunchKernelGGL(k_march_probe<1>, grid, block, 0, s, p);
    else if (mode == 2) hipLaunchKernelGGL(k_march_probe<2>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(k_march_probe<3>, grid, block, 0, s, p);
    return check_launch("kfx_debug_march_probe");
}
