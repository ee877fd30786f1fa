// summary.hip -- the brick summary of a TSDF volume (kfx_sdf_summary, include/kfx.h): creation, the conservative state
// changes for writers that do not track (invalidate), SdfReset with tracking, and the class tables R -> C the ray-march
// stages in LDS.  The summary is maintained by k_sdf_fuse_tiled<..., TRACK> (fuse.hip) and consumed by
// k_raycast_sdf_classes (raycast.hip).  No reference counterpart: the reference's march samples the volume at every step (cu_raycast.cu:58-81).
#include <algorithm>
#include <new>

#include "kfx_device.h"

namespace kfx {

__global__ __launch_bounds__(256) void k_summary_fill(float4* __restrict__ R, size_t n, float lo, float hi, int state)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) R[i] = make_float4(lo, hi, __int_as_float(state), 0.f);
}

// Class tables (ClassView, kfx_device.h).  One lane per entry, 64 consecutive entries of a row per wave; the two planes are
// the wave's ballots.
//   fine level (2^shift cells, shift 3 or 4): from the summary bricks [b << (shift - 3), (b + 1) << (shift - 3)] per axis
//   (the last one holds the +1 cells), clamped to the grid;
//   32^3-cell level: from the fine level's entries -- the cells [32 c, 32 c + 32] of entry c are exactly those of the fine
//   entries [c m, c m + m) per axis (m = 32 >> shift; each fine entry already includes its +1 cells), so the class is the
//   combination of m^3 two-bit reads instead of 125 brick ranges.  `src` = the fine level's words (same kernel, second launch).
// count != nullptr (the 32^3-cell level): the waves add up their entries of class != 0; the last one to arrive publishes the
// total in host-visible memory (*publish) and clears the counters for the next build -- the host reads that word without
// synchronising (raycast.hip, class_view).
struct ClassBuild {
    const float4* R;
    unsigned* C;
    int nbx, nby, nbz;
    int shift, nx, ny, nz, rw;
    float lo_ok, hi_ok;
    const unsigned* src;   // M == 0: the fine level's words
    int src_shift, src_nx, src_ny, src_nz, src_rw;
};
// one wave = 64 consecutive entries of a row; returns (lane 0) the number of them with class != 0
template <int M>   // summary bricks per fine entry and axis (1: 8^3 cells, 2: 16^3, 4: 32^3 straight from R); 0: combine `src`
__device__ __forceinline__ int class_row(const ClassBuild& b, const long long wave)
{
    const float4* __restrict__ R = b.R;
    unsigned* __restrict__ C = b.C;
    const unsigned* __restrict__ src = b.src;
    const int nbx = b.nbx, nby = b.nby, nbz = b.nbz, shift = b.shift, nx = b.nx, ny = b.ny, rw = b.rw;
    const int src_shift = b.src_shift, src_nx = b.src_nx, src_ny = b.src_ny, src_nz = b.src_nz, src_rw = b.src_rw;
    const float lo_ok = b.lo_ok, hi_ok = b.hi_ok;
    const int lane = threadIdx.x & 63;
    const int chunks = (nx + 63) >> 6;
    const int chunk = (int)(wave % chunks), by = (int)((wave / chunks) % ny), bz = (int)(wave / ((long long)chunks * ny));
    const int bx = chunk * 64 + lane;
    int cls = 0;
    if (bx < nx) {
        bool all_nan = true, all_free = true, all_either = true;
        if constexpr (M == 0) {
            const int m = 1 << (shift - src_shift);   // 2 (16^3-cell fine level) or 4 (8^3)
            // the m entries along x lie in one word pair (m divides 32): one 8-byte read per (y, z), all requested up front
            for (int z0 = 0; z0 < m; z0 += 2) {
                uint2 w2[2][4];
#pragma unroll
                for (int dz = 0; dz < 2; ++dz)
#pragma unroll
                    for (int dy = 0; dy < 4; ++dy) {
                        const int z = min(bz * m + z0 + dz, src_nz - 1), y = min(by * m + min(dy, m - 1), src_ny - 1);
                        w2[dz][dy] = *reinterpret_cast<const uint2*>(src + ((size_t)z * src_ny + y) * src_rw + (((bx * m) >> 5) << 1));
                    }
#pragma unroll
                for (int dz = 0; dz < 2; ++dz)
#pragma unroll
                    for (int dy = 0; dy < 4; ++dy)
                        for (int dx = 0; dx < m; ++dx) {
                            const int x = min(bx * m + dx, src_nx - 1);   // (clamped entries repeat a neighbour: same verdict)
                            const int c = (int)((w2[dz][dy].x >> (x & 31)) & 1u) | (int)(((w2[dz][dy].y >> (x & 31)) & 1u) << 1);
                            all_free = all_free && c == 1;
                            all_nan = all_nan && c == 2;
                            all_either = all_either && c != 0;
                        }
            }
        } else if constexpr (M > 2) {   // 125 ranges per entry: a loop (volumes whose finer tables do not fit LDS)
            for (int z = bz * M; z <= bz * M + M; ++z)
                for (int y = by * M; y <= by * M + M; ++y)
                    for (int x = bx * M; x <= bx * M + M; ++x) {
                        const float4 r = R[((size_t)min(z, nbz - 1) * nby + min(y, nby - 1)) * nbx + min(x, nbx - 1)];
                        const int st = __float_as_int(r.z);
                        const bool in_band = r.x >= lo_ok && r.y <= hi_ok;
                        all_nan = all_nan && st == 1;
                        all_free = all_free && st == 0 && in_band;
                        all_either = all_either && (st == 1 || in_band);
                    }
        } else {
            // (M + 1)^3 brick ranges, all requested before the first is looked at (fully unrolled: the loop form waited for
            // each row of loads in turn, 9.5 us per launch at 512^3 against ~3)
            float4 r[(M + 1) * (M + 1) * (M + 1)];
#pragma unroll
            for (int dz = 0; dz <= M; ++dz)
#pragma unroll
                for (int dy = 0; dy <= M; ++dy)
#pragma unroll
                    for (int dx = 0; dx <= M; ++dx)
                        r[(dz * (M + 1) + dy) * (M + 1) + dx] =
                            R[((size_t)min(bz * M + dz, nbz - 1) * nby + min(by * M + dy, nby - 1)) * nbx + min(bx * M + dx, nbx - 1)];
#pragma unroll
            for (int k = 0; k < (M + 1) * (M + 1) * (M + 1); ++k) {
                const int st = __float_as_int(r[k].z);
                const bool in_band = r[k].x >= lo_ok && r[k].y <= hi_ok;   // every valued cell of the brick holds vref (false for the unknown state's infinite range)
                all_nan = all_nan && st == 1;
                all_free = all_free && st == 0 && in_band;
                all_either = all_either && (st == 1 || in_band);
            }
        }
        cls = all_free ? 1 : (all_nan ? 2 : (all_either ? 3 : 0));
    }
    const unsigned long long p0 = __ballot(cls & 1), p1 = __ballot(cls & 2);
    if (lane == 0) {
        unsigned* row = C + ((size_t)bz * ny + by) * rw + chunk * 4;
        row[0] = (unsigned)p0; row[1] = (unsigned)p1;
        if (chunk * 4 + 2 < rw) { row[2] = (unsigned)(p0 >> 32); row[3] = (unsigned)(p1 >> 32); }
    }
    return __popcll(p0 | p1);
}

// the fine level: one wave per row of 64 entries
template <int M>
__global__ __launch_bounds__(256) void k_summary_classes(const ClassBuild b, const long long n_waves)
{
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave < n_waves) class_row<M>(b, wave);
}

// the 32^3-cell level: the workgroups add up their entries of class != 0 (LDS), take a ticket, and the last one publishes the
// total in host-visible memory and clears the counters for the next build -- the host reads that word without synchronising
// (raycast.hip, class_view).  count = {running total, tickets taken}.
template <int M>
__global__ __launch_bounds__(256) void k_summary_classes_coarse(const ClassBuild b, const long long n_waves, int* __restrict__ count, int* __restrict__ publish)
{
    __shared__ int s_count[4];
    const int wv = threadIdx.x >> 6;
    const long long wave = (long long)blockIdx.x * 4 + wv;
    const int n = wave < n_waves ? class_row<M>(b, wave) : 0;
    if ((threadIdx.x & 63) == 0) s_count[wv] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&count[0], s_count[0] + s_count[1] + s_count[2] + s_count[3]);
        __threadfence();
        if (atomicAdd(&count[1], 1) == (int)gridDim.x - 1) {   // every workgroup's contribution is in
            __threadfence();
            const int total = atomicExch(&count[0], 0);
            count[1] = 0;
            __hip_atomic_store(publish, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// kfx_sdf_summary_rebuild: R from the volume itself.  One wave per 8 x 8 x 8 brick: a lane reads one row of eight cells (four
// 16-byte loads: 64 contiguous bytes), keeps the range of the values and whether it saw a NaN / a value; DPP and permlane swaps
// reduce the wave.  The states are the TRACK kernels' (kfx_device.h): 0 every cell valued, 1 every cell NaN, 2 both kinds --
// with the exact range of the valued cells in every case, which is at least as tight as what tracking arrives at.
__global__ __launch_bounds__(256) void k_summary_rebuild(float4* __restrict__ R, const unsigned char* __restrict__ base, size_t pitch, size_t img_pitch,
                                                         int w, int h, int d, int nbx, int nby, int nbz, int aligned16)
{
    const long long brick = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (brick >= (long long)nbx * nby * nbz) return;   // wave-uniform
    const int lane = threadIdx.x & 63;
    const int bx = (int)(brick % nbx), by = (int)((brick / nbx) % nby), bz = (int)(brick / ((long long)nbx * nby));
    const int y = by * 8 + (lane & 7), z = bz * 8 + (lane >> 3), x0 = bx * 8;
    float lo = __builtin_inff(), hi = -__builtin_inff();
    bool nan = false, val = false;
    if (y < h && z < d) {
        const unsigned char* row = base + (size_t)z * img_pitch + (size_t)y * pitch + (size_t)x0 * 8;
        if (x0 + 8 <= w && aligned16) {
            float4 c[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = reinterpret_cast<const float4*>(row)[k];   // {val, w, val, w}
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = c[k].x, b = c[k].z;
                nan = nan || a != a || b != b;
                val = val || a == a || b == b;
                lo = fminf(lo, fminf(a, b));   // (minNum / maxNum: a NaN operand leaves the other)
                hi = fmaxf(hi, fmaxf(a, b));
            }
        } else {
            const int x1 = x0 + 8 < w ? x0 + 8 : w;   // this brick's eight cells (fewer in the volume's last brick)
            for (int x = x0; x < x1; ++x) {
                const float a = *reinterpret_cast<const float*>(row + (size_t)(x - x0) * 8);
                nan = nan || a != a;
                val = val || a == a;
                lo = fminf(lo, a);
                hi = fmaxf(hi, a);
            }
        }
    }
    const auto fmin2 = [](float a, float b) { return fminf(a, b); };
    const auto fmax2 = [](float a, float b) { return fmaxf(a, b); };
    lo = wave_xor_combine<1>(lo, fmin2); hi = wave_xor_combine<1>(hi, fmax2);
    lo = wave_xor_combine<2>(lo, fmin2); hi = wave_xor_combine<2>(hi, fmax2);
#pragma unroll
    for (int off = 4; off <= 8; off <<= 1) { lo = fminf(lo, __shfl_xor(lo, off, 64)); hi = fmaxf(hi, __shfl_xor(hi, off, 64)); }
    lo = wave_xor_combine<16>(lo, fmin2); hi = wave_xor_combine<16>(hi, fmax2);
    lo = wave_xor_combine<32>(lo, fmin2); hi = wave_xor_combine<32>(hi, fmax2);
    const bool any_nan = __ballot(nan) != 0ull, any_val = __ballot(val) != 0ull;
    if (lane == 0) R[brick] = make_float4(lo, hi, __int_as_float(any_val ? (any_nan ? 2 : 0) : 1), 0.f);
}

static void class_level(ClassLevel& L, int shift, int w, int h, int first)
{
    L.shift = shift;
    L.ny = ceil_div(h, 1 << shift);
    L.rw = 2 * ceil_div(ceil_div(w, 1 << shift), 32);
    L.first = first;
}
static int class_level_words(const ClassLevel& L, int d) { return L.rw * L.ny * ceil_div(d, 1 << L.shift); }

// layout of the tables for a fine level of 2^fine_shift cells (3 or 4; 5 = the coarse level only)
void summary_class_layout(const kfx_sdf_summary* s, int fine_shift, ClassView& cv)
{
    class_level(cv.fine, fine_shift, s->w, s->h, 0);
    const int fw = fine_shift < 5 ? class_level_words(cv.fine, s->d) : 0;
    class_level(cv.coarse, 5, s->w, s->h, (fw + 3) & ~3);
    cv.words = cv.coarse.first + ((class_level_words(cv.coarse, s->d) + 3) & ~3);
}

int summary_classes_prepare(kfx_sdf_summary* s, float tol, float vref, int fine_shift, hipStream_t stream)
{
    if (!s->c_dirty && s->c_tol == tol && s->c_vref == vref && s->c_shift == fine_shift) return 0;
    ClassView cv;
    summary_class_layout(s, fine_shift, cv);
    const float lo_ok = vref - tol * vref, hi_ok = vref + tol * vref;
    for (int pass = 0; pass < 2; ++pass) {
        const ClassLevel& L = pass ? cv.coarse : cv.fine;
        if (!pass && fine_shift >= 5) continue;
        ClassBuild b;
        b.R = s->R; b.C = s->C + L.first;
        b.nbx = s->nbx; b.nby = s->nby; b.nbz = s->nbz;
        b.shift = L.shift; b.nx = ceil_div(s->w, 1 << L.shift); b.ny = L.ny; b.nz = ceil_div(s->d, 1 << L.shift); b.rw = L.rw;
        b.lo_ok = lo_ok; b.hi_ok = hi_ok;
        const bool from_fine = pass && fine_shift < 5;   // the 32^3-cell level combines the fine level's entries
        b.src = from_fine ? s->C + cv.fine.first : nullptr;
        b.src_shift = cv.fine.shift; b.src_nx = ceil_div(s->w, 1 << cv.fine.shift); b.src_ny = cv.fine.ny;
        b.src_nz = ceil_div(s->d, 1 << cv.fine.shift); b.src_rw = cv.fine.rw;
        const long long waves = (long long)ceil_div(b.nx, 64) * b.ny * b.nz;
        if (pass) {
            const dim3 grid((unsigned)((waves + 3) / 4));
            const unsigned slot = s->builds % KFX_SUMMARY_RING;
            int* publish = s->d_skippable + (s->h_skippable ? slot : 0);
            if (from_fine) hipLaunchKernelGGL(k_summary_classes_coarse<0>, grid, dim3(256), 0, stream, b, waves, s->d_count, publish);
            else hipLaunchKernelGGL(k_summary_classes_coarse<4>, grid, dim3(256), 0, stream, b, waves, s->d_count, publish);
            if (s->h_skippable) (void)hipEventRecord(s->build_done[slot], stream);
            s->builds += 1;
        } else if (L.shift == 3) {
            hipLaunchKernelGGL(k_summary_classes<1>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, b, waves);
        } else {
            hipLaunchKernelGGL(k_summary_classes<2>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, b, waves);
        }
        if (int e = check_launch("kfx_sdf_summary (classes)")) return e;
    }
    s->c_dirty = 0; s->c_tol = tol; s->c_vref = vref; s->c_shift = fine_shift;
    return 0;
}

int summary_view_offset(const kfx_sdf_summary* s, const kfx_volume* view, int* ox, int* oy, int* oz)
{
    if (!s || !view || !view->ptr) return set_error(KFX_E_NULL, "summary: null argument");
    if (view->pitch != s->pitch || view->img_pitch != s->img_pitch) return set_error(KFX_E_SHAPE, "summary: the volume is not a view of the summary's volume (pitches)");
    const unsigned char* q = static_cast<const unsigned char*>(view->ptr);
    if (q < s->base) return set_error(KFX_E_SHAPE, "summary: the volume is not a view of the summary's volume");
    const size_t off = (size_t)(q - s->base);
    const size_t z = off / s->img_pitch, rem = off % s->img_pitch, y = rem / s->pitch, xb = rem % s->pitch;
    if (xb % 8 || z + view->d > (size_t)s->d || y + view->h > (size_t)s->h || xb / 8 + view->w > (size_t)s->w)
        return set_error(KFX_E_SHAPE, "summary: the volume is not a view of the summary's volume (extent)");
    *ox = (int)(xb / 8); *oy = (int)y; *oz = (int)z;
    return 0;
}

} // namespace kfx

using namespace kfx;

extern "C" int kfx_sdf_summary_create(kfx_sdf_summary** out, const kfx_volume* vol)
{
    if (!out || !vol || !vol->ptr) return set_error(KFX_E_NULL, "kfx_sdf_summary_create: null argument");
    if (vol->w < 2 || vol->h < 2 || vol->d < 2 || vol->w > 65535 || vol->h > 65535 || vol->d > 65535) return set_error(KFX_E_SHAPE, "kfx_sdf_summary_create: volume dimensions");
    kfx_sdf_summary* s = new (std::nothrow) kfx_sdf_summary;
    if (!s) return set_error(KFX_E_RANGE, "kfx_sdf_summary_create: out of memory");
    s->nbx = ceil_div((int)vol->w, 8); s->nby = ceil_div((int)vol->h, 8); s->nbz = ceil_div((int)vol->d, 8);
    s->w = (int)vol->w; s->h = (int)vol->h; s->d = (int)vol->d;
    s->base = static_cast<const unsigned char*>(vol->ptr);
    s->pitch = vol->pitch; s->img_pitch = vol->img_pitch;
    s->R = nullptr; s->C = nullptr; s->d_count = nullptr; s->h_skippable = nullptr; s->d_skippable = nullptr;
    s->c_dirty = 1; s->c_tol = -1.f; s->c_vref = 0.f; s->c_shift = 0; s->sweeps = 0;
    s->builds = 0; s->plain_calls = 0;
    for (auto& e : s->build_done) e = nullptr;
    s->n_coarse = ceil_div(s->w, 32) * ceil_div(s->h, 32) * ceil_div(s->d, 32);
    const size_t n = (size_t)s->nbx * s->nby * s->nbz;
    ClassView cv;
    summary_class_layout(s, 3, cv);   // the finest level is the largest table
    bool ok = hipMalloc((void**)&s->R, n * sizeof(float4)) == hipSuccess && hipMalloc((void**)&s->C, (size_t)cv.words * sizeof(unsigned)) == hipSuccess &&
              hipMalloc((void**)&s->d_count, 2 * sizeof(int)) == hipSuccess && hipMemset(s->d_count, 0, 2 * sizeof(int)) == hipSuccess;
    // the published count lives in pinned host memory the device can write; without it the march always uses the tables
    if (ok && hipHostMalloc((void**)&s->h_skippable, KFX_SUMMARY_RING * sizeof(int), hipHostMallocMapped) == hipSuccess) {
        for (int i = 0; i < KFX_SUMMARY_RING; ++i) s->h_skippable[i] = -1;
        bool evs = hipHostGetDevicePointer((void**)&s->d_skippable, s->h_skippable, 0) == hipSuccess;
        for (int i = 0; evs && i < KFX_SUMMARY_RING; ++i) evs = hipEventCreateWithFlags(&s->build_done[i], hipEventDisableTiming) == hipSuccess;
        if (!evs) {
            for (auto& e : s->build_done) { if (e) (void)hipEventDestroy(e); e = nullptr; }
            (void)hipHostFree(s->h_skippable);
            s->h_skippable = nullptr;
        }
    } else {
        s->h_skippable = nullptr;
    }
    if (ok && !s->h_skippable) {   // a device word nobody reads keeps the kernel's interface the same
        (void)hipGetLastError();
        ok = hipMalloc((void**)&s->d_skippable, sizeof(int)) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        kfx_sdf_summary_destroy(s);
        return set_error(KFX_E_NODEVICE, "kfx_sdf_summary_create: hipMalloc");
    }
    *out = s;
    return kfx_sdf_summary_invalidate(s, nullptr); // nothing is known about the volume's contents yet
}

extern "C" int kfx_sdf_summary_destroy(kfx_sdf_summary* s)
{
    if (!s) return 0;
    (void)hipDeviceSynchronize();   // a table build may still be about to publish its count
    if (s->R) (void)hipFree(s->R);
    if (s->C) (void)hipFree(s->C);
    if (s->d_count) (void)hipFree(s->d_count);
    for (auto& e : s->build_done) if (e) (void)hipEventDestroy(e);
    if (s->h_skippable) (void)hipHostFree(s->h_skippable);
    else if (s->d_skippable) (void)hipFree(s->d_skippable);
    delete s;
    return 0;
}

extern "C" int kfx_sdf_summary_rebuild(kfx_sdf_summary* s, kfx_stream stream)
{
    if (!s) return set_error(KFX_E_NULL, "kfx_sdf_summary_rebuild: null summary");
    const long long n = (long long)s->nbx * s->nby * s->nbz;
    const int aligned16 = ((((uintptr_t)s->base) | s->pitch | s->img_pitch) & 15) == 0;
    hipLaunchKernelGGL(k_summary_rebuild, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s->R, s->base, s->pitch, s->img_pitch,
                       s->w, s->h, s->d, s->nbx, s->nby, s->nbz, aligned16);
    s->c_dirty = 1;
    return check_launch("kfx_sdf_summary_rebuild");
}

// After the volume was written by anything that does not track (memcpy, LoadPXM, SdfSphere, an untracked SdfFuse):
// every brick becomes "unknown", so the march samples everywhere until tracked updates re-establish ranges.
extern "C" int kfx_sdf_summary_invalidate(kfx_sdf_summary* s, kfx_stream stream)
{
    if (!s) return set_error(KFX_E_NULL, "kfx_sdf_summary_invalidate: null summary");
    const size_t n = (size_t)s->nbx * s->nby * s->nbz;
    hipLaunchKernelGGL(k_summary_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s->R, n, -__builtin_inff(), __builtin_inff(), 2);
    s->c_dirty = 1;
    return check_launch("kfx_sdf_summary_invalidate");
}

// SdfReset(vol, trunc_dist) of the WHOLE volume with the summary set to match: every cell = trunc_dist (NaN: never observed)
extern "C" int kfx_sdf_reset_tracked(const kfx_volume* vol, kfx_sdf_summary* s, float trunc_dist, kfx_stream stream)
{
    if (!s) return set_error(KFX_E_NULL, "kfx_sdf_reset_tracked: null summary");
    int ox, oy, oz;
    if (int e = summary_view_offset(s, vol, &ox, &oy, &oz)) return e;
    if (ox || oy || oz || (int)vol->w != s->w || (int)vol->h != s->h || (int)vol->d != s->d)
        return set_error(KFX_E_SHAPE, "kfx_sdf_reset_tracked: resets the whole volume only (use kfx_sdf_reset + kfx_sdf_summary_invalidate for views)");
    if (int e = kfx_sdf_reset(vol, trunc_dist, stream)) return e;
    const size_t n = (size_t)s->nbx * s->nby * s->nbz;
    const bool nan = trunc_dist != trunc_dist;
    hipLaunchKernelGGL(k_summary_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s->R, n,
                       nan ? __builtin_inff() : trunc_dist, nan ? -__builtin_inff() : trunc_dist, nan ? 1 : 0);
    s->c_dirty = 1;
    return check_launch("kfx_sdf_reset_tracked");
}
