// raycast.hip -- per-pixel sphere-traced ray-march of the TSDF (roo::RaycastSdf) for gfx950.
//
// Reference behaviour: src/cu_raycast.cu:14-113 (PhongShade, KernRaycastSdf) with the
// samplers of BoundedVolume.h:93-106 / Volume.h:224-295.  New kernel: a wave64 owns a
// 32x2 pixel tile (coherent rays: the kernel is bound by the number of distinct 64-byte lines a
// wave touches per step, and cells are contiguous along x -- 8x8 tiles were 18 % slower, A/B in
// scripts/raycast_ab.py), a workgroup is 2x2 such tiles; each trilinear sample is four 16-byte
// loads (the x and x+1 cells of an AoS row are contiguous) instead of eight 8-byte ones;
// the gradient stencil is 20 distinct cells instead of 32 loads; missed rays skip the
// (discarded) normal evaluation of the reference.
#include <cmath>
#include <cstdlib>

#include <hip/hip_fp16.h>

#include "kfx_device.h"
#include "sampling.h"

namespace kfx {

struct RayParams {
    VolView vol;
    V3 size;          // bbox.Size()
    V3 dims1;         // (w-1.f, h-1.f, d-1.f)           Volume.h:226
    V3 hi2;           // ((float)(w-2), (float)(h-2), (float)(d-2))   Volume.h:229-231
    V3 voxel;         // VoxelSizeUnits()                BoundedVolume.h:67-76
    V3 inv_size;      // 1 / size, for div_uniform
    int fastdiv, off32;
    Pose T;           // T_wc
    Intr K;
    unsigned char *dptr, *nptr, *iptr;
    size_t dpitch, npitch, ipitch;
    int w, h;
    float near, far, trunc;
    int subpix;
    int tile_log2w;   // log2 of the wave's pixel-tile width (3: 8x8, 4: 16x4, 5: 32x2)
    int wg_log2x;     // log2 of the number of wave tiles side by side in a workgroup (0: 1x4, 1: 2x2, 2: 4x1)
    int sparse_lanes; // 0, or the number of lanes per wave that carry rays (small images; KFX_RAYCAST_LANES = 8 / 16 / 32 forces it, -1 disables it)
};

// PhongShade (cu_raycast.cu:14-28)
__device__ __forceinline__ float phong(const V3 p_c, const V3 n_c)
{
    const float ambient = (float)0.4, diffuse = (float)0.4, specular = (float)0.2;
    const V3 eyedir = div_s(p_c * -1.0f, length(p_c));
    const V3 l0 = v3((float)0.4, (float)0.4, -1.0f);
    const V3 lightdir = div_s(l0, length(l0));
    const float ldotn = dot(lightdir, n_c);
    const V3 lightreflect = n_c * (2 * ldotn) + lightdir * -1.0f;
    const float edotr = fmaxf(0.0f, dot(eyedir, lightreflect));
    const float spec = edotr * edotr * edotr * edotr * edotr * edotr * edotr * edotr * edotr * edotr;
    return ambient + diffuse * ldotn + specular * spec;
}

// pixel of thread `tid` of workgroup (bx, by): wave -> (1 << tile_log2w) x (64 >> tile_log2w) pixel tile,
// workgroup -> 2 x 2 such tiles (wg_log2x = 1)
__device__ __forceinline__ void ray_pixel_of(const RayParams& p, int bx, int by, int tid, int& u, int& v)
{
    const int lane = tid & 63, wv = tid >> 6;
    const int tw = 1 << p.tile_log2w, th = 64 >> p.tile_log2w;
    const int wgx = 1 << p.wg_log2x, wgy = 4 >> p.wg_log2x; // waves per workgroup along x / y
    u = (bx * wgx + (wv & (wgx - 1))) * tw + (lane & (tw - 1));
    v = (by * wgy + (wv >> p.wg_log2x)) * th + (lane >> p.tile_log2w);
}

// one ray: KernRaycastSdf (cu_raycast.cu:34-113) for pixel (u, v).
template <typename CELL, bool COLOR>
__device__ __forceinline__ float raycast_pixel(const RayParams& p, const ColorGeom& cv, const int u, const int v)
{
    if (u >= p.w || v >= p.h) return 0.f;

    const V3 c_w = v3(p.T.m[3], p.T.m[7], p.T.m[11]);                              // SE3Translation
    const V3 ray_c = v3(((float)u - p.K.u0) / p.K.fu, ((float)v - p.K.v0) / p.K.fv, 1.0f); // Unproject
    const V3 ray_w = so3_mul(p.T, ray_c);

    // slab test against the volume's box (cu_raycast.cu:46-51)
    const V3 ta = div_cw(p.vol.bmin - c_w, ray_w);
    const V3 tb = div_cw(p.vol.bmax - c_w, ray_w);
    const V3 tmin = v3(fminf(ta.x, tb.x), fminf(ta.y, tb.y), fminf(ta.z, tb.z));
    const V3 tmax = v3(fmaxf(ta.x, tb.x), fmaxf(ta.y, tb.y), fmaxf(ta.z, tb.z));
    const float max_tmin = fmaxf(fmaxf(fmaxf(tmin.x, tmin.y), tmin.z), p.near);
    const float min_tmax = fminf(fminf(fminf(tmax.x, tmax.y), tmax.z), p.far);

    float depth = 0.0f;
    if (max_tmin < min_tmax) {
        float lambda = max_tmin;
        float last_sdf = __builtin_nanf("");
        const float min_delta = p.voxel.x;
        float delta = 0.f;
        while (lambda < min_tmax) {
            const float sdf = trilinear<CELL>(p, c_w + ray_w * lambda);
            if (sdf <= 0) {
                if (last_sdf > 0) {
                    if (p.subpix) lambda = lambda + delta * sdf / (last_sdf - sdf);
                    depth = lambda;
                }
                break;
            }
            delta = march_step(sdf, min_delta, p.trunc);
            lambda += delta;
            last_sdf = sdf;
        }
    }

    float* pd = reinterpret_cast<float*>(p.dptr + (size_t)v * p.dpitch) + u;
    float* pi = reinterpret_cast<float*>(p.iptr + (size_t)v * p.ipitch) + u;
    float4* pn = reinterpret_cast<float4*>(p.nptr + (size_t)v * p.npitch) + u;
    if (depth > 0) {
        const V3 g = gradient<CELL>(p, c_w + ray_w * depth);
        const float len = length(g);
        const V3 n_w = len > 0 ? div_s(g, len) : v3(0.f, 0.f, 1.f);
        const V3 n_c = so3_mul_inv(p.T, n_w);
        const V3 p_c = ray_c * depth;
        *pd = depth;
        // colour variant: img = colorVol.GetUnitsTrilinearClamped(pos_w) instead of the Phong shade (cu_raycast.cu:172,179)
        if constexpr (COLOR) *pi = trilinear<RayC32>(cv, c_w + ray_w * depth);
        else *pi = phong(p_c, n_c);
        *pn = make_float4(n_c.x, n_c.y, n_c.z, 1.0f);
    } else {
        *pd = __builtin_nanf("");
        *pi = 0.f;
        *pn = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return depth > 0 ? depth : __builtin_nanf(""); // the value written to the depth image
}


// ---------------------------------------------------------------------------------------
// The march through the class tables (ClassView, kfx_device.h).  Every workgroup stages the tables -- a
// 32^3-cell level and a finer one, two bits per entry -- in LDS.  In an iteration a ray does ONE of two things:
//   * it consults the tables (LDS and arithmetic only): if the entry of its position says what a sample there would be --
//     trunc, NaN, or "trunc or NaN" -- the reference's step is known, and with it every following step that still starts
//     inside the entry: the whole run is taken at once (exact numerics: the reference's own lambda += delta additions, one
//     per step; fast numerics: one multiply-add);
//   * or it samples, exactly as the plain kernel does.
// The sampling lanes of a wave request their cells first and the consulting lanes work while those loads are in flight.
// The entry of a position comes from an affine estimate of the base-cell coordinate, pf(lambda) = A + B lambda: three FMAs
// instead of the ~30 separately rounded operations of cell_of().  The estimate is within `eps` cells of cell_of()'s
// coordinate (bound evaluated on the host from the box, the camera and the roundings of both, class_view()), a position
// closer than eps to a cell boundary -- where the two could disagree about the base cell -- is sampled, with cell_of()
// itself, and runs end eps inside the entry: the skipped steps are exactly those of the reference march, so with the
// exact-numerics tables (tol = 0: cells bit-equal to trunc, or NaN) depth, normals and shade stay bit-identical to the
// plain march.
// ---------------------------------------------------------------------------------------
// The coarser levels of the workgroup (ClassView::top_n of them: 64^3 and 128^3 cells, derived in LDS by classes_stage), described
// where the march can read them without holding them in registers across its loop.
struct TopLevels { ClassLevel lv[2]; int n; };

__device__ __forceinline__ int class_lookup(const unsigned* tab, const ClassLevel& L, int gx, int gy, int gz)
{
    const int bx = gx >> L.shift, by = gy >> L.shift, bz = gz >> L.shift;
    const uint2 w = *reinterpret_cast<const uint2*>(tab + L.first + (bz * L.ny + by) * L.rw + ((bx >> 5) << 1));
    return (int)((w.x >> (bx & 31)) & 1u) | (int)(((w.y >> (bx & 31)) & 1u) << 1);
}

// COUNT (kfx_raycast_sdf_count_tracked): the same march with every cell it reads marked in a bitmap and its samples, table
// look-ups and hits counted in cnt[] = {samples, look-ups, hit, newly marked cells}; writes no image.
__device__ __forceinline__ unsigned touch(unsigned* bitmap, const VolView& v, int x, int y, int z);
template <typename CELL, bool COLOR, bool COUNT = false>
__device__ __forceinline__ float raycast_pixel_classes(const RayParams& p, const RayParams& q, const ColorGeom& cv, const int u, const int v, const ClassView& cl, const unsigned* tab,
                                                       const TopLevels& top, unsigned* bitmap = nullptr, unsigned* cnt = nullptr)
{
    // p: the launch parameters as kernel arguments (scalar registers); q: the workgroup's copy of them in LDS, read by the
    // epilogue -- pose, intrinsics, output images and the gradient's geometry are then not held in scalar registers across the
    // march (the loop's own uniforms fill the register file: 22 spilled SGPRs, each a v_readlane in the loop, without this)
    if (u >= p.w || v >= p.h) return 0.f;

    const V3 c_w = v3(p.T.m[3], p.T.m[7], p.T.m[11]);
    const V3 ray_c = v3(((float)u - p.K.u0) / p.K.fu, ((float)v - p.K.v0) / p.K.fv, 1.0f);
    const V3 ray_w = so3_mul(p.T, ray_c);
    const V3 ta = div_cw(p.vol.bmin - c_w, ray_w);
    const V3 tb = div_cw(p.vol.bmax - c_w, ray_w);
    const V3 tmin = v3(fminf(ta.x, tb.x), fminf(ta.y, tb.y), fminf(ta.z, tb.z));
    const V3 tmax = v3(fmaxf(ta.x, tb.x), fmaxf(ta.y, tb.y), fmaxf(ta.z, tb.z));
    const float max_tmin = fmaxf(fmaxf(fmaxf(tmin.x, tmin.y), tmin.z), p.near);
    const float min_tmax = fminf(fminf(fminf(tmax.x, tmax.y), tmax.z), p.far);

    float depth = 0.0f;
    if (max_tmin < min_tmax) {
        float lambda = max_tmin;
        float last_sdf = __builtin_nanf("");
        const float min_delta = p.voxel.x;
        float delta = 0.f;
        // base-cell coordinate along the ray (an estimate: see above), and what leaving an entry costs per axis
        const V3 sc = v3(p.dims1.x / p.size.x, p.dims1.y / p.size.y, p.dims1.z / p.size.z);
        const V3 pfA = v3((c_w.x - p.vol.bmin.x) * sc.x, (c_w.y - p.vol.bmin.y) * sc.y, (c_w.z - p.vol.bmin.z) * sc.z);
        const V3 pfB = v3(ray_w.x * sc.x, ray_w.y * sc.y, ray_w.z * sc.z);
        const V3 inv = v3(1.0f / pfB.x, 1.0f / pfB.y, 1.0f / pfB.z);   // +-inf for a ray parallel to an axis plane: that axis never ends a run
        const float eps = cl.eps;
        const float step_free = fmaxf(cl.vref, min_delta);               // the reference's delta for sdf = vref
        const float inv_step_free = 1.0f / step_free, inv_trunc = 1.0f / p.trunc;
        const float band = cl.tol * cl.vref;
        bool pending = false;      // last_sdf is "vref or NaN": the last step left an entry of class 3
        float lambda_prev = 0.f;   // where that step started
        // The tables are consulted where they can help: at the start, after a run, and after a sample that came back as vref
        // or NaN (free or unseen space); a sample with any other value lies in an entry of class 0.  After a class-0 answer the
        // ray first leaves that entry (lam_retry).  A ray through a band of mixed values therefore marches as the plain kernel.
        bool consult = true;
        float lam_retry = lambda;
        int wait = 0, backoff = 1;   // samples to let pass before the next look at the tables: doubles with every class-0 answer
        // One iteration of a wave: the lanes that sample request their cells; the lanes that consult the tables do so (LDS and
        // arithmetic only) while those loads are in flight; then the sampling lanes wait, blend and step.  A lane whose
        // entry turns out to be class 0 samples in the next iteration.
        while (lambda < min_tmax) {
            const bool look = consult && wait == 0 && lambda >= lam_retry;
            CellPos c{};
            RayF32::InFlight fl;
            if (!look) {
                c = cell_of(p, c_w + ray_w * lambda);
                trilinear_issue(fl, p, c);
                if constexpr (COUNT) {
                    cnt[0] += 1;
                    for (int k = 0; k < 8; ++k) cnt[3] += touch(bitmap, p.vol, c.ix + (k & 1), c.iy + ((k >> 1) & 1), c.iz + (k >> 2));
                }
            }
            if (look) {
                if constexpr (COUNT) cnt[1] += 1;
                const float ex = __builtin_fmaf(pfB.x, lambda, pfA.x), ey = __builtin_fmaf(pfB.y, lambda, pfA.y), ez = __builtin_fmaf(pfB.z, lambda, pfA.z);
                const float flx = floorf(ex), fly = floorf(ey), flz = floorf(ez);
                const float lo_f = fminf(fminf(ex - flx, ey - fly), ez - flz), hi_f = fmaxf(fmaxf(ex - flx, ey - fly), ez - flz);
                // inside [0, dims - 1) with the margin on every axis: the base cell is floor(pf), no clamp involved; and at least
                // eps away from every cell boundary: cell_of() finds the same base cell
                bool run = false;
                if (fminf(fminf(ex, ey), ez) > eps && ex < p.dims1.x - eps && ey < p.dims1.y - eps && ez < p.dims1.z - eps && lo_f > eps && hi_f < 1.0f - eps) {
                    const int gx = (int)flx + cl.ox, gy = (int)fly + cl.oy, gz = (int)flz + cl.oz;
                    // the fine level answers "sample here?"; only a positive answer is worth the second look that may extend
                    // the run to the whole 32^3-cell entry
                    int cls = class_lookup(tab, cl.fine, gx, gy, gz), shift = cl.fine.shift;
                    if (cls == 3 && !cl.amb_ok) cls = 0;
                    if (cls != 0 && cl.fine.shift < 5) {
                        int c5 = class_lookup(tab, cl.coarse, gx, gy, gz);
                        if (c5 == 3 && !cl.amb_ok) c5 = 0;
                        if (c5 != 0) { cls = c5; shift = 5; }
                    }
                    // ... and a run through 32^3 cells may be one through 64^3 or 128^3 (the levels the workgroup derived in LDS):
                    // wide free or never-observed space is crossed in a third of the look-ups
                    if (shift == 5 && cl.top_n > 0) {
                        int c6 = class_lookup(tab, top.lv[0], gx, gy, gz);
                        if (c6 == 3 && !cl.amb_ok) c6 = 0;
                        if (c6 != 0) {
                            cls = c6; shift = 6;
                            if (cl.top_n > 1) {
                                int c7 = class_lookup(tab, top.lv[1], gx, gy, gz);
                                if (c7 == 3 && !cl.amb_ok) c7 = 0;
                                if (c7 != 0) { cls = c7; shift = 7; }
                            }
                        }
                    }
                    // the entry's cells are [lo, lo + L) per axis in the view's coordinates; its far side along the ray, pulled in
                    // by the margin: positions up to there certainly have their base cell in the entry
                    const float L = (float)(1 << shift);
                    const float lox = (float)(((gx >> shift) << shift) - cl.ox), loy = (float)(((gy >> shift) << shift) - cl.oy),
                                loz = (float)(((gz >> shift) << shift) - cl.oz);
                    const float bx = pfB.x >= 0.f ? lox + L - eps : lox + eps, by = pfB.y >= 0.f ? loy + L - eps : loy + eps,
                                bz = pfB.z >= 0.f ? loz + L - eps : loz + eps;
                    const float lam_exit = fminf(fminf((bx - pfA.x) * inv.x, (by - pfA.y) * inv.y), (bz - pfA.z) * inv.z);
                    if (cls != 0) {
                        // the reference's step for a sample of vref (class 1) or NaN (2; 3: the two steps are equal), taken from
                        // this position and from the following ones that still start inside the entry (and the box): n steps,
                        // n - 1 <= (lam_last - lambda) / delta less a hundredth for the quotient's rounding
                        delta = cls == 1 ? step_free : p.trunc;
                        pending = cls == 3;
                        last_sdf = cls == 1 ? cl.vref : __builtin_nanf("");
                        const float lam_last = fminf(lam_exit, min_tmax);
                        const float more = fminf(floorf((lam_last - lambda) * (cls == 1 ? inv_step_free : inv_trunc) - 0.01f), 4096.f);
                        if (cl.tol > 0.f) {
                            // fast numerics: the run in one multiply-add (the sum differs from n separately rounded additions by a
                            // few ulp of lambda: far inside the mode's tolerance)
                            const float nm = fmaxf(more, 0.f);
                            lambda_prev = __builtin_fmaf(nm, delta, lambda);
                            lambda = lambda_prev + delta;
                        } else {
                            // exact numerics: the reference's own additions, one per step
                            lambda_prev = lambda;
                            lambda += delta;
                            for (int k = (int)more; k > 0; --k) { lambda_prev = lambda; lambda += delta; }
                        }
                        backoff = 1;
                        run = true;
                    } else {
                        lam_retry = lam_exit;   // class 0: nothing to learn before the ray has left this entry
                    }
                }
                if (!run) {
                    wait = backoff;
                    backoff = min(backoff * 2, 32);
                }
            }
            if (!look) {
                const float sdf = trilinear_finish(fl, c);
                wait = max(wait - 1, 0);
                if (sdf <= 0) {
                    // a crossing needs the previous sample's value: the one thing a class-3 step left open
                    if (pending) {
                        last_sdf = trilinear<CELL>(p, c_w + ray_w * lambda_prev);
                        if constexpr (COUNT) {
                            const CellPos cp = cell_of(p, c_w + ray_w * lambda_prev);
                            cnt[0] += 1;
                            for (int k = 0; k < 8; ++k) cnt[3] += touch(bitmap, p.vol, cp.ix + (k & 1), cp.iy + ((k >> 1) & 1), cp.iz + (k >> 2));
                        }
                    }
                    if (last_sdf > 0) {
                        if (p.subpix) lambda = lambda + delta * sdf / (last_sdf - sdf);
                        depth = lambda;
                    }
                    break;
                }
                delta = march_step(sdf, min_delta, p.trunc);
                lambda += delta;
                last_sdf = sdf;
                pending = false;
                consult = !(fabsf(sdf - cl.vref) > band);   // vref (within the tables' tolerance) or NaN
            }
        }
    }

    if constexpr (COUNT) {   // the gradient's cells of a hit (the 20 of {-1, 0, 1}^3 with at most one coordinate at -1: sampling.h)
        if (depth > 0) {
            cnt[2] += 1;
            const V3 pos_v = div_cw(c_w + ray_w * depth - p.vol.bmin, p.size);
            const int ix = (int)fmaxf(fminf(p.hi2.x, floorf(pos_v.x * p.dims1.x)), 1.f), iy = (int)fmaxf(fminf(p.hi2.y, floorf(pos_v.y * p.dims1.y)), 1.f),
                      iz = (int)fmaxf(fminf(p.hi2.z, floorf(pos_v.z * p.dims1.z)), 1.f);
            for (int dz = -1; dz < 2; ++dz)
                for (int dy = -1; dy < 2; ++dy)
                    for (int dx = -1; dx < 2; ++dx)
                        if ((dx < 0) + (dy < 0) + (dz < 0) <= 1) cnt[3] += touch(bitmap, p.vol, ix + dx, iy + dy, iz + dz);
        }
        return depth > 0 ? depth : __builtin_nanf("");
    }
    float* pd = reinterpret_cast<float*>(q.dptr + (size_t)v * q.dpitch) + u;
    float* pi = reinterpret_cast<float*>(q.iptr + (size_t)v * q.ipitch) + u;
    float4* pn = reinterpret_cast<float4*>(q.nptr + (size_t)v * q.npitch) + u;
    if (depth > 0) {
        // the ray again, from the LDS copy: the same expressions on the same values
        const V3 cq = v3(q.T.m[3], q.T.m[7], q.T.m[11]);
        const V3 rcq = v3(((float)u - q.K.u0) / q.K.fu, ((float)v - q.K.v0) / q.K.fv, 1.0f);
        const V3 rq = so3_mul(q.T, rcq);
        const V3 g = gradient<CELL>(q, cq + rq * depth);
        const float len = length(g);
        const V3 n_w = len > 0 ? div_s(g, len) : v3(0.f, 0.f, 1.f);
        const V3 n_c = so3_mul_inv(q.T, n_w);
        const V3 p_c = rcq * depth;
        *pd = depth;
        if constexpr (COLOR) *pi = trilinear<RayC32>(cv, cq + rq * depth);
        else *pi = phong(p_c, n_c);
        *pn = make_float4(n_c.x, n_c.y, n_c.z, 1.0f);
    } else {
        *pd = __builtin_nanf("");
        *pi = 0.f;
        *pn = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return depth > 0 ? depth : __builtin_nanf("");
}

// one derived level: entry (bx, by, bz) = combination of the 2 x 2 x 2 entries of `src` below it (edge entries repeat a
// neighbour: same verdict).  One entry per thread and round: rows are padded to a power of two (or to whole waves when they are
// longer than a wave), so a wave's 64 lanes hold whole rows side by side, the two bit planes are its ballots, and the first
// lane of each row writes the row's share of them.
__device__ __forceinline__ void classes_level_up(unsigned* tab, const ClassLevel& src, int snx, int snz, const ClassLevel& dst, int dnx, int dnz)
{
    const int lane = threadIdx.x & 63;
    int lg = 0;                                      // log2 of the padded row length, at most 6 ...
    while ((1 << lg) < dnx && lg < 6) ++lg;
    const int chunks = (dnx + 63) >> 6;              // ... rows longer than a wave take `chunks` waves
    const int px = chunks > 1 ? chunks << 6 : 1 << lg;
    const int total = px * dst.ny * dnz;
    for (int e0 = 0; e0 < total; e0 += 256) {        // uniform: every wave runs the same number of rounds (ballots below)
        const int e = e0 + (int)threadIdx.x;
        const int bx = chunks > 1 ? e % px : e & (px - 1), r = chunks > 1 ? e / px : e >> lg;
        const int by = r % dst.ny, bz = r / dst.ny;
        int cls = 0;
        if (e < total && bx < dnx) {
            bool all_free = true, all_nan = true, all_either = true;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int x = min(2 * bx + (k & 1), snx - 1), y = min(2 * by + ((k >> 1) & 1), src.ny - 1), z = min(2 * bz + (k >> 2), snz - 1);
                const uint2 w = *reinterpret_cast<const uint2*>(tab + src.first + (z * src.ny + y) * src.rw + ((x >> 5) << 1));
                const int c = (int)((w.x >> (x & 31)) & 1u) | (int)(((w.y >> (x & 31)) & 1u) << 1);
                all_free = all_free && c == 1;
                all_nan = all_nan && c == 2;
                all_either = all_either && c != 0;
            }
            cls = all_free ? 1 : (all_nan ? 2 : (all_either ? 3 : 0));
        }
        const unsigned long long p0 = __ballot(cls & 1), p1 = __ballot(cls & 2);
        if (e < total && (bx & 63) == 0 && (chunks > 1 || bx == 0)) {   // first lane of a row (or of a row's 64-entry chunk)
            const int sh = chunks > 1 ? 0 : lane;                        // where the row's bits start in the ballots
            const unsigned long long m = (chunks > 1 || lg == 6) ? ~0ull : ((1ull << (1 << lg)) - 1ull);
            const unsigned long long r0 = (p0 >> sh) & m, r1 = (p1 >> sh) & m;
            unsigned* out = tab + dst.first + (bz * dst.ny + by) * dst.rw + (bx >> 6) * 4;
            out[0] = (unsigned)r0; out[1] = (unsigned)r1;
            if ((bx >> 6) * 4 + 2 < dst.rw) { out[2] = (unsigned)(r0 >> 32); out[3] = (unsigned)(r1 >> 32); }
        }
    }
}

// workgroup prologue of the class-table kernels: the tables into LDS (16-byte loads, all in flight together), then the
// coarser levels derived from the 32^3-cell level
__device__ __forceinline__ void classes_stage(const ClassView& cl, unsigned* tab, TopLevels& top)
{
    const uint4* src = reinterpret_cast<const uint4*>(cl.C);
    uint4* dst = reinterpret_cast<uint4*>(tab);
    for (int i = threadIdx.x; i < (cl.words >> 2); i += blockDim.x) dst[i] = src[i];
    ClassLevel l6, l7;
    int nx6, nz6, nx7, nz7, w6, w7;
    class_level_up(cl.nx5, cl.coarse.ny, cl.nz5, 6, cl.words, l6, nx6, nz6, w6);
    class_level_up(nx6, l6.ny, nz6, 7, cl.words + w6, l7, nx7, nz7, w7);
    if (threadIdx.x == 0) { top.lv[0] = l6; top.lv[1] = l7; top.n = cl.top_n; }
    __syncthreads();
    if (cl.top_n > 0) {   // launch-uniform
        classes_level_up(tab, cl.coarse, cl.nx5, cl.nz5, l6, nx6, nz6);
        __syncthreads();
        if (cl.top_n > 1) {
            classes_level_up(tab, l6, nx6, nz6, l7, nx7, nz7);
            __syncthreads();
        }
    }
}

template <typename CELL>
__global__ __launch_bounds__(256) void k_raycast_sdf_classes(const RayParams p, const ClassView cl)
{
    extern __shared__ unsigned s_tab[];
    __shared__ RayParams s_p;
    __shared__ TopLevels s_top;
    if (threadIdx.x == 0) s_p = p;
    classes_stage(cl, s_tab, s_top);   // (barriers inside)
    int u, v;
    if (p.sparse_lanes) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane >= p.sparse_lanes) return;
        u = (blockIdx.x * 2 + (wv & 1)) * p.sparse_lanes + lane;
        v = blockIdx.y * 2 + (wv >> 1);
    } else {
        ray_pixel_of(p, blockIdx.x, blockIdx.y, threadIdx.x, u, v);
    }
    raycast_pixel_classes<CELL, false>(p, s_p, ColorGeom{}, u, v, cl, s_tab, s_top);
}

template <typename CELL, bool COLOR>
__global__ __launch_bounds__(256) void k_raycast_sdf(const RayParams p, const ColorGeom cv)
{
    int u, v;
    if (p.sparse_lanes) { // only the first sparse_lanes lanes of a wave carry rays (a strip of one pixel row): small images
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane >= p.sparse_lanes) return;
        u = (blockIdx.x * 2 + (wv & 1)) * p.sparse_lanes + lane;
        v = blockIdx.y * 2 + (wv >> 1);
    } else {
        ray_pixel_of(p, blockIdx.x, blockIdx.y, threadIdx.x, u, v);
    }
    raycast_pixel<CELL, COLOR>(p, cv, u, v);
}

// Diagnostics (kfx_raycast_sdf_count): the march of k_raycast_sdf with every cell a sample or a hit's gradient stencil reads
// marked in a bitmap (one bit per voxel, dense x-fastest index): U = distinct voxels touched is the figure SURVEY.md 8(d)
// prices RaycastSdf's algorithmic bytes with (8 B x U + 24 B x w h).  counters: {samples, rays that enter the box, hits, U}.
__device__ __forceinline__ unsigned touch(unsigned* bitmap, const VolView& v, int x, int y, int z)
{
    const size_t i = ((size_t)z * v.h + y) * v.w + x;
    const unsigned bit = 1u << (i & 31);
    unsigned* word = bitmap + (i >> 5);
    if (*reinterpret_cast<volatile unsigned*>(word) & bit) return 0u;
    return (atomicOr(word, bit) & bit) ? 0u : 1u;
}
template <typename CELL>
__global__ __launch_bounds__(256) void k_raycast_sdf_count(const RayParams p, unsigned* __restrict__ bitmap, unsigned long long* __restrict__ counters)
{
    int u, v;
    ray_pixel_of(p, blockIdx.x, blockIdx.y, threadIdx.x, u, v);
    unsigned n_samples = 0, n_new = 0, entered = 0, hit = 0;
    if (u < p.w && v < p.h) {
        const V3 c_w = v3(p.T.m[3], p.T.m[7], p.T.m[11]);
        const V3 ray_c = v3(((float)u - p.K.u0) / p.K.fu, ((float)v - p.K.v0) / p.K.fv, 1.0f);
        const V3 ray_w = so3_mul(p.T, ray_c);
        const V3 ta = div_cw(p.vol.bmin - c_w, ray_w);
        const V3 tb = div_cw(p.vol.bmax - c_w, ray_w);
        const V3 tmin = v3(fminf(ta.x, tb.x), fminf(ta.y, tb.y), fminf(ta.z, tb.z));
        const V3 tmax = v3(fmaxf(ta.x, tb.x), fmaxf(ta.y, tb.y), fmaxf(ta.z, tb.z));
        const float max_tmin = fmaxf(fmaxf(fmaxf(tmin.x, tmin.y), tmin.z), p.near);
        const float min_tmax = fminf(fminf(fminf(tmax.x, tmax.y), tmax.z), p.far);
        float depth = 0.0f;
        if (max_tmin < min_tmax) {
            entered = 1;
            float lambda = max_tmin, last_sdf = __builtin_nanf(""), delta = 0.f;
            const float min_delta = p.voxel.x;
            while (lambda < min_tmax) {
                const CellPos c = cell_of(p, c_w + ray_w * lambda);
                for (int k = 0; k < 8; ++k) n_new += touch(bitmap, p.vol, c.ix + (k & 1), c.iy + ((k >> 1) & 1), c.iz + (k >> 2));
                n_samples += 1;
                const float sdf = trilinear_at<CELL>(p, c);
                if (sdf <= 0) {
                    if (last_sdf > 0) {
                        if (p.subpix) lambda = lambda + delta * sdf / (last_sdf - sdf);
                        depth = lambda;
                    }
                    break;
                }
                delta = march_step(sdf, min_delta, p.trunc);
                lambda += delta;
                last_sdf = sdf;
            }
        }
        if (depth > 0) { // the gradient's cells: {-1, 0, 1}^3 around its base cell with at most one coordinate at -1 (sampling.h)
            hit = 1;
            const V3 pos_v = div_cw(c_w + ray_w * depth - p.vol.bmin, p.size);
            const int ix = (int)fmaxf(fminf(p.hi2.x, floorf(pos_v.x * p.dims1.x)), 1.f), iy = (int)fmaxf(fminf(p.hi2.y, floorf(pos_v.y * p.dims1.y)), 1.f),
                      iz = (int)fmaxf(fminf(p.hi2.z, floorf(pos_v.z * p.dims1.z)), 1.f);
            for (int dz = -1; dz < 2; ++dz)
                for (int dy = -1; dy < 2; ++dy)
                    for (int dx = -1; dx < 2; ++dx)
                        if ((dx < 0) + (dy < 0) + (dz < 0) <= 1) n_new += touch(bitmap, p.vol, ix + dx, iy + dy, iz + dz);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n_samples += __shfl_xor(n_samples, off, 64); n_new += __shfl_xor(n_new, off, 64);
        entered += __shfl_xor(entered, off, 64); hit += __shfl_xor(hit, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (n_samples) atomicAdd(&counters[0], (unsigned long long)n_samples);
        if (entered) atomicAdd(&counters[1], (unsigned long long)entered);
        if (hit) atomicAdd(&counters[2], (unsigned long long)hit);
        if (n_new) atomicAdd(&counters[3], (unsigned long long)n_new);
    }
}

// the same for the march through the class tables: counters = {samples, rays that enter the box, hits, U, table look-ups, table bytes}
template <typename CELL>
__global__ __launch_bounds__(256) void k_raycast_sdf_classes_count(const RayParams p, const ClassView cl, unsigned* __restrict__ bitmap,
                                                                   unsigned long long* __restrict__ counters)
{
    extern __shared__ unsigned s_tab[];
    __shared__ RayParams s_p;
    __shared__ TopLevels s_top;
    if (threadIdx.x == 0) s_p = p;
    classes_stage(cl, s_tab, s_top);   // (barriers inside)
    int u, v;
    ray_pixel_of(p, blockIdx.x, blockIdx.y, threadIdx.x, u, v);
    unsigned cnt[4] = {0u, 0u, 0u, 0u}, entered = 0;
    if (u < p.w && v < p.h) {
        const V3 c_w = v3(p.T.m[3], p.T.m[7], p.T.m[11]);
        const V3 ray_c = v3(((float)u - p.K.u0) / p.K.fu, ((float)v - p.K.v0) / p.K.fv, 1.0f);
        const V3 ray_w = so3_mul(p.T, ray_c);
        const V3 ta = div_cw(p.vol.bmin - c_w, ray_w);
        const V3 tb = div_cw(p.vol.bmax - c_w, ray_w);
        const V3 tmin = v3(fminf(ta.x, tb.x), fminf(ta.y, tb.y), fminf(ta.z, tb.z));
        const V3 tmax = v3(fmaxf(ta.x, tb.x), fmaxf(ta.y, tb.y), fmaxf(ta.z, tb.z));
        entered = fmaxf(fmaxf(fmaxf(tmin.x, tmin.y), tmin.z), p.near) < fminf(fminf(fminf(tmax.x, tmax.y), tmax.z), p.far) ? 1u : 0u;
        raycast_pixel_classes<CELL, false, true>(p, s_p, ColorGeom{}, u, v, cl, s_tab, s_top, bitmap, cnt);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cnt[0] += __shfl_xor(cnt[0], off, 64); cnt[1] += __shfl_xor(cnt[1], off, 64);
        cnt[2] += __shfl_xor(cnt[2], off, 64); cnt[3] += __shfl_xor(cnt[3], off, 64);
        entered += __shfl_xor(entered, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (cnt[0]) atomicAdd(&counters[0], (unsigned long long)cnt[0]);
        if (entered) atomicAdd(&counters[1], (unsigned long long)entered);
        if (cnt[2]) atomicAdd(&counters[2], (unsigned long long)cnt[2]);
        if (cnt[3]) atomicAdd(&counters[3], (unsigned long long)cnt[3]);
        if (cnt[1]) atomicAdd(&counters[4], (unsigned long long)cnt[1]);
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicAdd(&counters[5], (unsigned long long)cl.words * 4ull);
}

// ---------------------------------------------------------------------------------------
// Several renderings of the same model in one launch.  The tracking loop raycasts the model at every pyramid level
// that has ICP iterations (main.cpp:280-288: 640x480, 160x120, 80x60), and a coarse level takes as long as the
// full-resolution one: the march is a chain of ~150 dependent misses whatever the number of rays (DESIGN.md 5.2;
// measured 0.168 / 0.147 / 0.167 ms for levels 0 / 2 / 3).  One grid over the workgroups of all levels lets the
// chains overlap: 0.465 -> 0.274 ms (S_room), 0.452 -> 0.255 ms (S_full), scripts/raycast_levels.py.  Each pixel runs raycast_pixel unchanged, so every image is
// bit-identical to its own kfx_raycast_sdf call.
// ---------------------------------------------------------------------------------------
constexpr int RAY_MAX_LEVELS = 8;
struct RayLevel {
    unsigned char *dptr, *nptr, *iptr;
    size_t dpitch, npitch, ipitch;
    int w, h;
    Intr K;
    unsigned char* vptr;       // optional vertex map of the rendering: DepthToVbo(depth, K) (cu_depth_tools.cu:59-70), or null
    size_t vpitch;
    int first_block, blocks_x; // this level's workgroups are [first_block, next level's first_block), row-major
    int sparse;                // 0, or rays per wave (a strip of one pixel row, the other lanes idle); workgroup = 2 x 2 strips
};
struct RayLevels {
    RayLevel lv[RAY_MAX_LEVELS];
    int n;
};

template <typename CELL>
__global__ __launch_bounds__(256) void k_raycast_sdf_levels(const RayParams base, const RayLevels L)
{
    int l = 0;
    for (int k = 1; k < L.n; ++k)
        if ((int)blockIdx.x >= L.lv[k].first_block) l = k; // uniform
    const RayLevel& lv = L.lv[l];
    RayParams p = base;
    p.dptr = lv.dptr; p.nptr = lv.nptr; p.iptr = lv.iptr;
    p.dpitch = lv.dpitch; p.npitch = lv.npitch; p.ipitch = lv.ipitch;
    p.w = lv.w; p.h = lv.h;
    p.K = lv.K;
    const int b = (int)blockIdx.x - lv.first_block;
    int u, v;
    if (lv.sparse) {
        // coarse levels: neighbouring rays are many voxels apart, so every lane of a load fetches its own line and a
        // wave-step waits for the slowest of 64 misses; with `sparse` rays per wave it waits for the slowest of those
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane >= lv.sparse) return;
        u = ((b % lv.blocks_x) * 2 + (wv & 1)) * lv.sparse + lane;
        v = (b / lv.blocks_x) * 2 + (wv >> 1);
    } else {
        ray_pixel_of(p, b % lv.blocks_x, b / lv.blocks_x, threadIdx.x, u, v);
    }
    const float kz = raycast_pixel<CELL, false>(p, ColorGeom{}, u, v);
    if (lv.vptr && u < p.w && v < p.h) // the application's DepthToVbo(ray_v[l], ray_d[l], K[l]) (main.cpp:286), same expression
        reinterpret_cast<float4*>(lv.vptr + (size_t)v * lv.vpitch)[u] =
            make_float4(kz * ((float)u - p.K.u0) / p.K.fu, kz * ((float)v - p.K.v0) / p.K.fv, kz, 1.0f);
}

template <typename CELL>
__global__ __launch_bounds__(256) void k_raycast_sdf_levels_classes(const RayParams base, const RayLevels L, const ClassView cl)
{
    extern __shared__ unsigned s_tab[];
    __shared__ RayParams s_p;
    int l = 0;
    for (int k = 1; k < L.n; ++k)
        if ((int)blockIdx.x >= L.lv[k].first_block) l = k; // uniform
    const RayLevel& lv = L.lv[l];
    RayParams p = base;
    p.dptr = lv.dptr; p.nptr = lv.nptr; p.iptr = lv.iptr;
    p.dpitch = lv.dpitch; p.npitch = lv.npitch; p.ipitch = lv.ipitch;
    p.w = lv.w; p.h = lv.h;
    p.K = lv.K;
    __shared__ TopLevels s_top;
    if (threadIdx.x == 0) s_p = p;
    classes_stage(cl, s_tab, s_top);   // before any lane leaves (barriers inside)
    const int b = (int)blockIdx.x - lv.first_block;
    int u, v;
    if (lv.sparse) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane >= lv.sparse) return;
        u = ((b % lv.blocks_x) * 2 + (wv & 1)) * lv.sparse + lane;
        v = (b / lv.blocks_x) * 2 + (wv >> 1);
    } else {
        ray_pixel_of(p, b % lv.blocks_x, b / lv.blocks_x, threadIdx.x, u, v);
    }
    const float kz = raycast_pixel_classes<CELL, false>(p, s_p, ColorGeom{}, u, v, cl, s_tab, s_top);
    if (lv.vptr && u < p.w && v < p.h)
        reinterpret_cast<float4*>(lv.vptr + (size_t)v * lv.vpitch)[u] =
            make_float4(kz * ((float)u - p.K.u0) / p.K.fu, kz * ((float)v - p.K.v0) / p.K.fv, kz, 1.0f);
}

// ---------------------------------------------------------------------------------------
// Exact multi-GPU march ("exact variant").  The volume is split into Z-slabs; a ray's
// march state (lambda, last_sdf, delta) is carried from slab to slab in ray order, so every sample is
// taken at exactly the position, and from exactly the cells, of the single-volume march.  `p` describes
// the FULL volume's geometry with a virtual base pointer (local storage - z_offset planes), so
// trilinear() / gradient() address global plane indices unchanged.  A rank advances a ray while the
// trilinear base cell iz of the current sample lies in the cells it owns, [own_lo, own_hi), and the
// planes it needs are stored locally, [avail_lo, avail_hi).
// State: 9 dense planes of h*w floats (structure of arrays): 0 lambda, 1 last_sdf, 2 delta, 3 status,
// 4 touched-this-round, 5-7 normal, 8 shade.  status: 0 marching, 1 hit (final), 2 miss (final),
// 3 hit found, normal pending (computed by the rank that owns the gradient's base plane).  A hit's depth is
// its lambda.  Exactly one rank touches a pixel per round (ownership is disjoint), so the host merges the
// ranks' states with one integer SUM all-reduce of the touched pixels' planes 0-4.
// ---------------------------------------------------------------------------------------
struct SlabRay {
    // State of tile t (rows [t R, (t + 1) R) of the image, pixel q = (v - t R) w + u of it): march planes 0-3 at
    // state + (t * 4 + k) * P + q -- what travels with a ray: one contiguous message per tile --, normal / shade planes 5-8 at
    // result + (t * 4 + k - 5) * P + q.  touched (plane 4; null: not kept): the dense state's, for ONE tile.  One tile of R = h rows with
    // P = w h, touched = state + 4 P and result = state + 5 P is the dense [9][h w] layout of kfx_raycast_sdf_slab.
    // packed: THREE march planes per tile, at state + (t * 3 + k) * P + q: lambda, last_sdf and delta-or-status -- a marching ray's
    // delta (positive: the caller guarantees a positive truncation distance and voxel size), or -status of a ray that is not
    // marching (-1 hit, -2 miss, -3 hit awaiting its normal: such a ray's delta is never read again).  12 bytes per ray and hop
    // (SURVEY 8(e)); the received snapshots have the same layout.
    float* state;
    float* touched;
    float* result;
    int packed;
    // normals_here: a hit's normal is evaluated by whichever rank holds the hit (its finder, or a rank that adopted it) as soon as the
    // three planes of the gradient stencil are STORED there, ghost planes included -- they hold the owner's bits; else only by the
    // rank that OWNS the stencil's base plane (the hit travels there in the hand-over's last stage)
    int normals_here;
    size_t P;           // plane stride in pixels (>= R w)
    int R;              // rows per tile
    int v0, v1;         // rows this launch processes
    int* fin;           // optional, dense w h: set where THIS launch gives a ray its final status (and, with claim_misses, where
    int claim_misses;   //   the initialisation finds a ray that never enters the box: one rank answers for those)
    int own_lo, own_hi; // owned trilinear base cells (global plane indices)
    int avail_lo, avail_hi; // planes stored locally (global indices)
    int init;           // 1: (re)initialise the state from the ray / box intersection
    // Snapshots of the same rays received from the two neighbour ranks (march planes 0-3 as above; null: none): before marching, a
    // ray takes a neighbour's snapshot when that one is NEWER than its own and still under way (kfx_slab_raycast_exact_tiled).
    // adopt_tile_major: the buffers hold every tile ([tiles][4][P]) instead of just the tile of this launch ([4][P]).
    const float* adopt_lo;
    const float* adopt_hi;
    int adopt_tile_major;
};

// Snapshots of one march order by progress: status 1 / 2 (final) after 3 (hit, normal pending) after 0 (marching), and of two
// marching snapshots the one with the larger lambda is later (every step adds a positive delta).
__device__ __forceinline__ int snapshot_order(float status) { return status == 0.0f ? 0 : (status == 3.0f ? 1 : 2); }

__device__ __forceinline__ int cell_z(const RayParams& p, const V3 pos_w)
{
    const float pfz = ((pos_w.z - p.vol.bmin.z) / p.size.z) * p.dims1.z;
    return (int)fmaxf(fminf(p.hi2.z, floorf(pfz)), 0.f);
}
__device__ __forceinline__ int grad_cell_z(const RayParams& p, const V3 pos_w)
{
    const float pfz = ((pos_w.z - p.vol.bmin.z) / p.size.z) * p.dims1.z;
    return (int)fmaxf(fminf(p.hi2.z, floorf(pfz)), 1.f);
}

template <typename CELL>
__global__ __launch_bounds__(256) void k_raycast_sdf_slab(const RayParams p, const SlabRay sl)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int u = blockIdx.x * 64 + (wv & 1) * 32 + (lane & 31); // wave = 32 x 2 pixels, as k_raycast_sdf's default
    const int v = sl.v0 + blockIdx.y * 4 + (wv >> 1) * 2 + (lane >> 5);
    if (u >= p.w || v >= sl.v1) return;
    const size_t plane = sl.P;
    const int tile = v / sl.R;
    const size_t q = (size_t)(v - tile * sl.R) * p.w + u;
    const int NP = sl.packed ? 3 : 4;
    float* st = sl.state + (size_t)tile * NP * plane + q;   // plane k at st[k * plane]
    float* rs = sl.result + (size_t)tile * 4 * plane + q;  // normal x / y / z, shade

    const V3 c_w = v3(p.T.m[3], p.T.m[7], p.T.m[11]);
    const V3 ray_c = v3(((float)u - p.K.u0) / p.K.fu, ((float)v - p.K.v0) / p.K.fv, 1.0f);
    const V3 ray_w = so3_mul(p.T, ray_c);
    const V3 ta = div_cw(p.vol.bmin - c_w, ray_w);
    const V3 tb = div_cw(p.vol.bmax - c_w, ray_w);
    const V3 tmin = v3(fminf(ta.x, tb.x), fminf(ta.y, tb.y), fminf(ta.z, tb.z));
    const V3 tmax = v3(fmaxf(ta.x, tb.x), fmaxf(ta.y, tb.y), fmaxf(ta.z, tb.z));
    const float max_tmin = fmaxf(fmaxf(fmaxf(tmin.x, tmin.y), tmin.z), p.near);
    const float min_tmax = fminf(fminf(fminf(tmax.x, tmax.y), tmax.z), p.far);

    float lambda, last_sdf, delta, status;
    if (sl.init) {
        lambda = max_tmin;
        last_sdf = __builtin_nanf("");
        delta = 0.f;
        status = (max_tmin < min_tmax) ? 0.f : 2.f;
        rs[0] = 0.f; rs[plane] = 0.f; rs[2 * plane] = 0.f; rs[3 * plane] = 0.f;
        if (sl.fin) sl.fin[(size_t)v * p.w + u] = (sl.claim_misses && status == 2.f) ? 1 : 0;
    } else {
        lambda = st[0]; last_sdf = st[plane]; delta = st[2 * plane];
        if (sl.packed) { status = delta < 0.f ? -delta : 0.f; delta = delta < 0.f ? 0.f : delta; }
        else status = st[3 * plane];
    }
    // a neighbour's newer, still open snapshot of this ray replaces the rank's own (a stale copy is never advanced: its position
    // lies in planes of a rank the ray has left, and final snapshots stay with the rank that finalised them)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float* src = k ? sl.adopt_hi : sl.adopt_lo;
        if (!src) continue;
        src += (sl.adopt_tile_major ? (size_t)tile * NP * plane : (size_t)0) + q;
        const float n_code = src[2 * plane];
        const float n_status = sl.packed ? (n_code < 0.f ? -n_code : 0.f) : src[3 * plane];
        if (!(n_status == 0.f || n_status == 3.f)) continue;
        const float n_lambda = src[0];
        const int on = snapshot_order(n_status), om = snapshot_order(status);
        if (on > om || (on == 0 && om == 0 && n_lambda > lambda)) {
            lambda = n_lambda; last_sdf = src[plane]; delta = (sl.packed && n_code < 0.f) ? 0.f : n_code; status = n_status;
        }
    }
    const float lambda_in = lambda, status_in = status;

    if (status == 0.f) {
        const float min_delta = p.voxel.x;
        while (true) {
            if (!(lambda < min_tmax)) { status = 2.f; break; }
            const V3 pos = c_w + ray_w * lambda;
            const int iz = cell_z(p, pos);
            if (iz < sl.own_lo || iz >= sl.own_hi || iz < sl.avail_lo || iz + 1 >= sl.avail_hi) break; // another rank's sample
            const float sdf = trilinear<CELL>(p, pos);
            if (sdf <= 0) {
                if (last_sdf > 0) {
                    if (p.subpix) lambda = lambda + delta * sdf / (last_sdf - sdf);
                    status = 3.f;
                } else {
                    status = 2.f;
                }
                break;
            }
            delta = march_step(sdf, min_delta, p.trunc);
            lambda += delta;
            last_sdf = sdf;
        }
    }
    if (status == 3.f) { // hit: the rank owning the gradient's base plane gz evaluates the normal
        const V3 pos = c_w + ray_w * lambda;
        const int gz = grad_cell_z(p, pos);
        if ((sl.normals_here || (gz >= sl.own_lo && gz < sl.own_hi)) && gz - 1 >= sl.avail_lo && gz + 1 < sl.avail_hi) {
            const V3 g = gradient<CELL>(p, pos);
            const float len = length(g);
            const V3 n_w = len > 0 ? div_s(g, len) : v3(0.f, 0.f, 1.f);
            const V3 n_c = so3_mul_inv(p.T, n_w);
            rs[0] = n_c.x; rs[plane] = n_c.y; rs[2 * plane] = n_c.z;
            rs[3 * plane] = phong(ray_c * lambda, n_c);
            status = 1.f;
        }
    }
    const bool changed = lambda != lambda_in || status != status_in;
    st[0] = lambda; st[plane] = last_sdf;
    if (sl.packed) st[2 * plane] = status == 0.f ? delta : -status;
    else { st[2 * plane] = delta; st[3 * plane] = status; }
    if (sl.touched) sl.touched[q] = changed ? 1.0f : 0.0f;
    if (sl.fin && changed && (status == 1.f || status == 2.f)) sl.fin[(size_t)v * p.w + u] = 1;
}

// state -> the three output images (hit: depth / normal / shade; otherwise NaN / 0 / 0)
__global__ __launch_bounds__(256) void k_raycast_state_to_images(const RayParams p, const float* __restrict__ state)
{
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= p.w || v >= p.h) return;
    const size_t plane = (size_t)p.w * p.h;
    const float* st = state + (size_t)v * p.w + u;
    const float depth = st[0];
    const bool hit = st[3 * plane] == 1.f && depth > 0.f;
    *(reinterpret_cast<float*>(p.dptr + (size_t)v * p.dpitch) + u) = hit ? depth : __builtin_nanf("");
    *(reinterpret_cast<float*>(p.iptr + (size_t)v * p.ipitch) + u) = hit ? st[8 * plane] : 0.f;
    *(reinterpret_cast<float4*>(p.nptr + (size_t)v * p.npitch) + u) =
        hit ? make_float4(st[5 * plane], st[6 * plane], st[7 * plane], 1.0f) : make_float4(0.f, 0.f, 0.f, 0.f);
}

} // namespace kfx

using namespace kfx;

// argument checks of one output image set + volume, and the kernel parameters they give
template <typename CELL>
static int ray_params(RayParams& p, const kfx_image* depth, const kfx_image* norm, const kfx_image* img,
                      const kfx_volume* vol, const float T_wc[12], const float K[4], float near,
                      float far, float trunc_dist, int subpix)
{
    if (!depth || !norm || !img || !vol || !T_wc || !K || !depth->ptr || !norm->ptr || !img->ptr || !vol->ptr)
        return set_error(KFX_E_NULL, "RaycastSdf: null argument");
    if (depth->w < img->w || depth->h < img->h || norm->w < img->w || norm->h < img->h)
        return set_error(KFX_E_SHAPE, "RaycastSdf: output images smaller than img");
    if (depth->pitch < img->w * 4 || img->pitch < img->w * 4 || norm->pitch < img->w * 16)
        return set_error(KFX_E_SHAPE, "RaycastSdf: image pitch");
    if ((((uintptr_t)depth->ptr | depth->pitch | (uintptr_t)img->ptr | img->pitch) & 3) ||
        (((uintptr_t)norm->ptr | norm->pitch) & 15) || (((uintptr_t)vol->ptr | vol->pitch | vol->img_pitch) & (CELL::BYTES - 1)))
        return set_error(KFX_E_ALIGN, "RaycastSdf: alignment");
    // the gradient stencil reads cells [1-1, (dim-2)+1] (Volume.h:271-273)
    if (vol->w < 3 || vol->h < 3 || vol->d < 3 || vol->w > 65535 || vol->h > 65535 || vol->d > 65535)
        return set_error(KFX_E_SHAPE, "RaycastSdf: volume dimensions");
    if (vol->pitch < vol->w * CELL::BYTES || vol->img_pitch < vol->pitch * (vol->h - 1) + vol->w * CELL::BYTES)
        return set_error(KFX_E_SHAPE, "RaycastSdf: volume pitch");
    set_geometry(p, vol);
    set_voxel_size(p, vol);
    for (int i = 0; i < 12; ++i) p.T.m[i] = T_wc[i];
    p.K = Intr{K[0], K[1], K[2], K[3]};
    p.dptr = (unsigned char*)depth->ptr;
    p.nptr = (unsigned char*)norm->ptr;
    p.iptr = (unsigned char*)img->ptr;
    p.dpitch = depth->pitch;
    p.npitch = norm->pitch;
    p.ipitch = img->pitch;
    p.w = (int)img->w; // the reference bounds the launch by img (cu_raycast.cu:39,110)
    p.h = (int)img->h;
    p.near = near;
    p.far = far;
    p.trunc = trunc_dist;
    p.subpix = subpix ? 1 : 0;
    p.tile_log2w = 5; // 32 x 2 pixel wave tiles, 2 x 2 of them per workgroup
    p.wg_log2x = 1;
    p.sparse_lanes = 0;
    return 0;
}

// The class-table march's view of a summary for a launch on `vol` with camera T_wc: brings the tables up to date on `stream`,
// evaluates the margin that covers the affine cell estimate, reports the LDS bytes.  *usable = 0: march plainly (the margin
// would be too wide, or trunc is not positive).
static int class_view(ClassView& cl, size_t* lds_bytes, int* usable, kfx_sdf_summary* summary, const kfx_volume* vol, const RayParams& p, kfx_stream stream)
{
    *usable = 0;
    if (int e = summary_view_offset(summary, vol, &cl.ox, &cl.oy, &cl.oz)) return e;
    if (!(p.trunc > 0.f) || !(p.trunc < __builtin_inff())) return 0;
    static const int kb_env = [] { const char* e = getenv("KFX_RAYCAST_CLASS_KB"); const int v = e ? atoi(e) : 16; return v < 1 ? 1 : (v > 60 ? 60 : v); }();
    int fine = 3;
    for (; fine < 5; ++fine) {
        summary_class_layout(summary, fine, cl);
        if ((size_t)cl.words * 4 <= (size_t)kb_env * 1024) break;
    }
    summary_class_layout(summary, fine, cl);
    if ((size_t)cl.words * 4 > 60 * 1024) return 0;
    const float tol = math_mode() == KFX_MATH_FAST ? 1e-5f : 0.f;
    // Worth it?  The march through the tables costs ~8 % per sampled step (table look-ups, staging); it pays where a fair part
    // of the volume can be crossed without sampling: at least a quarter of the 32^3-cell entries of class != 0.  Every table
    // build publishes that count in a host-visible ring; the choice is made from the count of the build BEFORE THE PREVIOUS ONE
    // (two frames old: finished long ago, so the event wait below returns at once) -- it only steers a choice between two
    // kernels that render the same images, but in fast numerics "the same" means within tolerance, so the choice is made a
    // function of the sequence of calls, not of how far the GPU happens to have got (round-3 advice).  Exact numerics on a
    // running stream, where only never-observed space qualifies, fall back to the plain march this way; while the choice is
    // "plain" the tables are rebuilt on every 8th call only.  KFX_RAYCAST_SUMMARY=1 always uses the tables, -1 never.
    static const int force_env = [] { const char* e = getenv("KFX_RAYCAST_SUMMARY"); return e ? atoi(e) : 0; }();
    if (force_env < 0) return 0;
    const bool steer = force_env == 0 && summary->h_skippable;
    if (steer && summary->plain_calls && (summary->plain_calls++ % 8u) != 0u) return 0;   // still plain: no build, no look
    if (int e = summary_classes_prepare(summary, tol, p.trunc, fine, (hipStream_t)stream)) return e;
    if (steer && summary->builds >= 3) {
        const unsigned ref = (summary->builds - 3) % KFX_SUMMARY_RING;   // builds - 1 is the one just issued (or the last one)
        (void)hipEventSynchronize(summary->build_done[ref]);
        const int known = ((volatile int*)summary->h_skippable)[ref];
        if (known >= 0 && (long long)known * 4 < summary->n_coarse) {   // less than a quarter
            if (!summary->plain_calls) summary->plain_calls = 1;
            return 0;
        }
        summary->plain_calls = 0;
    }
    cl.C = summary->C;
    cl.vref = p.trunc;
    cl.amb_ok = p.trunc >= p.voxel.x ? 1 : 0;   // class 3 needs equal steps for a vref and a NaN sample: max(trunc, min_delta) = trunc
    cl.tol = tol;
    // margin: |cell_of()'s coordinate - the affine estimate| per axis, from the roundings of both (u = 2^-24):
    //   cell_of:   pos = c + ray lambda (2 roundings of magnitudes <= |pos| + |c|), - bmin, / size, * dims1
    //   estimate:  A = (c - bmin) sc (3 roundings), B = ray sc (2), fma(B, lambda, A) (1); and the run's far side
    //              (bound - A) / B, whose error in this axis' coordinate is 3 u (|bound| + |A|) whatever B is
    const double u24 = 1.0 / 16777216.0;
    double eps = 0.0;
    const float bmin[3] = {p.vol.bmin.x, p.vol.bmin.y, p.vol.bmin.z}, bmax[3] = {p.vol.bmax.x, p.vol.bmax.y, p.vol.bmax.z};
    const float size[3] = {p.size.x, p.size.y, p.size.z}, dims1[3] = {p.dims1.x, p.dims1.y, p.dims1.z};
    const float cw[3] = {p.T.m[3], p.T.m[7], p.T.m[11]};
    for (int a = 0; a < 3; ++a) {
        if (!(size[a] > 0.f)) return 0;
        const double sc = (double)dims1[a] / size[a];
        const double pmax = std::fmax(std::fabs((double)bmin[a]), std::fabs((double)bmax[a])), ca = std::fabs((double)cw[a]);
        const double A = std::fabs(((double)cw[a] - bmin[a]) * sc);
        const double t = sc * (3.0 * pmax + 2.0 * ca + size[a]) * u24 + 2.0 * dims1[a] * u24;
        const double e = u24 * (3.0 * A + 2.0 * sc * (pmax + ca) + dims1[a]) + 3.0 * u24 * (dims1[a] + A);
        eps = std::fmax(eps, t + e);
    }
    eps *= 2.0;   // twice the bound
    if (!(eps < 0.05)) return 0;
    cl.eps = (float)std::fmax(eps, 1e-4);
    // the coarser levels every workgroup derives in LDS (classes_stage): 64^3 cells where the 32^3-cell level has more than one
    // entry along some axis, 128^3 cells likewise on top of that
    cl.nx5 = ceil_div(summary->w, 32); cl.nz5 = ceil_div(summary->d, 32);
    static const int top_env = [] { const char* e = getenv("KFX_RAYCAST_TOP_LEVELS"); const int v = e ? atoi(e) : 2; return v < 0 ? 0 : (v > 2 ? 2 : v); }();
    ClassLevel l6, l7;
    int nx6, nz6, nx7, nz7, w6, w7;
    class_level_up(cl.nx5, cl.coarse.ny, cl.nz5, 6, cl.words, l6, nx6, nz6, w6);
    class_level_up(nx6, l6.ny, nz6, 7, cl.words + w6, l7, nx7, nz7, w7);
    cl.top_n = 0;
    if (cl.nx5 > 1 || cl.coarse.ny > 1 || cl.nz5 > 1) cl.top_n = 1;
    if (cl.top_n && (nx6 > 1 || l6.ny > 1 || nz6 > 1)) cl.top_n = 2;
    if (cl.top_n > top_env) cl.top_n = top_env;
    cl.lds_words = cl.words + (cl.top_n > 0 ? w6 : 0) + (cl.top_n > 1 ? w7 : 0);
    *lds_bytes = (size_t)cl.lds_words * sizeof(unsigned);
    // the launch asks for the staged tables PLUS the derived levels PLUS the kernels' static LDS (RayParams, TopLevels, the
    // levels kernel's table): the whole must stay below the 64 KiB a launch may have, or the launch fails where the plain march
    // would have worked (round-4 advice: the 60 KiB test above sees the staged words only)
    if (*lds_bytes + sizeof(RayParams) + sizeof(TopLevels) + 2048 > 64 * 1024) return 0;
    *usable = 1;
    return 0;
}

template <typename CELL>
static int raycast_levels_launch(int n_levels, const kfx_image* const* depth, const kfx_image* const* norm, const kfx_image* const* img,
                                 const kfx_image* const* vbo, const kfx_volume* vol, const float T_wc[12], const float* K, float near, float far,
                                 float trunc_dist, int subpix, kfx_stream stream, kfx_sdf_summary* summary = nullptr)
{
    if (n_levels < 0 || n_levels > RAY_MAX_LEVELS) return set_error(KFX_E_RANGE, "RaycastSdf(levels): number of levels");
    if (!depth || !norm || !img || !K) return set_error(KFX_E_NULL, "RaycastSdf(levels): null argument");
    RayParams base{};
    RayLevels L{};
    int blocks = 0;
    for (int l = 0; l < n_levels; ++l) { // (coarse levels first was tried: no faster)
        RayParams p;
        if (int e = ray_params<CELL>(p, depth[l], norm[l], img[l], vol, T_wc, K + 4 * l, near, far, trunc_dist, subpix)) return e;
        if (p.w == 0 || p.h == 0) continue; // an empty level launches nothing, as kfx_raycast_sdf
        RayLevel& lv = L.lv[L.n++];
        lv.dptr = p.dptr; lv.nptr = p.nptr; lv.iptr = p.iptr;
        lv.dpitch = p.dpitch; lv.npitch = p.npitch; lv.ipitch = p.ipitch;
        lv.w = p.w; lv.h = p.h;
        lv.K = p.K;
        lv.vptr = nullptr;
        lv.vpitch = 0;
        if (vbo && vbo[l]) {
            const kfx_image* vb = vbo[l];
            if (!vb->ptr) return set_error(KFX_E_NULL, "RaycastSdf(levels): null vertex map");
            if (vb->w < (size_t)p.w || vb->h < (size_t)p.h || vb->pitch < (size_t)p.w * 16) return set_error(KFX_E_SHAPE, "RaycastSdf(levels): vertex map smaller than img");
            if (((uintptr_t)vb->ptr | vb->pitch) & 15) return set_error(KFX_E_ALIGN, "RaycastSdf(levels): vertex map alignment");
            lv.vptr = (unsigned char*)vb->ptr;
            lv.vpitch = vb->pitch;
        }
        lv.first_block = blocks;
        static const int sparse_env = [] { const char* e = getenv("KFX_RAYCAST_SPARSE"); const int v = e ? atoi(e) : 16; return (v == 8 || v == 16 || v == 32) ? v : 0; }();
        lv.sparse = (long long)p.w * p.h <= 160 * 120 ? sparse_env : 0; // coarse levels only: at full resolution dense waves are faster
        lv.blocks_x = lv.sparse ? ceil_div(p.w, 2 * lv.sparse) : ceil_div(p.w, 64);
        blocks += lv.blocks_x * (lv.sparse ? ceil_div(p.h, 2) : ceil_div(p.h, 4));
        base = p;
    }
    if (L.n == 0) return 0;
    if (summary) {
        if constexpr (CELL::BYTES == 8) {
            ClassView cl;
            size_t cl_bytes = 0;
            int usable = 0;
            if (int e = class_view(cl, &cl_bytes, &usable, summary, vol, base, stream)) return e;
            if (usable) {
                hipLaunchKernelGGL((k_raycast_sdf_levels_classes<CELL>), dim3(blocks), dim3(256), cl_bytes, (hipStream_t)stream, base, L, cl);
                return check_launch("kfx_raycast_sdf_levels_tracked");
            }
        } else {
            return set_error(KFX_E_RANGE, "kfx_raycast_sdf_levels_tracked: fp32 cells only");
        }
    }
    hipLaunchKernelGGL((k_raycast_sdf_levels<CELL>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, base, L);
    return check_launch("kfx_raycast_sdf_levels");
}

template <typename CELL>
static int raycast_launch(const kfx_image* depth, const kfx_image* norm, const kfx_image* img,
                          const kfx_volume* vol, const float T_wc[12], const float K[4], float near,
                          float far, float trunc_dist, int subpix, kfx_stream stream, const kfx_volume* colorvol = nullptr,
                          kfx_sdf_summary* summary = nullptr)
{
    RayParams p;
    if (int e = ray_params<CELL>(p, depth, norm, img, vol, T_wc, K, near, far, trunc_dist, subpix)) return e;
    if (p.w == 0 || p.h == 0) return 0;

    static const int tile_env = [] { const char* e = getenv("KFX_RAYCAST_TILE"); const int v = e ? atoi(e) : 5; return v < 0 ? 0 : (v > 6 ? 6 : v); }();
    static const int wg_env = [] { const char* e = getenv("KFX_RAYCAST_WG"); const int v = e ? atoi(e) : 1; return v < 0 ? 0 : (v > 2 ? 2 : v); }();
    p.tile_log2w = tile_env;
    p.wg_log2x = wg_env;
    dim3 grid(ceil_div(p.w, (1 << p.wg_log2x) << p.tile_log2w), ceil_div(p.h, (4 >> p.wg_log2x) * (64 >> p.tile_log2w)));
    static const int lanes_env = [] { const char* e = getenv("KFX_RAYCAST_LANES"); const int v = e ? atoi(e) : 0; return (v == 8 || v == 16 || v == 32) ? v : (v < 0 ? -1 : 0); }();
    // small images (pyramid levels): rays of a wave are many voxels apart, 16 rays per wave wait for fewer misses per step
    const int lanes = lanes_env ? lanes_env : ((long long)p.w * p.h <= 160 * 120 ? 16 : 0);
    if (lanes && lanes_env >= 0) {
        p.sparse_lanes = lanes;
        grid = dim3(ceil_div(p.w, 2 * lanes), ceil_div(p.h, 2));
    }
    ColorGeom cv{};
    if (colorvol) {
        if (!colorvol->ptr) return set_error(KFX_E_NULL, "RaycastSdf(colour): null colour volume");
        if (colorvol->w < 2 || colorvol->h < 2 || colorvol->d < 2 || colorvol->pitch < colorvol->w * 4 ||
            colorvol->img_pitch < colorvol->pitch * (colorvol->h - 1) + colorvol->w * 4)
            return set_error(KFX_E_SHAPE, "RaycastSdf(colour): colour volume dimensions / pitch");
        if (((uintptr_t)colorvol->ptr | colorvol->pitch | colorvol->img_pitch) & 3) return set_error(KFX_E_ALIGN, "RaycastSdf(colour): alignment");
        set_geometry(cv, colorvol);
        hipLaunchKernelGGL((k_raycast_sdf<CELL, true>), grid, dim3(256), 0, (hipStream_t)stream, p, cv);
    } else if (summary) {
        if constexpr (CELL::BYTES == 8) {
            ClassView cl;
            size_t cl_bytes = 0;
            int usable = 0;
            if (int e = class_view(cl, &cl_bytes, &usable, summary, vol, p, stream)) return e;
            if (usable) hipLaunchKernelGGL((k_raycast_sdf_classes<CELL>), grid, dim3(256), cl_bytes, (hipStream_t)stream, p, cl);
            else hipLaunchKernelGGL((k_raycast_sdf<CELL, false>), grid, dim3(256), 0, (hipStream_t)stream, p, cv);
            return check_launch("kfx_raycast_sdf_tracked");
        } else {
            return set_error(KFX_E_RANGE, "kfx_raycast_sdf_tracked: fp32 cells only");
        }
    } else {
        hipLaunchKernelGGL((k_raycast_sdf<CELL, false>), grid, dim3(256), 0, (hipStream_t)stream, p, cv);
    }
    return check_launch("kfx_raycast_sdf");
}

extern "C" int kfx_raycast_sdf(const kfx_image* depth, const kfx_image* norm, const kfx_image* img,
                               const kfx_volume* vol, const float T_wc[12], const float K[4], float near,
                               float far, float trunc_dist, int subpix, kfx_stream stream)
{
    return raycast_launch<RayF32>(depth, norm, img, vol, T_wc, K, near, far, trunc_dist, subpix, stream);
}

template <typename CELL>
static int raycast_count_launch(const kfx_volume* vol, unsigned w, unsigned h, const float T_wc[12], const float K[4], float near, float far,
                                float trunc_dist, int subpix, unsigned* d_bitmap, unsigned long long* d_counters, kfx_stream stream)
{
    if (!d_bitmap || !d_counters) return set_error(KFX_E_NULL, "kfx_raycast_sdf_count: null argument");
    // the image arguments of ray_params() only give the launch its size: nothing is written to them
    kfx_image dummy = {(size_t)w * 16, (void*)(uintptr_t)16, w, h};
    RayParams p;
    if (int e = ray_params<CELL>(p, &dummy, &dummy, &dummy, vol, T_wc, K, near, far, trunc_dist, subpix)) return e;
    if (p.w == 0 || p.h == 0) return 0;
    p.dptr = p.nptr = p.iptr = nullptr;
    hipLaunchKernelGGL(k_raycast_sdf_count<CELL>, dim3(ceil_div(p.w, 64), ceil_div(p.h, 4)), dim3(256), 0, (hipStream_t)stream, p, d_bitmap, d_counters);
    return check_launch("kfx_raycast_sdf_count");
}

extern "C" int kfx_raycast_sdf_count(const kfx_volume* vol, unsigned w, unsigned h, const float T_wc[12], const float K[4], float near, float far,
                                     float trunc_dist, int subpix, unsigned* d_bitmap, unsigned long long* d_counters, kfx_stream stream)
{
    return raycast_count_launch<RayF32>(vol, w, h, T_wc, K, near, far, trunc_dist, subpix, d_bitmap, d_counters, stream);
}

extern "C" int kfx_raycast_sdf_count_h(const kfx_volume* vol, unsigned w, unsigned h, const float T_wc[12], const float K[4], float near, float far,
                                       float trunc_dist, int subpix, unsigned* d_bitmap, unsigned long long* d_counters, kfx_stream stream)
{
    return raycast_count_launch<RayF16>(vol, w, h, T_wc, K, near, far, trunc_dist, subpix, d_bitmap, d_counters, stream);
}

extern "C" int kfx_raycast_sdf_count_tracked(const kfx_volume* vol, kfx_sdf_summary* summary, unsigned w, unsigned h, const float T_wc[12], const float K[4],
                                             float near, float far, float trunc_dist, int subpix, unsigned* d_bitmap, unsigned long long* d_counters,
                                             kfx_stream stream)
{
    if (!d_bitmap || !d_counters || !summary) return set_error(KFX_E_NULL, "kfx_raycast_sdf_count_tracked: null argument");
    kfx_image dummy = {(size_t)w * 16, (void*)(uintptr_t)16, w, h};
    RayParams p;
    if (int e = ray_params<RayF32>(p, &dummy, &dummy, &dummy, vol, T_wc, K, near, far, trunc_dist, subpix)) return e;
    if (p.w == 0 || p.h == 0) return 0;
    p.dptr = p.nptr = p.iptr = nullptr;
    ClassView cl;
    size_t cl_bytes = 0;
    int usable = 0;
    // a diagnostics call must not steer the calls that follow: class_view() advances the plain / table choice's call counter
    // (the tables it may build are a pure function of the summary: harmless)
    const unsigned plain_calls = summary->plain_calls;
    const int ce = class_view(cl, &cl_bytes, &usable, summary, vol, p, stream);
    summary->plain_calls = plain_calls;
    if (ce) return ce;
    const dim3 grid(ceil_div(p.w, 64), ceil_div(p.h, 4));
    if (usable) hipLaunchKernelGGL(k_raycast_sdf_classes_count<RayF32>, grid, dim3(256), cl_bytes, (hipStream_t)stream, p, cl, d_bitmap, d_counters);
    else hipLaunchKernelGGL(k_raycast_sdf_count<RayF32>, grid, dim3(256), 0, (hipStream_t)stream, p, d_bitmap, d_counters);   // what the tracked call would launch
    return check_launch("kfx_raycast_sdf_count_tracked");
}

extern "C" int kfx_raycast_sdf_tracked(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, const kfx_volume* vol,
                                       kfx_sdf_summary* summary, const float T_wc[12], const float K[4], float near, float far,
                                       float trunc_dist, int subpix, kfx_stream stream)
{
    if (!summary) return set_error(KFX_E_NULL, "kfx_raycast_sdf_tracked: null summary");
    return raycast_launch<RayF32>(depth, norm, img, vol, T_wc, K, near, far, trunc_dist, subpix, stream, nullptr, summary);
}

extern "C" int kfx_raycast_sdf_h(const kfx_image* depth, const kfx_image* norm, const kfx_image* img,
                                 const kfx_volume* vol, const float T_wc[12], const float K[4], float near,
                                 float far, float trunc_dist, int subpix, kfx_stream stream)
{
    return raycast_launch<RayF16>(depth, norm, img, vol, T_wc, K, near, far, trunc_dist, subpix, stream);
}

// RaycastSdf(depth, norm, img, vol, colorVol, T_wc, K, near, far, trunc_dist, subpix) (cu_raycast.cu:119-196)
extern "C" int kfx_raycast_sdf_levels(int n_levels, const kfx_image* const* depth, const kfx_image* const* norm, const kfx_image* const* img,
                                      const kfx_image* const* vbo, const kfx_volume* vol, const float T_wc[12], const float* K, float near, float far,
                                      float trunc_dist, int subpix, kfx_stream stream)
{
    return raycast_levels_launch<RayF32>(n_levels, depth, norm, img, vbo, vol, T_wc, K, near, far, trunc_dist, subpix, stream);
}

extern "C" int kfx_raycast_sdf_levels_tracked(int n_levels, const kfx_image* const* depth, const kfx_image* const* norm, const kfx_image* const* img,
                                              const kfx_image* const* vbo, const kfx_volume* vol, kfx_sdf_summary* summary, const float T_wc[12],
                                              const float* K, float near, float far, float trunc_dist, int subpix, kfx_stream stream)
{
    if (!summary) return set_error(KFX_E_NULL, "kfx_raycast_sdf_levels_tracked: null summary");
    return raycast_levels_launch<RayF32>(n_levels, depth, norm, img, vbo, vol, T_wc, K, near, far, trunc_dist, subpix, stream, summary);
}

extern "C" int kfx_raycast_sdf_levels_h(int n_levels, const kfx_image* const* depth, const kfx_image* const* norm, const kfx_image* const* img,
                                      const kfx_image* const* vbo, const kfx_volume* vol, const float T_wc[12], const float* K, float near, float far,
                                      float trunc_dist, int subpix, kfx_stream stream)
{
    return raycast_levels_launch<RayF16>(n_levels, depth, norm, img, vbo, vol, T_wc, K, near, far, trunc_dist, subpix, stream);
}

extern "C" int kfx_raycast_sdf_color(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, const kfx_volume* vol,
                                     const kfx_volume* colorvol, const float T_wc[12], const float K[4], float near, float far,
                                     float trunc_dist, int subpix, kfx_stream stream)
{
    if (!colorvol) return set_error(KFX_E_NULL, "RaycastSdf(colour): null colour volume");
    return raycast_launch<RayF32>(depth, norm, img, vol, T_wc, K, near, far, trunc_dist, subpix, stream, colorvol);
}

// Exact multi-GPU march: one round of a rank (see k_raycast_sdf_slab).  `vol` holds planes
// [slab->z_offset, slab->z_offset + vol->d) of the full volume described by `slab`; the rank owns the
// trilinear base cells [own_lo, own_hi).  `state` is KFX_RAY_STATE_PLANES dense planes of h*w floats; init != 0 starts the rays.
template <typename CELL>
static int raycast_slab_launch(const SlabRay& geom, const kfx_volume* vol, const kfx_slab* slab, int own_lo, int own_hi,
                                    int w, int h, const float T_wc[12], const float K[4], float near, float far,
                                    float trunc_dist, int subpix, kfx_stream stream)
{
    if (!geom.state || !geom.result || !vol || !vol->ptr || !slab || !T_wc || !K) return set_error(KFX_E_NULL, "RaycastSdf(slab): null argument");
    if (w <= 0 || h <= 0 || geom.v1 <= geom.v0) return 0;
    if (geom.R < 1 || geom.v0 < 0 || geom.v1 > h || geom.P < (size_t)geom.R * (size_t)w) return set_error(KFX_E_SHAPE, "RaycastSdf(slab): tile geometry");
    if (slab->full_d < 3 || slab->z_offset + vol->d > slab->full_d || vol->w < 3 || vol->h < 3)
        return set_error(KFX_E_SHAPE, "RaycastSdf(slab): slab outside the full volume");
    if ((((uintptr_t)vol->ptr | vol->pitch | vol->img_pitch) & (CELL::BYTES - 1)) || (((uintptr_t)geom.state | (uintptr_t)geom.result | (uintptr_t)geom.fin) & 3))
        return set_error(KFX_E_ALIGN, "RaycastSdf(slab): alignment");
    RayParams p;
    // full-volume geometry, virtual base pointer (never dereferenced outside [avail_lo, avail_hi))
    p.vol.ptr = (unsigned char*)vol->ptr - (ptrdiff_t)slab->z_offset * (ptrdiff_t)vol->img_pitch;
    p.vol.pitch = vol->pitch;
    p.vol.img_pitch = vol->img_pitch;
    p.vol.w = (int)vol->w;
    p.vol.h = (int)vol->h;
    p.vol.d = (int)slab->full_d;
    p.vol.bmin = V3{vol->boxmin[0], vol->boxmin[1], slab->full_zmin};
    p.vol.bmax = V3{vol->boxmax[0], vol->boxmax[1], slab->full_zmax};
    p.size = V3{vol->boxmax[0] - vol->boxmin[0], vol->boxmax[1] - vol->boxmin[1], slab->full_zmax - slab->full_zmin};
    p.dims1 = V3{(float)vol->w - 1.f, (float)vol->h - 1.f, (float)slab->full_d - 1.f};
    p.hi2 = V3{(float)(vol->w - 2), (float)(vol->h - 2), (float)(slab->full_d - 2)};
    p.voxel = V3{p.size.x / (float)(vol->w - 1), p.size.y / (float)(vol->h - 1), p.size.z / (float)(slab->full_d - 1)};
    set_shortcuts(p);
    for (int i = 0; i < 12; ++i) p.T.m[i] = T_wc[i];
    p.K = Intr{K[0], K[1], K[2], K[3]};
    p.dptr = p.nptr = p.iptr = nullptr;
    p.dpitch = p.npitch = p.ipitch = 0;
    p.w = w;
    p.h = h;
    p.near = near;
    p.far = far;
    p.trunc = trunc_dist;
    p.subpix = subpix ? 1 : 0;
    p.tile_log2w = 3;
    p.wg_log2x = 1;
    SlabRay sl = geom;
    sl.own_lo = own_lo; sl.own_hi = own_hi;
    sl.avail_lo = (int)slab->z_offset; sl.avail_hi = (int)(slab->z_offset + vol->d);
    dim3 grid(ceil_div(w, 64), ceil_div(sl.v1 - sl.v0, 4));
    hipLaunchKernelGGL(k_raycast_sdf_slab<CELL>, grid, dim3(256), 0, (hipStream_t)stream, p, sl);
    return check_launch("kfx_raycast_sdf_slab");
}

// the dense [9][h w] state of kfx_raycast_sdf_slab as one tile
static SlabRay dense_state(float* state, int init, int w, int h)
{
    SlabRay g{};
    const size_t n = (size_t)(w > 0 ? w : 0) * (size_t)(h > 0 ? h : 0);
    g.state = state; g.touched = state ? state + 4 * n : nullptr; g.result = state ? state + 5 * n : nullptr; g.P = n; g.R = h > 0 ? h : 1; g.v0 = 0; g.v1 = h; g.fin = nullptr; g.claim_misses = 0;
    g.init = init ? 1 : 0;
    g.adopt_lo = g.adopt_hi = nullptr; g.adopt_tile_major = 0;
    return g;
}

extern "C" int kfx_raycast_sdf_slab_tiles(float* state, float* result, size_t plane_stride, int rows_per_tile, int v0, int v1, int init, int* fin,
                                          int claim_misses, const float* adopt_lo, const float* adopt_hi, int layout_flags, const kfx_volume* vol,
                                          const kfx_slab* slab, int own_lo, int own_hi, int w, int h, const float T_wc[12], const float K[4], float near,
                                          float far, float trunc_dist, int subpix, kfx_stream stream)
{
    SlabRay g{};
    g.state = state; g.result = result; g.P = plane_stride; g.R = rows_per_tile; g.v0 = v0; g.v1 = v1; g.fin = fin; g.claim_misses = claim_misses ? 1 : 0;
    g.init = init ? 1 : 0;
    const int adopt_tile_major = layout_flags & 1;
    g.adopt_lo = adopt_lo; g.adopt_hi = adopt_hi; g.adopt_tile_major = adopt_tile_major;
    g.packed = (layout_flags & 2) ? 1 : 0;
    g.normals_here = (layout_flags & 4) ? 1 : 0;
    if (g.packed && !(trunc_dist > 0.f)) return set_error(KFX_E_RANGE, "RaycastSdf(slab): the packed tile state needs a positive truncation distance");
    if (((uintptr_t)adopt_lo | (uintptr_t)adopt_hi) & 3) return set_error(KFX_E_ALIGN, "RaycastSdf(slab): alignment of the received snapshots");
    if (!adopt_tile_major && (adopt_lo || adopt_hi) && rows_per_tile > 0 && (v0 / rows_per_tile != (v1 - 1) / rows_per_tile))
        return set_error(KFX_E_SHAPE, "RaycastSdf(slab): one-tile snapshots with rows of several tiles");
    return raycast_slab_launch<RayF32>(g, vol, slab, own_lo, own_hi, w, h, T_wc, K, near, far, trunc_dist, subpix, stream);
}

extern "C" int kfx_raycast_sdf_slab(float* state, int init, const kfx_volume* vol, const kfx_slab* slab, int own_lo, int own_hi,
                                    int w, int h, const float T_wc[12], const float K[4], float near, float far,
                                    float trunc_dist, int subpix, kfx_stream stream)
{
    return raycast_slab_launch<RayF32>(dense_state(state, init, w, h), vol, slab, own_lo, own_hi, w, h, T_wc, K, near, far, trunc_dist, subpix, stream);
}

extern "C" int kfx_raycast_sdf_slab_h(float* state, int init, const kfx_volume* vol, const kfx_slab* slab, int own_lo, int own_hi,
                                      int w, int h, const float T_wc[12], const float K[4], float near, float far,
                                      float trunc_dist, int subpix, kfx_stream stream)
{
    return raycast_slab_launch<RayF16>(dense_state(state, init, w, h), vol, slab, own_lo, own_hi, w, h, T_wc, K, near, far, trunc_dist, subpix, stream);
}

extern "C" int kfx_raycast_state_to_images(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, const float* state,
                                           kfx_stream stream)
{
    if (!depth || !norm || !img || !state || !depth->ptr || !norm->ptr || !img->ptr) return set_error(KFX_E_NULL, "raycast state: null argument");
    if (img->w == 0 || img->h == 0) return 0;
    if (depth->w < img->w || depth->h < img->h || norm->w < img->w || norm->h < img->h) return set_error(KFX_E_SHAPE, "raycast state: image sizes");
    RayParams p{};
    p.dptr = (unsigned char*)depth->ptr; p.nptr = (unsigned char*)norm->ptr; p.iptr = (unsigned char*)img->ptr;
    p.dpitch = depth->pitch; p.npitch = norm->pitch; p.ipitch = img->pitch;
    p.w = (int)img->w; p.h = (int)img->h;
    hipLaunchKernelGGL(k_raycast_state_to_images, dim3(ceil_div(p.w, 64), ceil_div(p.h, 4)), dim3(256), 0, (hipStream_t)stream, p, state);
    return check_launch("kfx_raycast_state_to_images");
}
