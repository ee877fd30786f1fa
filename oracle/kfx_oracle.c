/*
 * kfx_oracle.c -- CPU restatement (parity oracle) of the KinectFusion volumetric
 * hot path of arpg/Kangaroo.  TEST INFRASTRUCTURE ONLY -- see kfx_oracle.h.
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 * Every arithmetic expression keeps the reference's operand order and
 * association; all arithmetic is IEEE binary32 with no fused multiply-add, so a
 * HIP kernel built with -ffp-contract=off and correctly rounded div/sqrt can be
 * compared bit-for-bit.  Citations are paths inside the reference tree.
 */
#include "kfx_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;
typedef struct { float val, w; } sdf_t; /* Sdf.h:11-36 */

/* ---- scalar helpers: CUDA_SDK/cutil_math.h ---------------------------------- */
static inline float lerpf(float a, float b, float t) { return a + t * (b - a); }       /* :80-83 */
static inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); } /* :86-89 */
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }      /* :392-395 */
static inline float length3(f3 v) { return sqrtf(dot3(v, v)); }                         /* :404-407 */
static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }       /* :341-352 */
static inline f3 div33(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }     /* :354-357 */
/* float3 / float is multiply-by-reciprocal: cutil_math.h:358-362 (quirk Q5) */
static inline f3 div3s(f3 a, float s) { float inv = 1.0f / s; return scale3(a, inv); }
static inline f3 lerp3(f3 a, f3 b, float t) { return add3(a, scale3(sub3(b, a), t)); }  /* :375-378 */
static inline int clampi(int f, int a, int b) { return f < a ? a : (f > b ? b : f); }   /* :653-656 */

/* ---- Mat<float,3,4> helpers: MatUtils.h ------------------------------------- */
#define T_(r, c) T[(r) * 4 + (c)] /* row-major, Mat.h:36-42 */
static inline f3 se3_mul(const float* T, f3 p) /* MatUtils.h:117-125 */
{
    return mk3(T_(0, 0) * p.x + T_(0, 1) * p.y + T_(0, 2) * p.z + T_(0, 3),
               T_(1, 0) * p.x + T_(1, 1) * p.y + T_(1, 2) * p.z + T_(1, 3),
               T_(2, 0) * p.x + T_(2, 1) * p.y + T_(2, 2) * p.z + T_(2, 3));
}
static inline f3 so3_mul(const float* T, f3 r) /* MatUtils.h:147-155 */
{
    return mk3(T_(0, 0) * r.x + T_(0, 1) * r.y + T_(0, 2) * r.z,
               T_(1, 0) * r.x + T_(1, 1) * r.y + T_(1, 2) * r.z,
               T_(2, 0) * r.x + T_(2, 1) * r.y + T_(2, 2) * r.z);
}
static inline f3 so3_mul_inv(const float* T, f3 r) /* MatUtils.h:177-185 */
{
    return mk3(T_(0, 0) * r.x + T_(1, 0) * r.y + T_(2, 0) * r.z,
               T_(0, 1) * r.x + T_(1, 1) * r.y + T_(2, 1) * r.z,
               T_(0, 2) * r.x + T_(1, 2) * r.y + T_(2, 2) * r.z);
}
static inline f3 se3_mul_inv(const float* T, f3 r) /* MatUtils.h:192-200 */
{
    const float ax = r.x - T_(0, 3), ay = r.y - T_(1, 3), az = r.z - T_(2, 3);
    return mk3(T_(0, 0) * ax + T_(1, 0) * ay + T_(2, 0) * az,
               T_(0, 1) * ax + T_(1, 1) * ay + T_(2, 1) * az,
               T_(0, 2) * ax + T_(1, 2) * ay + T_(2, 2) * az);
}
static inline f3 se3_translation(const float* T) { return mk3(T_(0, 3), T_(1, 3), T_(2, 3)); } /* :216-220 */

void kfo_se3_inverse(float o[12], const float T[12]) /* MatUtils.h:202-214 */
{
    o[0] = T_(0, 0); o[1] = T_(1, 0); o[2]  = T_(2, 0);
    o[4] = T_(0, 1); o[5] = T_(1, 1); o[6]  = T_(2, 1);
    o[8] = T_(0, 2); o[9] = T_(1, 2); o[10] = T_(2, 2);
    o[3]  = -(o[0] * T_(0, 3) + o[1] * T_(1, 3) + o[2]  * T_(2, 3));
    o[7]  = -(o[4] * T_(0, 3) + o[5] * T_(1, 3) + o[6]  * T_(2, 3));
    o[11] = -(o[8] * T_(0, 3) + o[9] * T_(1, 3) + o[10] * T_(2, 3));
}

/* ---- ImageIntrinsics {fu,fv,u0,v0}: ImageIntrinsics.h ------------------------ */
#define FU K[0]
#define FV K[1]
#define U0 K[2]
#define V0 K[3]
static inline f3 unproject1(const float* K, float u, float v) /* :109-113 */
{
    return mk3((u - U0) / FU, (v - V0) / FV, 1.0f);
}

/* ---- pitched containers ------------------------------------------------------ */
static inline unsigned char* img_row(const kfo_image* im, size_t y) /* Image.h:235-245 */
{
    return (unsigned char*)im->ptr + y * im->pitch;
}
static inline sdf_t* vol_row(const kfo_volume* v, size_t y, size_t z) /* Volume.h:125-135 */
{
    return (sdf_t*)((unsigned char*)v->ptr + z * v->img_pitch + y * v->pitch);
}
/* fp16 cells (BASELINE config C5): {half val; half w;}, 4 bytes, the arithmetic of the reference's
 * commented-out half variant (Sdf.h:38-62): every intermediate is rounded to half, round-to-nearest-even.
 * F16C conversions (vcvtps2ph / vcvtph2ps) are exact IEEE binary16 <-> binary32. */
typedef struct { uint16_t val, w; } sdfh_t;
static inline uint16_t f2h(float x) { return (uint16_t)_cvtss_sh(x, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }
static inline float h2f(uint16_t h) { return _cvtsh_ss(h); }
static inline float qh(float x) { return h2f(f2h(x)); }
static inline sdfh_t* volh_row(const kfo_volume* v, size_t y, size_t z)
{
    return (sdfh_t*)((unsigned char*)v->ptr + z * v->img_pitch + y * v->pitch);
}
/* `half` selects the cell type of the volume behind the (layout-identical) kfo_volume view */
static inline float vol_val_c(const kfo_volume* v, int x, int y, int z, int half)
{
    return half ? h2f(volh_row(v, (size_t)y, (size_t)z)[x].val) : vol_row(v, (size_t)y, (size_t)z)[x].val;
}
#define vol_val(v, x, y, z) vol_val_c(v, x, y, z, half) /* Volume.h:161-171 + Sdf.h:16-18 */
static inline f3 box_size(const kfo_volume* v) /* BoundingBox.h:139-143 */
{
    return mk3(v->boxmax[0] - v->boxmin[0], v->boxmax[1] - v->boxmin[1], v->boxmax[2] - v->boxmin[2]);
}
static inline f3 box_min(const kfo_volume* v) { return mk3(v->boxmin[0], v->boxmin[1], v->boxmin[2]); }
static inline f3 box_max(const kfo_volume* v) { return mk3(v->boxmax[0], v->boxmax[1], v->boxmax[2]); }
static inline f3 voxel_size_units(const kfo_volume* v) /* BoundedVolume.h:67-76 */
{
    return div33(box_size(v), mk3((float)(v->w - 1), (float)(v->h - 1), (float)(v->d - 1)));
}
static inline f3 voxel_position(const kfo_volume* v, int x, int y, int z) /* BoundedVolume.h:115-125 */
{
    const f3 s = box_size(v);
    return mk3(v->boxmin[0] + s.x * (float)x / (float)(v->w - 1),
               v->boxmin[1] + s.y * (float)y / (float)(v->h - 1),
               v->boxmin[2] + s.z * (float)z / (float)(v->d - 1));
}

static int pick_threads(int n)
{
#ifdef _OPENMP
    if (n <= 0) n = omp_get_max_threads();
    return n;
#else
    (void)n;
    return 1;
#endif
}
int kfo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ============================================================================
 * BilateralFilter -- src/cu_bilateral.cu
 * ========================================================================== */
#define BILATERAL_BODY(TI, READ_MINVAL_TEST_P, READ_MINVAL_TEST_Q)                                \
    const int W = (int)in->w, H = (int)in->h;                                                     \
    const int nt = pick_threads(nthreads);                                                        \
    (void)nt;                                                                                     \
    _Pragma("omp parallel for num_threads(nt) schedule(static)")                                  \
    for (int y = 0; y < (int)out->h; ++y) {                                                       \
        for (int x = 0; x < (int)out->w; ++x) {                                                   \
            const TI p = ((const TI*)img_row(in, (size_t)y))[x]; /* :21 / :67 */                  \
            float sum = 0, sumw = 0;                                                              \
            if (READ_MINVAL_TEST_P) {                                                             \
                for (int r = -size; r <= size; ++r) {                                             \
                    for (int c = -size; c <= size; ++c) {                                         \
                        /* GetWithClampedRange, Image.h:297-303 */                                \
                        const int qx = clampi(x + c, 0, W - 1), qy = clampi(y + r, 0, H - 1);     \
                        const TI q = ((const TI*)img_row(in, (size_t)qy))[qx];                    \
                        if (READ_MINVAL_TEST_Q) {                                                 \
                            const float sd2 = (float)(r * r + c * c);                             \
                            const float id = (float)(p - q);                                      \
                            const float id2 = id * id;                                            \
                            const float sw = expf(-(sd2) / (2 * gs * gs)); /* __expf, :31-32 */   \
                            const float iw = expf(-(id2) / (2 * gr * gr));                        \
                            const float w = sw * iw;                                              \
                            sumw += w;                                                            \
                            sum += w * (float)q;                                                  \
                        }                                                                         \
                    }                                                                             \
                }                                                                                 \
            }                                                                                     \
            ((float*)img_row(out, (size_t)y))[x] = sum / sumw; /* 0/0 = NaN, :89-90 */            \
        }                                                                                         \
    }

void kfo_bilateral_f32(const kfo_image* out, const kfo_image* in, float gs, float gr, int size,
                       float minval, int use_minval, int nthreads)
{
    if (use_minval) { /* cu_bilateral.cu:59-92 */
        BILATERAL_BODY(float, p >= minval, q >= minval)
    } else { /* cu_bilateral.cu:13-41 */
        BILATERAL_BODY(float, 1, 1)
    }
}
void kfo_bilateral_u16(const kfo_image* out, const kfo_image* in, float gs, float gr, int size,
                       unsigned short minval, int nthreads)
{
    BILATERAL_BODY(unsigned short, p >= minval, q >= minval) /* cu_bilateral.cu:59-92,104 */
}
void kfo_bilateral_u8(const kfo_image* out, const kfo_image* in, float gs, float gr, int size,
                      int nthreads)
{
    BILATERAL_BODY(unsigned char, 1, 1) /* cu_bilateral.cu:13-41,53 */
}

/* ============================================================================
 * DepthToVbo -- src/cu_depth_tools.cu:59-78 ; Unproject ImageIntrinsics.h:127-131
 * ========================================================================== */
void kfo_depth_to_vbo_f32(const kfo_image* vbo, const kfo_image* depth, const float K[4], float scale)
{
    for (size_t v = 0; v < vbo->h; ++v) {
        const float* drow = (const float*)img_row(depth, v);
        f4* orow = (f4*)img_row(vbo, v);
        for (size_t u = 0; u < vbo->w; ++u) {
            const float kz = scale * drow[u];
            f4 P = {kz * ((float)(int)u - U0) / FU, kz * ((float)(int)v - V0) / FV, kz, 1.0f};
            orow[u] = P;
        }
    }
}
void kfo_depth_to_vbo_u16(const kfo_image* vbo, const kfo_image* depth, const float K[4], float scale)
{
    for (size_t v = 0; v < vbo->h; ++v) {
        const unsigned short* drow = (const unsigned short*)img_row(depth, v);
        f4* orow = (f4*)img_row(vbo, v);
        for (size_t u = 0; u < vbo->w; ++u) {
            const float kz = scale * (float)drow[u];
            f4 P = {kz * ((float)(int)u - U0) / FU, kz * ((float)(int)v - V0) / FV, kz, 1.0f};
            orow[u] = P;
        }
    }
}

/* ============================================================================
 * NormalsFromVbo -- src/cu_normals.cu:12-45
 * ========================================================================== */
void kfo_normals_from_vbo(const kfo_image* nrm, const kfo_image* vbo)
{
    const int W = (int)nrm->w, H = (int)nrm->h;
    for (int v = 0; v < H; ++v) {
        f4* orow = (f4*)img_row(nrm, (size_t)v);
        for (int u = 0; u < W; ++u) {
            if (u + 1 < W && v + 1 < H) {
                const f4 Vc = ((const f4*)img_row(vbo, (size_t)v))[u];
                const f4 Vr = ((const f4*)img_row(vbo, (size_t)v))[u + 1];
                const f4 Vu = ((const f4*)img_row(vbo, (size_t)v + 1))[u];
                const f3 a = {Vr.x - Vc.x, Vr.y - Vc.y, Vr.z - Vc.z};
                const f3 b = {Vu.x - Vc.x, Vu.y - Vc.y, Vu.z - Vc.z};
                const f3 axb = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
                const float mag = length3(axb);
                f4 N = {-axb.x / mag, -axb.y / mag, -axb.z / mag, 1.0f};
                orow[u] = N;
            } else {
                f4 Z = {0, 0, 0, 0};
                orow[u] = Z;
            }
        }
    }
}

/* ============================================================================
 * SdfReset / SdfSphere -- src/cu_sdffusion.cu:153-195
 * ========================================================================== */
void kfo_sdf_reset(const kfo_volume* vol, float trunc)
{
    /* thrust::fill(begin,end) over ptr .. RowPtr(h-1,d-1)+w: contiguous, includes
     * the pitch padding (Volume.h:343-356). */
    sdf_t* b = (sdf_t*)vol->ptr;
    sdf_t* e = vol_row(vol, vol->h - 1, vol->d - 1) + vol->w;
    const sdf_t v = {trunc, 0.0f};
    for (; b < e; ++b) *b = v;
}

void kfo_sdf_sphere(const kfo_volume* vol, const float center[3], float r)
{
    /* grid is w/8,h/8,d/8 blocks of 8 (cu_sdffusion.cu:189-191) */
    const int X = (int)(vol->w / 8) * 8, Y = (int)(vol->h / 8) * 8, Z = (int)(vol->d / 8) * 8;
    const f3 c = {center[0], center[1], center[2]};
    for (int z = 0; z < Z; ++z)
        for (int y = 0; y < Y; ++y) {
            sdf_t* row = vol_row(vol, (size_t)y, (size_t)z);
            for (int x = 0; x < X; ++x) {
                const f3 pos = voxel_position(vol, x, y, z);
                const float dist = length3(sub3(pos, c));
                sdf_t s = {dist - r, 1.0f}; /* SDF_t(float v): w = 1, Sdf.h:13 */
                row[x] = s;
            }
        }
}

/* ============================================================================
 * SdfFuse -- src/cu_sdffusion.cu:16-61
 * ========================================================================== */
static inline int fuse_voxel(const kfo_volume* vol, const kfo_image* depth, const kfo_image* normals,
                             const float* T, const float* K, float trunc_dist, float max_w,
                             float mincostheta, int x, int y, int z, int half, const kfo_slab* slab)
{
    f3 P_w = voxel_position(vol, x, y, z);               /* :22 */
    if (slab) /* Z-slab of a larger volume: z position by the FULL volume's expression (BoundedVolume.h:115-125) */
        P_w.z = slab->full_zmin + (slab->full_zmax - slab->full_zmin) * (float)(z + (int)slab->z_offset) / (float)(slab->full_d - 1);
    const f3 P_c = se3_mul(T, P_w);                       /* :23 */
    /* K.Project, ImageIntrinsics.h:87-91 */
    const float pu = U0 + FU * P_c.x / P_c.z;
    const float pv = V0 + FV * P_c.y / P_c.z;
    /* depth.InBounds(p_c, 2), Image.h:287-291 */
    const float border = 2.0f;
    if (!(border <= pu && pu < ((float)depth->w - border) && border <= pv &&
          pv < ((float)depth->h - border)))
        return 0;

    const float vd = P_c.z;
    /* GetBilinear, Image.h:317-334 (quirk Q6: floorf -> float -> size_t) */
    const float ix = floorf(pu), iy = floorf(pv);
    const float fx = pu - ix, fy = pv - iy;
    const float* dbl = (const float*)img_row(depth, (size_t)iy) + (size_t)ix;
    const float* dtl = (const float*)img_row(depth, (size_t)(iy + 1)) + (size_t)ix;
    const float md = lerpf(lerpf(dbl[0], dbl[1], fx), lerpf(dtl[0], dtl[1], fx), fy);
    const f4* nbl = (const f4*)img_row(normals, (size_t)iy) + (size_t)ix;
    const f4* ntl = (const f4*)img_row(normals, (size_t)(iy + 1)) + (size_t)ix;
    f3 mdn;
    mdn.x = lerpf(lerpf(nbl[0].x, nbl[1].x, fx), lerpf(ntl[0].x, ntl[1].x, fx), fy);
    mdn.y = lerpf(lerpf(nbl[0].y, nbl[1].y, fx), lerpf(ntl[0].y, ntl[1].y, fx), fy);
    mdn.z = lerpf(lerpf(nbl[0].z, nbl[1].z, fx), lerpf(ntl[0].z, ntl[1].z, fx), fy);

    const float costheta = dot3(mdn, P_c) / -length3(P_c); /* :35 */
    const float sd = costheta * (md - vd);                 /* :36 */
    const float w = costheta * 1.0f / vd;                  /* :37 */

    if (sd <= -trunc_dist) return 0;                       /* :39-42 */
    if (half && isfinite(md) && isfinite(w) && costheta > mincostheta) {
        /* same update on a half cell: SDF_t(v, w) rounds both to half, then Sdf.h:52-58 */
        sdfh_t* cell = &volh_row(vol, (size_t)y, (size_t)z)[x];
        float sval = qh(clampf(sd, -trunc_dist, trunc_dist)), sw = qh(w);
        const float rval = h2f(cell->val), rw = h2f(cell->w);
        if (rw > 0) {
            sval = qh(sw * sval + rw * rval);
            sw = qh(sw + rw);
            sval = qh(sval / sw);
        }
        sw = qh(fminf(sw, max_w));
        cell->val = f2h(sval);
        cell->w = f2h(sw);
        return 1;
    }
    if (isfinite(md) && isfinite(w) && costheta > mincostheta) { /* :44 */
        sdf_t* cell = &vol_row(vol, (size_t)y, (size_t)z)[x];
        sdf_t s = {clampf(sd, -trunc_dist, trunc_dist), w}; /* :45 */
        const sdf_t rhs = *cell;                            /* :46, Sdf.h:25-32 */
        if (rhs.w > 0) {
            s.val = (s.w * s.val + rhs.w * rhs.val);
            s.w += rhs.w;
            s.val /= s.w;
        }
        s.w = fminf(s.w, max_w); /* LimitWeight, Sdf.h:22-24 */
        *cell = s;               /* :49 */
        return 1;
    }
    return 0;
}

static uint64_t sdf_fuse_any(const kfo_volume* vol, const kfo_image* depth, const kfo_image* norm,
                             const float T_cw[12], const float K[4], float trunc, float max_w,
                             float mincostheta, int full_extent, int nthreads, int half, const kfo_slab* slab)
{
    /* gridDim = (w/8, h/8, d/8), blockDim = (8,8,8): integer division, no tail (quirk Q1) */
    /* full_extent: 0 = the reference's (dim/8)*8 extents of `vol` (quirk Q1), 1 = every voxel, 2 (slabs) = the reference's
     * extents on the WHOLE volume: x / y (dim/8)*8, z = the local planes below (full_d/8)*8 */
    const int full = full_extent == 1;
    const int X = full ? (int)vol->w : (int)(vol->w / 8) * 8;
    const int Y = full ? (int)vol->h : (int)(vol->h / 8) * 8;
    int Z = full ? (int)vol->d : (int)(vol->d / 8) * 8;
    if (full_extent == 2 && slab) {
        const size_t zlim = (slab->full_d / 8) * 8;
        const size_t z_end = slab->z_offset + vol->d < zlim ? slab->z_offset + vol->d : zlim;
        Z = z_end > slab->z_offset ? (int)(z_end - slab->z_offset) : 0;
    }
    uint64_t updated = 0;
    const int nt = pick_threads(nthreads);
    (void)nt;
#pragma omp parallel for num_threads(nt) schedule(static) reduction(+ : updated)
    for (int z = 0; z < Z; ++z)
        for (int y = 0; y < Y; ++y)
            for (int x = 0; x < X; ++x)
                updated += (uint64_t)fuse_voxel(vol, depth, norm, T_cw, K, trunc, max_w, mincostheta, x, y, z, half, slab);
    return updated;
}

uint64_t kfo_sdf_fuse(const kfo_volume* vol, const kfo_image* depth, const kfo_image* norm,
                      const float T_cw[12], const float K[4], float trunc, float max_w,
                      float mincostheta, int full_extent, int nthreads)
{
    return sdf_fuse_any(vol, depth, norm, T_cw, K, trunc, max_w, mincostheta, full_extent, nthreads, 0, NULL);
}
uint64_t kfo_sdf_fuse_slab(const kfo_volume* vol, const kfo_slab* slab, const kfo_image* depth, const kfo_image* norm,
                           const float T_cw[12], const float K[4], float trunc, float max_w,
                           float mincostheta, int full_extent, int nthreads)
{
    return sdf_fuse_any(vol, depth, norm, T_cw, K, trunc, max_w, mincostheta, full_extent, nthreads, 0, slab);
}
uint64_t kfo_sdf_fuse_h(const kfo_volume* vol, const kfo_image* depth, const kfo_image* norm,
                        const float T_cw[12], const float K[4], float trunc, float max_w,
                        float mincostheta, int full_extent, int nthreads)
{
    return sdf_fuse_any(vol, depth, norm, T_cw, K, trunc, max_w, mincostheta, full_extent, nthreads, 1, NULL);
}

/* ============================================================================
 * RaycastSdf -- src/cu_raycast.cu:14-113
 * ========================================================================== */
static inline float phong_shade(f3 p_c, f3 n_c) /* cu_raycast.cu:14-28 */
{
    const float ambient = (float)0.4, diffuse = (float)0.4, specular = (float)0.2;
    const f3 eyedir = div3s(scale3(p_c, -1.0f), length3(p_c));
    const f3 _lightdir = {(float)0.4, (float)0.4, -1.0f};
    const f3 lightdir = div3s(_lightdir, length3(_lightdir));
    const float ldotn = dot3(lightdir, n_c);
    const f3 lightreflect = add3(scale3(n_c, 2 * ldotn), scale3(lightdir, -1.0f));
    const float edotr = fmaxf(0.0f, dot3(eyedir, lightreflect));
    const float spec = edotr * edotr * edotr * edotr * edotr * edotr * edotr * edotr * edotr * edotr;
    return ambient + diffuse * ldotn + specular * spec;
}

typedef struct {
    uint8_t* bitmap;
    const kfo_volume* vol;
} touch_t;
static inline void touch(touch_t* t, int x, int y, int z)
{
    if (t && t->bitmap) {
        const size_t i = ((size_t)z * t->vol->h + (size_t)y) * t->vol->w + (size_t)x;
        t->bitmap[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
}

/* BoundedVolume::GetUnitsTrilinearClamped (BoundedVolume.h:93-98) ->
 * Volume::GetFractionalTrilinearClamped (Volume.h:224-250); quirk Q4: only the
 * integer cell is clamped, the fraction extrapolates. */
static inline float trilinear_clamped(const kfo_volume* v, f3 pos_w, touch_t* t, int half)
{
    const f3 pos_v = div33(sub3(pos_w, box_min(v)), box_size(v));
    const f3 pf = {pos_v.x * ((float)v->w - 1.f), pos_v.y * ((float)v->h - 1.f), pos_v.z * ((float)v->d - 1.f)};
    const int ix = (int)fmaxf(fminf((float)(v->w - 2), floorf(pf.x)), 0);
    const int iy = (int)fmaxf(fminf((float)(v->h - 2), floorf(pf.y)), 0);
    const int iz = (int)fmaxf(fminf((float)(v->d - 2), floorf(pf.z)), 0);
    const float fx = pf.x - (float)ix, fy = pf.y - (float)iy, fz = pf.z - (float)iz;
    const float v0 = vol_val(v, ix, iy, iz), vx = vol_val(v, ix + 1, iy, iz);
    const float vy = vol_val(v, ix, iy + 1, iz), vxy = vol_val(v, ix + 1, iy + 1, iz);
    const float vz = vol_val(v, ix, iy, iz + 1), vxz = vol_val(v, ix + 1, iy, iz + 1);
    const float vyz = vol_val(v, ix, iy + 1, iz + 1), vxyz = vol_val(v, ix + 1, iy + 1, iz + 1);
    if (t && t->bitmap)
        for (int c = 0; c < 8; ++c) touch(t, ix + (c & 1), iy + ((c >> 1) & 1), iz + (c >> 2));
    return lerpf(lerpf(lerpf(v0, vx, fx), lerpf(vy, vxy, fx), fy),
                 lerpf(lerpf(vz, vxz, fx), lerpf(vyz, vxyz, fx), fy), fz);
}

static inline f3 backward_diff(const kfo_volume* v, int x, int y, int z, touch_t* t, int half) /* Volume.h:256-265 */
{
    const float v0 = vol_val(v, x, y, z);
    if (t && t->bitmap) { touch(t, x, y, z); touch(t, x - 1, y, z); touch(t, x, y - 1, z); touch(t, x, y, z - 1); }
    return mk3(v0 - vol_val(v, x - 1, y, z), v0 - vol_val(v, x, y - 1, z), v0 - vol_val(v, x, y, z - 1));
}

/* BoundedVolume::GetUnitsBackwardDiffDxDyDz (BoundedVolume.h:100-106) ->
 * Volume::GetFractionalBackwardDiffDxDyDz (Volume.h:267-295) */
static inline f3 units_backward_diff(const kfo_volume* v, f3 pos_w, touch_t* t, int half)
{
    const f3 pos_v = div33(sub3(pos_w, box_min(v)), box_size(v));
    const f3 pf = {pos_v.x * ((float)v->w - 1.f), pos_v.y * ((float)v->h - 1.f), pos_v.z * ((float)v->d - 1.f)};
    const int ix = (int)fmaxf(fminf((float)(v->w - 2), floorf(pf.x)), 1);
    const int iy = (int)fmaxf(fminf((float)(v->h - 2), floorf(pf.y)), 1);
    const int iz = (int)fmaxf(fminf((float)(v->d - 2), floorf(pf.z)), 1);
    const float fx = pf.x - (float)ix, fy = pf.y - (float)iy, fz = pf.z - (float)iz;
    const f3 g0 = backward_diff(v, ix, iy, iz, t, half), gx = backward_diff(v, ix + 1, iy, iz, t, half);
    const f3 gy = backward_diff(v, ix, iy + 1, iz, t, half), gxy = backward_diff(v, ix + 1, iy + 1, iz, t, half);
    const f3 gz = backward_diff(v, ix, iy, iz + 1, t, half), gxz = backward_diff(v, ix + 1, iy, iz + 1, t, half);
    const f3 gyz = backward_diff(v, ix, iy + 1, iz + 1, t, half), gxyz = backward_diff(v, ix + 1, iy + 1, iz + 1, t, half);
    const f3 deriv = lerp3(lerp3(lerp3(g0, gx, fx), lerp3(gy, gxy, fx), fy),
                           lerp3(lerp3(gz, gxz, fx), lerp3(gyz, gxyz, fx), fy), fz);
    return div33(deriv, voxel_size_units(v));
}

static inline void raycast_pixel(const kfo_image* imgdepth, const kfo_image* norm, const kfo_image* img,
                                 const kfo_volume* vol, const float* T, const float* K, float near,
                                 float far, float trunc_dist, int subpix, int u, int v, touch_t* t,
                                 uint64_t* n_rays, uint64_t* n_steps, uint64_t* n_hits, int half)
{
    const f3 c_w = se3_translation(T);                               /* :40 */
    const f3 ray_c = unproject1(K, (float)u, (float)v);              /* :41 */
    const f3 ray_w = so3_mul(T, ray_c);                              /* :42 */

    const f3 tminbound = div33(sub3(box_min(vol), c_w), ray_w);      /* :46 */
    const f3 tmaxbound = div33(sub3(box_max(vol), c_w), ray_w);      /* :47 */
    const f3 tmin = {fminf(tminbound.x, tmaxbound.x), fminf(tminbound.y, tmaxbound.y), fminf(tminbound.z, tmaxbound.z)};
    const f3 tmax = {fmaxf(tminbound.x, tmaxbound.x), fmaxf(tminbound.y, tmaxbound.y), fmaxf(tminbound.z, tmaxbound.z)};
    const float max_tmin = fmaxf(fmaxf(fmaxf(tmin.x, tmin.y), tmin.z), near); /* :50 */
    const float min_tmax = fminf(fminf(fminf(tmax.x, tmax.y), tmax.z), far);  /* :51 */

    float depth = 0.0f;
    if (max_tmin < min_tmax) {                                       /* :56 */
        float lambda = max_tmin;
        float last_sdf = NAN;                                        /* InvalidValue<float>, :59 */
        const float min_delta_lambda = voxel_size_units(vol).x;      /* :60 */
        float delta_lambda = 0;
        ++*n_rays;
        while (lambda < min_tmax) {                                  /* :64 */
            const f3 pos_w = add3(c_w, scale3(ray_w, lambda));
            const float sdf = trilinear_clamped(vol, pos_w, t, half);
            ++*n_steps;
            if (sdf <= 0) {                                          /* :68 */
                if (last_sdf > 0) {
                    if (subpix) lambda = lambda + delta_lambda * sdf / (last_sdf - sdf); /* :72 */
                    depth = lambda;
                }
                break;
            }
            delta_lambda = sdf > 0 ? fmaxf(sdf, min_delta_lambda) : trunc_dist; /* :78 */
            lambda += delta_lambda;
            last_sdf = sdf;
        }
    }

    float* pd = &((float*)img_row(imgdepth, (size_t)v))[u];
    float* pi = &((float*)img_row(img, (size_t)v))[u];
    f4* pn = &((f4*)img_row(norm, (size_t)v))[u];
    if (depth > 0) {                                                 /* :92 */
        const f3 pos_w = add3(c_w, scale3(ray_w, depth));            /* :85 */
        const f3 _n_w = units_backward_diff(vol, pos_w, t, half);          /* :86 */
        const float len_n_w = length3(_n_w);
        const f3 n_w = len_n_w > 0 ? div3s(_n_w, len_n_w) : mk3(0, 0, 1); /* :88 */
        const f3 n_c = so3_mul_inv(T, n_w);                          /* :89 */
        const f3 p_c = scale3(ray_c, depth);                         /* :90 */
        *pd = depth;
        *pi = phong_shade(p_c, n_c);
        f4 N = {n_c.x, n_c.y, n_c.z, 1.0f};
        *pn = N;
        ++*n_hits;
    } else { /* the reference evaluates the normal here too, but discards it (:85-102) */
        *pd = NAN;
        *pi = 0;
        f4 Z = {0, 0, 0, 0};
        *pn = Z;
    }
}

static void raycast_any(const kfo_image* depth, const kfo_image* norm, const kfo_image* img,
                        const kfo_volume* vol, const float T_wc[12], const float K[4], float near,
                        float far, float trunc, int subpix, int nthreads, kfo_raycast_stats* stats, int half)
{
    uint64_t rays = 0, steps = 0, hits = 0;
    const int nt = pick_threads(nthreads);
    (void)nt;
#pragma omp parallel for num_threads(nt) schedule(dynamic, 4) reduction(+ : rays, steps, hits)
    for (int v = 0; v < (int)img->h; ++v)
        for (int u = 0; u < (int)img->w; ++u)
            raycast_pixel(depth, norm, img, vol, T_wc, K, near, far, trunc, subpix, u, v, NULL, &rays, &steps, &hits, half);
    if (stats) { stats->rays = rays; stats->steps = steps; stats->hits = hits; }
}

void kfo_raycast_sdf(const kfo_image* depth, const kfo_image* norm, const kfo_image* img,
                     const kfo_volume* vol, const float T_wc[12], const float K[4], float near,
                     float far, float trunc, int subpix, int nthreads, kfo_raycast_stats* stats)
{
    raycast_any(depth, norm, img, vol, T_wc, K, near, far, trunc, subpix, nthreads, stats, 0);
}
void kfo_raycast_sdf_h(const kfo_image* depth, const kfo_image* norm, const kfo_image* img,
                       const kfo_volume* vol, const float T_wc[12], const float K[4], float near,
                       float far, float trunc, int subpix, int nthreads, kfo_raycast_stats* stats)
{
    raycast_any(depth, norm, img, vol, T_wc, K, near, far, trunc, subpix, nthreads, stats, 1);
}

void kfo_raycast_sdf_touch(const kfo_image* depth, const kfo_image* norm, const kfo_image* img,
                           const kfo_volume* vol, const float T_wc[12], const float K[4], float near,
                           float far, float trunc, int subpix, uint8_t* bitmap, kfo_raycast_stats* stats)
{
    uint64_t rays = 0, steps = 0, hits = 0;
    touch_t t = {bitmap, vol};
    for (int v = 0; v < (int)img->h; ++v)
        for (int u = 0; u < (int)img->w; ++u)
            raycast_pixel(depth, norm, img, vol, T_wc, K, near, far, trunc, subpix, u, v, &t, &rays, &steps, &hits, 0);
    if (stats) { stats->rays = rays; stats->steps = steps; stats->hits = hits; }
}

/* ============================================================================
 * Analytic renderers -- src/cu_raycast.cu:202-310
 * ========================================================================== */
void kfo_raycast_box(const kfo_image* imgd, const float T[12], const float K[4], const float bmin[3],
                     const float bmax[3])
{
    const f3 lo = {bmin[0], bmin[1], bmin[2]}, hi = {bmax[0], bmax[1], bmax[2]};
    for (int v = 0; v < (int)imgd->h; ++v)
        for (int u = 0; u < (int)imgd->w; ++u) {
            const f3 c_w = se3_translation(T);
            const f3 ray_w = so3_mul(T, unproject1(K, (float)u, (float)v));
            const f3 a = div33(sub3(lo, c_w), ray_w), b = div33(sub3(hi, c_w), ray_w);
            const f3 tmin = {fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)};
            const f3 tmax = {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)};
            const float max_tmin = fmaxf(fmaxf(tmin.x, tmin.y), tmin.z);
            const float min_tmax = fminf(fminf(tmax.x, tmax.y), tmax.z);
            ((float*)img_row(imgd, (size_t)v))[u] = (max_tmin < min_tmax) ? max_tmin : NAN;
        }
}

void kfo_raycast_sphere(const kfo_image* imgd, const kfo_image* img, const float T[12], const float K[4],
                        const float center[3], float r)
{
    const f3 center_c = se3_mul_inv(T, mk3(center[0], center[1], center[2])); /* :276 */
    for (int v = 0; v < (int)imgd->h; ++v)
        for (int u = 0; u < (int)imgd->w; ++u) {
            const f3 ray_c = unproject1(K, (float)u, (float)v);
            const float ldotc = dot3(ray_c, center_c);
            const float lsq = dot3(ray_c, ray_c);
            const float csq = dot3(center_c, center_c);
            /* `sqrt` of a float argument: the float overload in device code (:256) */
            const float depth = (ldotc - sqrtf(ldotc * ldotc - lsq * (csq - r * r))) / lsq;
            float* pd = &((float*)img_row(imgd, (size_t)v))[u];
            const float prev = *pd;
            if (depth > 0 && (depth < prev || !isfinite(prev))) {
                *pd = depth;
                if (img && img->ptr) {
                    const f3 p_c = scale3(ray_c, depth);
                    const f3 n_c = sub3(p_c, center_c);
                    ((float*)img_row(img, (size_t)v))[u] = phong_shade(p_c, div3s(n_c, length3(n_c)));
                }
            }
        }
}

void kfo_raycast_plane(const kfo_image* imgd, const kfo_image* img, const float T[12], const float K[4],
                       const float n_w[3])
{
    /* Plane_b_from_a, MatUtils.h:474-488 */
    const float dn = T_(0, 3) * n_w[0] + T_(1, 3) * n_w[1] + T_(2, 3) * n_w[2] + 1.0f;
    const f3 n_c = {(T_(0, 0) * n_w[0] + T_(1, 0) * n_w[1] + T_(2, 0) * n_w[2]) / dn,
                    (T_(0, 1) * n_w[0] + T_(1, 1) * n_w[1] + T_(2, 1) * n_w[2]) / dn,
                    (T_(0, 2) * n_w[0] + T_(1, 2) * n_w[1] + T_(2, 2) * n_w[2]) / dn};
    for (int v = 0; v < (int)imgd->h; ++v)
        for (int u = 0; u < (int)imgd->w; ++u) {
            const f3 ray_c = unproject1(K, (float)u, (float)v);
            const float depth = -1 / dot3(n_c, ray_c);
            float* pd = &((float*)img_row(imgd, (size_t)v))[u];
            const float prev = *pd;
            if (depth > 0 && (depth < prev || !isfinite(prev))) {
                if (img && img->ptr) {
                    const f3 p_c = scale3(ray_c, depth);
                    ((float*)img_row(img, (size_t)v))[u] = phong_shade(p_c, div3s(n_c, length3(n_c)));
                }
                *pd = depth;
            }
        }
}

/* ============================================================================
 * Synthetic scenes of SURVEY 8(d) (own code, not reference)
 * ========================================================================== */
void kfo_render_scene(const kfo_image* depth, int scene, const float T[12], const float K[4])
{
    for (int v = 0; v < (int)depth->h; ++v)
        for (int u = 0; u < (int)depth->w; ++u) {
            const f3 c = se3_translation(T);
            const f3 r = so3_mul(T, unproject1(K, (float)u, (float)v));
            float best = INFINITY;
            if (scene == 0) {
                /* room: exit distance from box x,y in [-0.9,0.9], z in [-10,3.8] */
                const f3 lo = {-0.9f, -0.9f, -10.0f}, hi = {0.9f, 0.9f, 3.8f};
                const f3 a = div33(sub3(lo, c), r), b = div33(sub3(hi, c), r);
                const float tx = fmaxf(a.x, b.x), ty = fmaxf(a.y, b.y), tz = fmaxf(a.z, b.z);
                const float texit = fminf(fminf(tx, ty), tz);
                if (texit > 0) best = texit;
                /* sphere c=(0,0,3) r=0.5 */
                const f3 sc = {0.0f, 0.0f, 3.0f};
                const f3 oc = sub3(sc, c);
                const float ldotc = dot3(r, oc), lsq = dot3(r, r), csq = dot3(oc, oc);
                const float disc = ldotc * ldotc - lsq * (csq - 0.5f * 0.5f);
                if (disc >= 0) {
                    const float ts = (ldotc - sqrtf(disc)) / lsq;
                    if (ts > 0 && ts < best) best = ts;
                }
            } else {
                /* wall z = 5.95 */
                const float t = (5.95f - c.z) / r.z;
                if (t > 0) best = t;
            }
            ((float*)img_row(depth, (size_t)v))[u] = isfinite(best) ? best : NAN;
        }
}

/* ============================================================================
 * Host-side ROI helpers
 * ========================================================================== */
void kfo_fit_to_frustum(float bmin[3], float bmax[3], const float T[12], float w, float h,
                        const float K[4], float near, float far) /* BoundingBox.h:72-96 */
{
    const f3 c_w = se3_translation(T);
    const f3 rays[4] = {so3_mul(T, mk3((0 - U0) / FU, (0 - V0) / FV, 1)), so3_mul(T, mk3((w - U0) / FU, (0 - V0) / FV, 1)),
                        so3_mul(T, mk3((0 - U0) / FU, (h - V0) / FV, 1)), so3_mul(T, mk3((w - U0) / FU, (h - V0) / FV, 1))};
    f3 lo = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    f3 hi = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    const float ds[2] = {near, far};
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < 4; ++i) {
            const f3 p = add3(c_w, scale3(rays[i], ds[k]));
            hi = mk3(fmaxf(p.x, hi.x), fmaxf(p.y, hi.y), fmaxf(p.z, hi.z));
            lo = mk3(fminf(p.x, lo.x), fminf(p.y, lo.y), fminf(p.z, lo.z));
        }
    bmin[0] = lo.x; bmin[1] = lo.y; bmin[2] = lo.z;
    bmax[0] = hi.x; bmax[1] = hi.y; bmax[2] = hi.z;
}

void kfo_sub_bounding_volume(kfo_volume* out, const kfo_volume* vol, const float rmin[3],
                             const float rmax[3]) /* BoundedVolume.h:137-165 */
{
    const f3 bs = box_size(vol);
    const f3 min_fv = div33(sub3(mk3(rmin[0], rmin[1], rmin[2]), box_min(vol)), bs);
    const f3 max_fv = div33(sub3(mk3(rmax[0], rmax[1], rmax[2]), box_min(vol)), bs);
    const float W1 = (float)(vol->w - 1), H1 = (float)(vol->h - 1), D1 = (float)(vol->d - 1);
    const int min_v[3] = {(int)fmaxf(W1 * min_fv.x, 0), (int)fmaxf(H1 * min_fv.y, 0), (int)fmaxf(D1 * min_fv.z, 0)};
    const int max_v[3] = {(int)fminf(ceilf(W1 * max_fv.x), W1), (int)fminf(ceilf(H1 * max_fv.y), H1),
                          (int)fminf(ceilf(D1 * max_fv.z), D1)};
    int size_v[3];
    for (int i = 0; i < 3; ++i) {
        size_v[i] = (max_v[i] - min_v[i]) + 1;
        if (size_v[i] < 0) size_v[i] = 0;
    }
    const f3 nlo = voxel_position(vol, min_v[0], min_v[1], min_v[2]);
    const f3 nhi = voxel_position(vol, max_v[0], max_v[1], max_v[2]);
    /* Volume::SubVolume (Volume.h:305-311): same pitches, offset pointer */
    out->pitch = vol->pitch;
    out->img_pitch = vol->img_pitch;
    out->ptr = (void*)&vol_row(vol, (size_t)min_v[1], (size_t)min_v[2])[min_v[0]];
    out->w = (size_t)size_v[0];
    out->h = (size_t)size_v[1];
    out->d = (size_t)size_v[2];
    out->boxmin[0] = nlo.x; out->boxmin[1] = nlo.y; out->boxmin[2] = nlo.z;
    out->boxmax[0] = nhi.x; out->boxmax[1] = nhi.y; out->boxmax[2] = nhi.z;
}

/* ============================================================================
 * Point-wise helper exports (tests pin these against the reference headers)
 * ========================================================================== */
float kfo_trilinear(const kfo_volume* vol, const float pos_w[3])
{
    return trilinear_clamped(vol, mk3(pos_w[0], pos_w[1], pos_w[2]), NULL, 0);
}
void kfo_gradient(const kfo_volume* vol, const float pos_w[3], float out[3])
{
    const f3 g = units_backward_diff(vol, mk3(pos_w[0], pos_w[1], pos_w[2]), NULL, 0);
    out[0] = g.x; out[1] = g.y; out[2] = g.z;
}
float kfo_phong_shade(const float p_c[3], const float n_c[3]) /* PhongShade, cu_raycast.cu:14-28 */
{
    const f3 p = {p_c[0], p_c[1], p_c[2]}, n = {n_c[0], n_c[1], n_c[2]};
    return phong_shade(p, n);
}

void kfo_sdf_accumulate(float val, float w, float old_val, float old_w, float max_w, float out[2])
{
    /* Sdf.h:25-32 then :22-24 */
    if (old_w > 0) {
        val = (w * val + old_w * old_val);
        w += old_w;
        val /= w;
    }
    w = fminf(w, max_w);
    out[0] = val; out[1] = w;
}
void kfo_intrinsics_level(float out[4], const float K[4], int level) /* ImageIntrinsics.h:137-142 */
{
    const float scale = 1.0f / (float)(1 << level);
    out[0] = scale * FU; out[1] = scale * FV;
    out[2] = scale * (U0 + 0.5f) - 0.5f; out[3] = scale * (V0 + 0.5f) - 0.5f;
}
void kfo_voxel_position(const kfo_volume* vol, int x, int y, int z, float out[3])
{
    const f3 p = voxel_position(vol, x, y, z);
    out[0] = p.x; out[1] = p.y; out[2] = p.z;
}

/* ============================================================================
 * fp16-cell volume initialisers (config C5)
 * ========================================================================== */
void kfo_sdf_reset_h(const kfo_volume* vol, float trunc)
{
    sdfh_t* b = (sdfh_t*)vol->ptr;
    sdfh_t* e = volh_row(vol, vol->h - 1, vol->d - 1) + vol->w;
    const sdfh_t v = {f2h(trunc), f2h(0.0f)};
    for (; b < e; ++b) *b = v;
}

void kfo_sdf_sphere_h(const kfo_volume* vol, const float center[3], float r)
{
    const int X = (int)(vol->w / 8) * 8, Y = (int)(vol->h / 8) * 8, Z = (int)(vol->d / 8) * 8;
    const f3 c = {center[0], center[1], center[2]};
    for (int z = 0; z < Z; ++z)
        for (int y = 0; y < Y; ++y) {
            sdfh_t* row = volh_row(vol, (size_t)y, (size_t)z);
            for (int x = 0; x < X; ++x) {
                const float dist = length3(sub3(voxel_position(vol, x, y, z), c));
                sdfh_t s = {f2h(dist - r), f2h(1.0f)};
                row[x] = s;
            }
        }
}

/* ============================================================================
 * Frame pre-amble: ElementwiseScaleBias, BoxHalfIgnoreInvalid
 * ========================================================================== */
void kfo_elementwise_scale_bias_f32(const kfo_image* b, const kfo_image* a, float s, float offset) /* cu_operations.cu:39-49 */
{
    for (size_t y = 0; y < b->h; ++y)
        for (size_t x = 0; x < b->w; ++x)
            ((float*)img_row(b, y))[x] = s * ((const float*)img_row(a, y))[x] + offset;
}

void kfo_box_half_ignore_invalid_f32(const kfo_image* out, const kfo_image* in) /* cu_resample.cu:89-111 */
{
    for (size_t y = 0; y < out->h; ++y)
        for (size_t x = 0; x < out->w; ++x) {
            const float* tl = (const float*)img_row(in, 2 * y) + 2 * x;
            const float* bl = (const float*)img_row(in, 2 * y + 1) + 2 * x;
            const float v[4] = {tl[0], tl[1], bl[0], bl[1]};
            int n = 0;
            float sum = 0;
            for (int i = 0; i < 4; ++i)
                if (isfinite(v[i])) { sum += v[i]; n++; }
            ((float*)img_row(out, y))[x] = n > 0 ? (sum / n) : NAN;
        }
}

/* ============================================================================
 * Exact multi-GPU march (SURVEY 8(e) "exact variant"): one round of one rank.
 * The march of cu_raycast.cu:58-81 with its state (lambda, last_sdf, delta) carried between Z-slabs:
 * a rank advances a ray while the trilinear base cell of the current sample is one it owns.
 * State: 9 dense planes of h*w floats: 0 lambda, 1 last_sdf, 2 delta, 3 status (0 marching, 1 hit, 2 miss,
 * 3 hit awaiting its normal), 4 touched-this-round, 5-7 normal, 8 shade.  A hit's depth is its lambda; the
 * normal is evaluated by the rank that owns the gradient's base plane.
 * ========================================================================== */
void kfo_raycast_sdf_slab(float* state, int init, const kfo_volume* vol, const kfo_slab* slab, int own_lo, int own_hi,
                          int w, int h, const float T[12], const float K[4], float near, float far,
                          float trunc_dist, int subpix)
{
    const int half = 0;
    kfo_volume fv = *vol; /* full-volume geometry over a virtual base pointer */
    fv.ptr = (unsigned char*)vol->ptr - (ptrdiff_t)slab->z_offset * (ptrdiff_t)vol->img_pitch;
    fv.d = slab->full_d;
    fv.boxmin[2] = slab->full_zmin;
    fv.boxmax[2] = slab->full_zmax;
    const int avail_lo = (int)slab->z_offset, avail_hi = (int)(slab->z_offset + vol->d);
    for (int v = 0; v < h; ++v)
        for (int u = 0; u < w; ++u) {
            const size_t P = (size_t)w * (size_t)h;
            float* st = state + (size_t)v * (size_t)w + (size_t)u;
            const f3 c_w = se3_translation(T);
            const f3 ray_c = unproject1(K, (float)u, (float)v);
            const f3 ray_w = so3_mul(T, ray_c);
            const f3 ta = div33(sub3(box_min(&fv), c_w), ray_w), tb = div33(sub3(box_max(&fv), c_w), ray_w);
            const float max_tmin = fmaxf(fmaxf(fmaxf(fminf(ta.x, tb.x), fminf(ta.y, tb.y)), fminf(ta.z, tb.z)), near);
            const float min_tmax = fminf(fminf(fminf(fmaxf(ta.x, tb.x), fmaxf(ta.y, tb.y)), fmaxf(ta.z, tb.z)), far);
            float lambda, last_sdf, delta, status;
            if (init) {
                lambda = max_tmin; last_sdf = NAN; delta = 0.f;
                status = (max_tmin < min_tmax) ? 0.f : 2.f;
                for (int i = 5; i < 9; ++i) st[i * P] = 0.f;
            } else {
                lambda = st[0]; last_sdf = st[P]; delta = st[2 * P]; status = st[3 * P];
            }
            const float lambda_in = lambda, status_in = status;
            if (status == 0.f) {
                const float min_delta = voxel_size_units(&fv).x;
                for (;;) {
                    if (!(lambda < min_tmax)) { status = 2.f; break; }
                    const f3 pos = add3(c_w, scale3(ray_w, lambda));
                    const float pfz = ((pos.z - fv.boxmin[2]) / (fv.boxmax[2] - fv.boxmin[2])) * ((float)fv.d - 1.f);
                    const int iz = (int)fmaxf(fminf((float)(fv.d - 2), floorf(pfz)), 0);
                    if (iz < own_lo || iz >= own_hi || iz < avail_lo || iz + 1 >= avail_hi) break;
                    const float sdf = trilinear_clamped(&fv, pos, NULL, half);
                    if (sdf <= 0) {
                        if (last_sdf > 0) {
                            if (subpix) lambda = lambda + delta * sdf / (last_sdf - sdf);
                            status = 3.f;
                        } else {
                            status = 2.f;
                        }
                        break;
                    }
                    delta = sdf > 0 ? fmaxf(sdf, min_delta) : trunc_dist;
                    lambda += delta;
                    last_sdf = sdf;
                }
            }
            if (status == 3.f) {
                const f3 pos = add3(c_w, scale3(ray_w, lambda));
                const float pfz = ((pos.z - fv.boxmin[2]) / (fv.boxmax[2] - fv.boxmin[2])) * ((float)fv.d - 1.f);
                const int gz = (int)fmaxf(fminf((float)(fv.d - 2), floorf(pfz)), 1);
                if (gz >= own_lo && gz < own_hi && gz - 1 >= avail_lo && gz + 1 < avail_hi) {
                    const f3 g = units_backward_diff(&fv, pos, NULL, half);
                    const float len = length3(g);
                    const f3 n_w = len > 0 ? div3s(g, len) : mk3(0, 0, 1);
                    const f3 n_c = so3_mul_inv(T, n_w);
                    st[5 * P] = n_c.x; st[6 * P] = n_c.y; st[7 * P] = n_c.z;
                    st[8 * P] = phong_shade(scale3(ray_c, lambda), n_c);
                    status = 1.f;
                }
            }
            st[0] = lambda; st[P] = last_sdf; st[2 * P] = delta; st[3 * P] = status;
            st[4 * P] = (lambda != lambda_in || status != status_in) ? 1.0f : 0.0f;
        }
}

/* ============================================================================
 * Projective point-to-plane ICP: cu_model_refinement.cu:541-608.
 * Per-pixel system as KernPoseRefinementProjectiveIcpPointPlane builds it, block sum in the order of
 * SumLeastSquaresSystem::ReducePutBlock (LeastSquareSum.h:71-85: s[t] += s[t+S], S = n/2 .. 1), block
 * geometry of InitDimFromOutputImage(dPl, 16, 16) (launch_utils.h:61-65).  The reference sums the blocks
 * with thrust::reduce (order unspecified); the order fixed here is the one the HIP path uses: 256 partial
 * sums, partial t = 0 + block t + block t+256 + ..., then the same tree over the 256 partials.
 * ========================================================================== */
static void lss_add(kfo_lss6* a, const kfo_lss6* b) /* LeastSquaresSystem::operator+= (Mat.h:496-504) */
{
    for (int i = 0; i < 6; ++i) a->JTy[i] += b->JTy[i];
    for (int i = 0; i < 21; ++i) a->JTJ[i] += b->JTJ[i];
    a->sqErr += b->sqErr;
    a->obs += b->obs;
}
static void lss_tree(kfo_lss6* s, unsigned n)
{
    for (unsigned S = n / 2; S > 0; S >>= 1)
        for (unsigned t = 0; t < S; ++t) lss_add(&s[t], &s[t + S]);
}
static unsigned gcd_u(unsigned a, unsigned b) { return b == 0 ? a : gcd_u(b, a % b); }

void kfo_icp_block_dims(size_t w, size_t h, unsigned out[4])
{
    out[0] = gcd_u((unsigned)w, 16);
    out[1] = gcd_u((unsigned)h, 16);
    out[2] = (unsigned)(w / out[0]);
    out[3] = (unsigned)(h / out[1]);
}

void kfo_icp_point_plane(const kfo_image* Pl, const kfo_image* Pr_img, const kfo_image* Nr_img, const float KT_lr[12],
                         const float T_rl[12], float c, const kfo_image* debug, kfo_lss6* out, kfo_lss6* block_sums)
{
    unsigned g[4];
    memset(out, 0, sizeof(*out));
    if (Pl->w == 0 || Pl->h == 0) return;
    kfo_icp_block_dims(Pl->w, Pl->h, g);
    const unsigned bx = g[0], by = g[1], gx = g[2], gy = g[3], n = bx * by;
    kfo_lss6 partial[256];
    memset(partial, 0, sizeof(partial));
    for (unsigned bj = 0; bj < gy; ++bj)
        for (unsigned bi = 0; bi < gx; ++bi) {
            kfo_lss6 s[256];
            memset(s, 0, sizeof(s)); /* ZeroThisObs */
            for (unsigned ty = 0; ty < by; ++ty)
                for (unsigned tx = 0; tx < bx; ++tx) {
                    const unsigned u = bi * bx + tx, v = bj * by + ty;
                    kfo_lss6* sum = &s[ty * bx + tx];
                    const f4 Pr = *((const f4*)((const unsigned char*)Pr_img->ptr + (size_t)v * Pr_img->pitch) + u);
                    const f4 Nr = *((const f4*)((const unsigned char*)Nr_img->ptr + (size_t)v * Nr_img->pitch) + u);
                    const f3 KPl = se3_mul(KT_lr, mk3(Pr.x, Pr.y, Pr.z));
                    const float plx = KPl.x / KPl.z, ply = KPl.y / KPl.z; /* dn(): Mat.h:623-627 */
                    f4 dbg;
                    if (isfinite(Pr.z) && Nr.w == 1.0f && 3.0f <= plx && plx < ((float)Pl->w - 3.0f) && 3.0f <= ply &&
                        ply < ((float)Pl->h - 3.0f)) { /* Image::InBounds(pl, 3), Image.h:288-291 */
                        /* GetNearestNeighbour = Get(u + 0.5, v + 0.5): double sum truncated to int (Image.h:337-340) */
                        const int nx = (int)((double)plx + 0.5), ny = (int)((double)ply + 0.5);
                        const f4 P = *((const f4*)((const unsigned char*)Pl->ptr + (size_t)ny * Pl->pitch) + nx);
                        if (isfinite(P.z)) {
                            const f3 _Pr = se3_mul(T_rl, mk3(P.x, P.y, P.z));
                            const f3 Dr = mk3(_Pr.x - Pr.x, _Pr.y - Pr.y, _Pr.z - Pr.z);
                            const f3 N = mk3(Nr.x, Nr.y, Nr.z);
                            const float y = dot3(Dr, N);
                            float J[6]; /* -dot(SE3gen_i mul _Pr, Nr): MatUtils.h:379-402 */
                            J[0] = -dot3(mk3(1.f, 0.f, 0.f), N);
                            J[1] = -dot3(mk3(0.f, 1.f, 0.f), N);
                            J[2] = -dot3(mk3(0.f, 0.f, 1.f), N);
                            J[3] = -dot3(mk3(0.f, -_Pr.z, _Pr.y), N);
                            J[4] = -dot3(mk3(_Pr.z, 0.f, -_Pr.x), N);
                            J[5] = -dot3(mk3(-_Pr.y, _Pr.x, 0.f), N);
                            const float absr = fabsf(y); /* LSReweightTukey, reweighting.h:22-28 */
                            const float roc = y / c;
                            const float omroc2 = 1.0f - roc * roc;
                            const float tukey = (absr <= c) ? omroc2 * omroc2 : 0.0f;
                            const float w = (1.0f / Pr.z) * tukey;
                            const float yw = y * w;
                            int i = 0;
                            for (int r = 0; r < 6; ++r) sum->JTy[r] = J[r] * yw;          /* mul_aTb, Mat.h:263-275 */
                            for (int r = 0; r < 6; ++r)
                                for (int cc = 0; cc <= r; ++cc) sum->JTJ[i++] = J[r] * J[cc] * w; /* OuterProduct, Mat.h:457-468 */
                            sum->obs = 1;
                            sum->sqErr = y * y;
                            dbg.x = dbg.y = dbg.z = absr; dbg.w = 1.f;
                        } else {
                            dbg.x = 0.f; dbg.y = 0.f; dbg.z = 1.f; dbg.w = 1.f;
                        }
                    } else {
                        dbg.x = 1.f; dbg.y = 0.f; dbg.z = 0.f; dbg.w = 1.f;
                    }
                    if (debug && debug->ptr) *((f4*)((unsigned char*)debug->ptr + (size_t)v * debug->pitch) + u) = dbg;
                }
            lss_tree(s, n);
            const unsigned bid = bj * gx + bi;
            if (block_sums) block_sums[bid] = s[0];
            lss_add(&partial[bid % 256], &s[0]);
        }
    lss_tree(partial, 256);
    *out = partial[0];
}

/* ============================================================================
 * Colour fusion / colour raycast.
 * ========================================================================== */
static inline float* cvol_row(const kfo_volume* v, size_t y, size_t z)
{
    return (float*)((unsigned char*)v->ptr + z * v->img_pitch + y * v->pitch);
}

/* SdfReset(BoundedVolume<float>) = vol.Fill(0.5) over the contiguous span (cu_sdffusion.cu:166-169, Volume.h:343-356) */
void kfo_color_reset(const kfo_volume* cv)
{
    const size_t n = ((cv->d - 1) * cv->img_pitch + (cv->h - 1) * cv->pitch + cv->w * 4) / 4;
    float* p = (float*)cv->ptr;
    for (size_t i = 0; i < n; ++i) p[i] = 0.5f;
}

typedef struct { unsigned char x, y, z; } uc3;
/* lerp(uchar3, uchar3, float), sampling.h:23-30: a.x + t*(b.x - a.x) with the difference taken in int */
static inline f3 lerp_uc3(uc3 a, uc3 b, float t)
{
    return mk3((float)a.x + t * (float)((int)b.x - (int)a.x), (float)a.y + t * (float)((int)b.y - (int)a.y),
               (float)a.z + t * (float)((int)b.z - (int)a.z));
}

/* KernSdfFuse with colour, cu_sdffusion.cu:70-118; launch (16,16) over x,y, z loop over all of vol.d (:132-135) */
uint64_t kfo_sdf_fuse_color(const kfo_volume* vol, const kfo_volume* cvol, const kfo_image* depth, const kfo_image* normals,
                            const float T[12], const float K[4], const kfo_image* img, const float T_iw[12], const float Kimg[4],
                            float trunc_dist, float max_w, float mincostheta, int full_extent, int nthreads)
{
    const int X = full_extent ? (int)vol->w : (int)(vol->w / 16) * 16;
    const int Y = full_extent ? (int)vol->h : (int)(vol->h / 16) * 16;
    const int Z = (int)vol->d;
    uint64_t updated = 0;
    const int nt = pick_threads(nthreads);
    (void)nt;
#pragma omp parallel for num_threads(nt) schedule(static) reduction(+ : updated)
    for (int z = 0; z < Z; ++z)
        for (int y = 0; y < Y; ++y)
            for (int x = 0; x < X; ++x) {
                const f3 P_w = voxel_position(vol, x, y, z);
                const f3 P_c = se3_mul(T, P_w);
                const float pu = U0 + FU * P_c.x / P_c.z, pv = V0 + FV * P_c.y / P_c.z;  /* K.Project */
                const f3 P_i = se3_mul(T_iw, P_w);
                const float qu = Kimg[2] + Kimg[0] * P_i.x / P_i.z, qv = Kimg[3] + Kimg[1] * P_i.y / P_i.z;
                const float b = 2.0f;
                if (!(b <= pu && pu < ((float)depth->w - b) && b <= pv && pv < ((float)depth->h - b))) continue;
                if (!(b <= qu && qu < ((float)img->w - b) && b <= qv && qv < ((float)img->h - b))) continue;
                const float vd = P_c.z;
                const float ix = floorf(pu), iy = floorf(pv);
                const float fx = pu - ix, fy = pv - iy;
                const float* dbl = (const float*)img_row(depth, (size_t)iy) + (size_t)ix;
                const float* dtl = (const float*)img_row(depth, (size_t)(iy + 1)) + (size_t)ix;
                const float md = lerpf(lerpf(dbl[0], dbl[1], fx), lerpf(dtl[0], dtl[1], fx), fy);
                const f4* nbl = (const f4*)img_row(normals, (size_t)iy) + (size_t)ix;
                const f4* ntl = (const f4*)img_row(normals, (size_t)(iy + 1)) + (size_t)ix;
                f3 mdn;
                mdn.x = lerpf(lerpf(nbl[0].x, nbl[1].x, fx), lerpf(ntl[0].x, ntl[1].x, fx), fy);
                mdn.y = lerpf(lerpf(nbl[0].y, nbl[1].y, fx), lerpf(ntl[0].y, ntl[1].y, fx), fy);
                mdn.z = lerpf(lerpf(nbl[0].z, nbl[1].z, fx), lerpf(ntl[0].z, ntl[1].z, fx), fy);
                /* c = ConvertPixel<float,float3>(img.GetBilinear<float3>(p_i)) / 255.0  (:98; the division is in double) */
                const float jx = floorf(qu), jy = floorf(qv);
                const float gx = qu - jx, gy = qv - jy;
                const uc3* cbl = (const uc3*)img_row(img, (size_t)jy) + (size_t)jx;
                const uc3* ctl = (const uc3*)img_row(img, (size_t)(jy + 1)) + (size_t)jx;
                const f3 rgb = lerp3(lerp_uc3(cbl[0], cbl[1], gx), lerp_uc3(ctl[0], ctl[1], gx), gy);
                const float c = (float)((double)((rgb.x + rgb.y + rgb.z) / 3.0f) / 255.0);

                const float costheta = dot3(mdn, P_c) / -length3(P_c);
                const float sd = costheta * (md - vd);
                const float w = costheta * 1.0f / vd;
                if (sd <= -trunc_dist) continue;
                if (isfinite(md) && isfinite(w) && costheta > mincostheta) {
                    sdf_t* cell = &vol_row(vol, (size_t)y, (size_t)z)[x];
                    const sdf_t curvol = *cell;
                    sdf_t sdf = {clampf(sd, -trunc_dist, trunc_dist), w};
                    if (curvol.w > 0) {
                        sdf.val = (sdf.w * sdf.val + curvol.w * curvol.val);
                        sdf.w += curvol.w;
                        sdf.val /= sdf.w;
                    }
                    sdf.w = fminf(sdf.w, max_w);
                    *cell = sdf;
                    float* cc = &cvol_row(cvol, (size_t)y, (size_t)z)[x];
                    *cc = (w * c + *cc * curvol.w) / (w + curvol.w);
                    ++updated;
                }
            }
    return updated;
}

/* Volume<float>::GetFractionalTrilinearClamped through BoundedVolume<float>::GetUnitsTrilinearClamped */
static inline float trilinear_clamped_f32(const kfo_volume* v, f3 pos_w)
{
    const f3 pos_v = div33(sub3(pos_w, box_min(v)), box_size(v));
    const f3 pf = {pos_v.x * ((float)v->w - 1.f), pos_v.y * ((float)v->h - 1.f), pos_v.z * ((float)v->d - 1.f)};
    const int ix = (int)fmaxf(fminf((float)(v->w - 2), floorf(pf.x)), 0);
    const int iy = (int)fmaxf(fminf((float)(v->h - 2), floorf(pf.y)), 0);
    const int iz = (int)fmaxf(fminf((float)(v->d - 2), floorf(pf.z)), 0);
    const float fx = pf.x - (float)ix, fy = pf.y - (float)iy, fz = pf.z - (float)iz;
    const float* r00 = cvol_row(v, (size_t)iy, (size_t)iz), *r10 = cvol_row(v, (size_t)iy + 1, (size_t)iz);
    const float* r01 = cvol_row(v, (size_t)iy, (size_t)iz + 1), *r11 = cvol_row(v, (size_t)iy + 1, (size_t)iz + 1);
    return lerpf(lerpf(lerpf(r00[ix], r00[ix + 1], fx), lerpf(r10[ix], r10[ix + 1], fx), fy),
                 lerpf(lerpf(r01[ix], r01[ix + 1], fx), lerpf(r11[ix], r11[ix + 1], fx), fy), fz);
}

/* KernRaycastSdf with colour, cu_raycast.cu:119-189: the march and the normal of the grey variant; on a hit
 * img = colorVol.GetUnitsTrilinearClamped(c_w + depth * ray_w) instead of the Phong shade. */
void kfo_raycast_sdf_color(const kfo_image* depth, const kfo_image* norm, const kfo_image* img, const kfo_volume* vol,
                           const kfo_volume* cvol, const float T[12], const float K[4], float near, float far, float trunc,
                           int subpix, int nthreads)
{
    raycast_any(depth, norm, img, vol, T, K, near, far, trunc, subpix, nthreads, NULL, 0);
    for (int v = 0; v < (int)img->h; ++v)
        for (int u = 0; u < (int)img->w; ++u) {
            const float d = ((const float*)img_row(depth, (size_t)v))[u];
            if (!(d > 0)) continue; /* miss: depth NaN, img already 0 */
            const f3 c_w = se3_translation(T);
            const f3 ray_w = so3_mul(T, unproject1(K, (float)u, (float)v));
            ((float*)img_row(img, (size_t)v))[u] = trilinear_clamped_f32(cvol, add3(c_w, scale3(ray_w, d)));
        }
}

/* ============================================================================
 * Marching cubes: MarchingCubes.h:43-143 inside SaveMesh's loop nest (:226-232).
 * The case tables are passed in (the product's generated tables; tests also pass the reference's own).
 * verts / norms: 3 floats per vertex, colors: 4 floats per vertex, three vertices per triangle, emitted in the
 * reference's order (x outer, y, z inner).  With verts == NULL only counts.  Returns the number of triangles.
 * ========================================================================== */
uint64_t kfo_marching_cubes(const kfo_volume* vol, const kfo_volume* cvol, const unsigned char* ntris, const unsigned short* emask,
                            const signed char* tris, int tri_stride, float* verts, float* norms, float* colors)
{
    static const int OFF[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
    static const int EC[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
    const int half = 0;
    const int color_valid = cvol && cvol->ptr && cvol->w >= 8 && cvol->h >= 8 && cvol->d >= 8; /* BoundedVolume::IsValid */
    uint64_t nt = 0;
    const f3 fScale = voxel_size_units(vol);
    for (int x = 0; x < (int)vol->w - 1; ++x)
        for (int y = 0; y < (int)vol->h - 1; ++y)
            for (int z = 0; z < (int)vol->d - 1; ++z) {
                const f3 p = voxel_position(vol, x, y, z);
                float v[8];
                int finite = 1, flag = 0;
                for (int i = 0; i < 8; ++i) {
                    v[i] = vol_val(vol, x + OFF[i][0], y + OFF[i][1], z + OFF[i][2]);
                    if (!isfinite(v[i])) { finite = 0; break; }
                }
                if (!finite) continue;
                for (int i = 0; i < 8; ++i)
                    if (v[i] <= 0.0f) flag |= 1 << i;
                const int mask = emask[flag];
                if (mask == 0) continue;
                f3 ev[12], en[12];
                for (int e = 0; e < 12; ++e) {
                    if (!(mask & (1 << e))) continue;
                    const int c0 = EC[e][0], c1 = EC[e][1];
                    const double fDelta = v[c1] - v[c0];                                   /* fGetOffset :25-32 */
                    const float fOffset = fDelta == 0.0 ? 0.5f : (float)((0.0f - v[c0]) / fDelta);
                    const float dir[3] = {(float)(OFF[c1][0] - OFF[c0][0]), (float)(OFF[c1][1] - OFF[c0][1]), (float)(OFF[c1][2] - OFF[c0][2])};
                    ev[e] = mk3(p.x + ((float)OFF[c0][0] + fOffset * dir[0]) * fScale.x, p.y + ((float)OFF[c0][1] + fOffset * dir[1]) * fScale.y,
                                p.z + ((float)OFF[c0][2] + fOffset * dir[2]) * fScale.z);
                    const f3 deriv = units_backward_diff(vol, ev[e], NULL, half);
                    en[e] = div3s(deriv, length3(deriv));
                    if (!isfinite(en[e].x) || !isfinite(en[e].y) || !isfinite(en[e].z)) en[e] = mk3(0, 0, 0);
                }
                for (int t = 0; t < ntris[flag]; ++t) {
                    if (verts)
                        for (int c = 0; c < 3; ++c) {
                            const int e = tris[flag * tri_stride + 3 * t + c];
                            const uint64_t o = nt * 3 + (uint64_t)c;
                            verts[o * 3 + 0] = ev[e].x; verts[o * 3 + 1] = ev[e].y; verts[o * 3 + 2] = ev[e].z;
                            norms[o * 3 + 0] = en[e].x; norms[o * 3 + 1] = en[e].y; norms[o * 3 + 2] = en[e].z;
                            if (color_valid && colors) {
                                const float cc = trilinear_clamped_f32(cvol, ev[e]);
                                colors[o * 4 + 0] = cc; colors[o * 4 + 1] = cc; colors[o * 4 + 2] = cc; colors[o * 4 + 3] = 1.0f;
                            }
                        }
                    ++nt;
                }
            }
    return nt;
}

/* SdfDistance, cu_sdffusion.cu:200-225: dist(u,v) = vol.GetUnitsTrilinearClamped(T_wc * (depth(u,v) * K.Unproject(u,v))) */
void kfo_sdf_distance(const kfo_image* dist, const kfo_image* depth, const kfo_volume* vol, const float T[12], const float K[4])
{
    const int half = 0;
    for (int v = 0; v < (int)depth->h; ++v)
        for (int u = 0; u < (int)depth->w; ++u) {
            const float z = ((const float*)img_row(depth, (size_t)v))[u];
            const f3 p_c = scale3(unproject1(K, (float)u, (float)v), z);
            const f3 p_w = se3_mul(T, p_c);
            ((float*)img_row(dist, (size_t)v))[u] = trilinear_clamped(vol, p_w, NULL, half);
        }
}

/* ---- cu_depth_tools.cu:15-53, :86-112 ---------------------------------------------------------------------- */
void kfo_disp2depth(const kfo_image* in, const kfo_image* out, float fu, float baseline, float min_disp)
{
    for (size_t y = 0; y < out->h; ++y)
        for (size_t x = 0; x < out->w; ++x) {
            const float d = ((const float*)img_row(in, y))[x];
            ((float*)img_row(out, y))[x] = d >= min_disp ? fu * baseline / d : NAN;
        }
}
void kfo_filter_bad_kinect(const kfo_image* out, const kfo_image* in, int in_is_u16)
{
    for (size_t y = 0; y < out->h; ++y)
        for (size_t x = 0; x < out->w; ++x) {
            const float z_mm = in_is_u16 ? (float)((const uint16_t*)img_row(in, y))[x] : ((const float*)img_row(in, y))[x];
            ((float*)img_row(out, y))[x] = z_mm >= 200 ? z_mm : NAN;
        }
}
void kfo_colour_vbo(const kfo_image* id, const kfo_image* vbo, const kfo_image* rgb, const float T[12])
{
    for (size_t v = 0; v < id->h; ++v)
        for (size_t u = 0; u < id->w; ++u) {
            const f4 Pd = ((const f4*)img_row(vbo, v))[u];
            const float k0 = T_(0, 0) * Pd.x + T_(0, 1) * Pd.y + T_(0, 2) * Pd.z + T_(0, 3) * 1.0f;
            const float k1 = T_(1, 0) * Pd.x + T_(1, 1) * Pd.y + T_(1, 2) * Pd.z + T_(1, 3) * 1.0f;
            const float k2 = T_(2, 0) * Pd.x + T_(2, 1) * Pd.y + T_(2, 2) * Pd.z + T_(2, 3) * 1.0f;
            const float pu = k0 / k2, pv = k1 / k2;
            unsigned char* o = img_row(id, v) + 4 * u;
            if (1.0f <= pu && pu < ((float)rgb->w - 1.0f) && 1.0f <= pv && pv < ((float)rgb->h - 1.0f)) { /* InBounds(x,y,1) */
                const float jx = floorf(pu), jy = floorf(pv);
                const float gx = pu - jx, gy = pv - jy;
                const uc3* cbl = (const uc3*)img_row(rgb, (size_t)jy) + (size_t)jx;
                const uc3* ctl = (const uc3*)img_row(rgb, (size_t)(jy + 1)) + (size_t)jx;
                const f3 c = lerp3(lerp_uc3(cbl[0], cbl[1], gx), lerp_uc3(ctl[0], ctl[1], gx), gy);
                o[0] = (unsigned char)c.x; o[1] = (unsigned char)c.y; o[2] = (unsigned char)c.z; o[3] = 255;
            } else {
                o[0] = o[1] = o[2] = o[3] = 0;
            }
        }
}

/* Joint bilateral filter with a guide image, cu_bilateral.cu:110-143 (guide float or unsigned char) */
void kfo_bilateral_guided(const kfo_image* out, const kfo_image* in, const kfo_image* guide, int guide_is_u8, float gs, float gr, float gc,
                          int size)
{
    const int W = (int)in->w, H = (int)in->h, GW = (int)guide->w, GH = (int)guide->h;
    for (int y = 0; y < (int)out->h; ++y)
        for (int x = 0; x < (int)out->w; ++x) {
            const float p = ((const float*)img_row(in, (size_t)y))[x];
            const float pc = guide_is_u8 ? (float)img_row(guide, (size_t)y)[x] : ((const float*)img_row(guide, (size_t)y))[x];
            float sum = 0, sumw = 0;
            for (int r = -size; r <= size; ++r)
                for (int c = -size; c <= size; ++c) {
                    const float q = ((const float*)img_row(in, (size_t)clampi(y + r, 0, H - 1)))[clampi(x + c, 0, W - 1)];
                    const int gx = clampi(x + c, 0, GW - 1), gy = clampi(y + r, 0, GH - 1);
                    const float qc = guide_is_u8 ? (float)img_row(guide, (size_t)gy)[gx] : ((const float*)img_row(guide, (size_t)gy))[gx];
                    const float rd = p - q, cd = pc - qc;
                    const float sd2 = (float)(r * r + c * c);
                    const float rd2 = rd * rd, cd2 = cd * cd;
                    const float sw = expf(-(sd2) / (2 * gs * gs));
                    const float rw = expf(-(rd2) / (2 * gr * gr));
                    const float cw = expf(-(cd2) / (2 * gc * gc));
                    const float w = sw * rw * cw;
                    sumw += w;
                    sum += w * q;
                }
            ((float*)img_row(out, (size_t)y))[x] = sumw == 0 ? p : sum / sumw;
        }
}

/* TextureDepth, cu_depth_tools.cu:123-207.  kf arrays: K[4], T_iw[12], image per keyframe; phong == NULL selects the
 * single-keyframe kernel.  `color` starts at zero (uninitialised in the reference's multi-keyframe kernel). */
static inline f3 rgb_bilinear(const kfo_image* img, float pu, float pv)
{
    const float jx = floorf(pu), jy = floorf(pv);
    const float gx = pu - jx, gy = pv - jy;
    const uc3* cbl = (const uc3*)img_row(img, (size_t)jy) + (size_t)jx;
    const uc3* ctl = (const uc3*)img_row(img, (size_t)(jy + 1)) + (size_t)jx;
    return lerp3(lerp_uc3(cbl[0], cbl[1], gx), lerp_uc3(ctl[0], ctl[1], gx), gy);
}
void kfo_texture_depth(const kfo_image* out, const kfo_keyframe* kfs, int n_kf, const kfo_image* depth, const kfo_image* norm,
                       const kfo_image* phong, const float T[12], const float K[4])
{
    for (size_t v = 0; v < out->h; ++v)
        for (size_t u = 0; u < out->w; ++u) {
            const float d = ((const float*)img_row(depth, v))[u];
            const f4 N_d = ((const f4*)img_row(norm, v))[u];
            const f3 N_w = so3_mul(T, mk3(N_d.x, N_d.y, N_d.z));
            const f3 P_d = mk3(d * ((float)(int)u - U0) / FU, d * ((float)(int)v - V0) / FV, d);
            const f3 P_w = se3_mul(T, P_d);
            f4 o = {0, 0, 0, 1};
            if (!phong) {
                const kfo_keyframe* k = &kfs[0];
                const f3 P_kf = se3_mul(k->T_iw, P_w);
                const float pu = k->K[2] + k->K[0] * P_kf.x / P_kf.z, pv = k->K[3] + k->K[1] * P_kf.y / P_kf.z;
                const f3 N_c = so3_mul(k->T_iw, N_w);
                const float facing = N_c.x * 0.f + N_c.y * 0.f + N_c.z * 1.f;
                if (2.0f <= pu && pu < ((float)k->img.w - 2.0f) && 2.0f <= pv && pv < ((float)k->img.h - 2.0f) && (double)facing < -0.2) { /* double literal */
                    const f3 c = scale3(rgb_bilinear(&k->img, pu, pv), 1.0f / 255.0f);
                    o.x = c.x; o.y = c.y; o.z = c.z;
                }
            } else {
                float w = 0;
                f3 color = mk3(0, 0, 0);
                for (int i = 0; i < n_kf && kfs[i].img.ptr; ++i) {
                    const kfo_keyframe* k = &kfs[i];
                    const f3 P_kf = se3_mul(k->T_iw, P_w);
                    const float pu = k->K[2] + k->K[0] * P_kf.x / P_kf.z, pv = k->K[3] + k->K[1] * P_kf.y / P_kf.z;
                    const f3 N_c = so3_mul(k->T_iw, N_w);
                    const float ndot = dot3(N_c, P_kf) / -length3(P_kf);
                    if (2.0f <= pu && pu < ((float)k->img.w - 2.0f) && 2.0f <= pv && pv < ((float)k->img.h - 2.0f) && (double)ndot > 0.1 && P_kf.z > 0) {
                        color = add3(color, scale3(rgb_bilinear(&k->img, pu, pv), ndot / 255.0f));
                        w += ndot;
                    }
                }
                if (w == 0) {
                    w = 1;
                    const float ph = ((const float*)img_row(phong, v))[u];
                    color = mk3(ph, ph, ph);
                }
                const f3 c = div3s(color, w);
                o.x = c.x; o.y = c.y; o.z = c.z;
            }
            ((f4*)img_row(out, v))[u] = o;
        }
}
