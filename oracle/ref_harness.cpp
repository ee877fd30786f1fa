// ref_harness.cpp -- drives the REFERENCE'S OWN header code on the CPU.
// TEST INFRASTRUCTURE ONLY (built by `make -C oracle ref` into oracle/_ref/).
//
// What this is: the reference keeps most of the hot path's arithmetic in
// __host__ __device__ header methods: voxel positions,
// SE3 products, pinhole projection, bounds tests, bilinear / trilinear sampling,
// backward-difference gradients, the SDF_t running average, the ROI helpers.
// Those headers are ordinary C++ once <cuda_runtime.h> resolves (the genuine CUDA
// 12.8 headers ship inside this image's triton wheel) and kangaroo/config.h
// exists (generated from the reference's config.h.in by gen_ref_config.cmake).
// This file #includes them where they lie under /root/reference -- nothing is
// copied -- and exposes them through a C ABI that takes the same POD structs as
// oracle/kfx_oracle.h, so tests can demand bit-equality between the plain-C
// restatement and the reference's compiled code.
//
// What this is NOT: the __global__ kernels in src/cu_*.cu need nvcc (absent) and
// are not built.  The per-element drivers below are this repo's own loops; every
// arithmetic step inside them is a call into reference code.  Host fminf/fmaxf
// are the reference's ternary fallbacks (CUDA_SDK/cutil_math.h:55-63): identical
// to device semantics unless an operand is NaN.
#include <kangaroo/BoundedVolume.h>
#include <kangaroo/Image.h>
#include <kangaroo/ImageIntrinsics.h>
#include <kangaroo/ImageKeyframe.h>
#include <kangaroo/InvalidValue.h>
#include <kangaroo/Mat.h>
#include <kangaroo/MatUtils.h>
#include <kangaroo/Sdf.h>
#include <kangaroo/launch_utils.h>
#include <kangaroo/pixel_convert.h>
#include <kangaroo/reweighting.h>
#include <kangaroo/extra/SavePPM.h>
#include <kangaroo/MarchingCubesTables.h>

#include <cstring>

#include "kfx_oracle.h"

// PhongShade (src/cu_raycast.cu:14-28) is a plain __host__ __device__ inline function without thread builtins, but it
// lives in the .cu.  oracle/Makefile extracts exactly those lines from the read-only reference tree into a temporary
// file at build time (REF_PHONG_INC, deleted after the compile; nothing is copied into the repo) and it is compiled
// here, inside the reference's namespace, as the reference wrote it.
namespace roo {
#include REF_PHONG_INC
}

using namespace roo;

typedef Image<float, TargetHost, DontManage> HImgF;
typedef Image<float4, TargetHost, DontManage> HImgF4;
typedef BoundedVolume<SDF_t, TargetHost, DontManage> HVol;

static HImgF imf(const kfo_image* p) { return HImgF((float*)p->ptr, p->w, p->h, p->pitch); }
static HImgF4 imf4(const kfo_image* p) { return HImgF4((float4*)p->ptr, p->w, p->h, p->pitch); }
static HVol mkvol(const kfo_volume* p)
{
    Volume<SDF_t, TargetHost, DontManage> v((SDF_t*)p->ptr, p->w, p->h, p->d, p->pitch, p->img_pitch);
    return HVol(v, BoundingBox(make_float3(p->boxmin[0], p->boxmin[1], p->boxmin[2]),
                               make_float3(p->boxmax[0], p->boxmax[1], p->boxmax[2])));
}
static Mat<float, 3, 4> mkT(const float* t)
{
    Mat<float, 3, 4> T;
    for (int i = 0; i < 12; ++i) T.m[i] = t[i];
    return T;
}
static ImageIntrinsics mkK(const float* k) { return ImageIntrinsics(k[0], k[1], k[2], k[3]); }

// KernBilateralFilter, both forms (cu_bilateral.cu:13-41 and :59-92), on the reference's Image::InBounds and
// GetWithClampedRange.  The reference's __expf is a device intrinsic; expf here (the oracle uses the same libm call, the
// GPU kernels are held to 2e-6 relative).  Arithmetic types as in the kernel: `p - q` in Ti's promoted type, then float.
template <typename Ti>
static void bilateral_t(const kfo_image* pout, const kfo_image* pin, float gs, float gr, int size, Ti minval, int use_minval)
{
    HImgF dOut = imf(pout);
    Image<Ti, TargetHost, DontManage> dIn((Ti*)pin->ptr, pin->w, pin->h, pin->pitch);
    for (unsigned y = 0; y < dOut.h; ++y)
        for (unsigned x = 0; x < dOut.w; ++x) {
            if (!dOut.InBounds(x, y)) continue;
            const Ti p = dIn(x, y);
            float sum = 0;
            float sumw = 0;
            if (!use_minval || p >= minval) {
                for (int r = -size; r <= size; ++r) {
                    for (int c = -size; c <= size; ++c) {
                        const Ti q = dIn.GetWithClampedRange(x + c, y + r);
                        if (!use_minval || q >= minval) {
                            const float sd2 = r * r + c * c;
                            const float id = p - q;
                            const float id2 = id * id;
                            const float sw = expf(-(sd2) / (2 * gs * gs));
                            const float iw = expf(-(id2) / (2 * gr * gr));
                            const float w = sw * iw;
                            sumw += w;
                            sum += w * q;
                        }
                    }
                }
            }
            dOut(x, y) = (float)(sum / sumw);
        }
}
extern "C" {

// layouts the C-ABI mirrors
void ref_sizeof(size_t out[8])
{
    out[0] = sizeof(Image<float>);
    out[1] = sizeof(Volume<SDF_t>);
    out[2] = sizeof(BoundedVolume<SDF_t>);
    out[3] = sizeof(SDF_t);
    out[4] = sizeof(Mat<float, 3, 4>);
    out[5] = sizeof(ImageIntrinsics);
    out[6] = sizeof(BoundingBox);
    out[7] = sizeof(float4);
}

void ref_voxel_position(const kfo_volume* v, int x, int y, int z, float out[3])
{
    const float3 p = mkvol(v).VoxelPositionInUnits(x, y, z);
    out[0] = p.x; out[1] = p.y; out[2] = p.z;
}

void ref_voxel_size(const kfo_volume* v, float out[3])
{
    const float3 p = mkvol(v).VoxelSizeUnits();
    out[0] = p.x; out[1] = p.y; out[2] = p.z;
}

// Per-voxel TSDF integration, every step a reference header call:
// VoxelPositionInUnits, Mat*float3, Project, InBounds, GetBilinear, dot, length,
// clamp, SDF_t::operator+=, LimitWeight.
uint64_t ref_sdf_fuse(const kfo_volume* pv, const kfo_image* pd, const kfo_image* pn, const float* t,
                      const float* k, float trunc_dist, float max_w, float mincostheta, int full_extent)
{
    HVol vol = mkvol(pv);
    HImgF depth = imf(pd);
    HImgF4 normals = imf4(pn);
    const Mat<float, 3, 4> T_cw = mkT(t);
    const ImageIntrinsics K = mkK(k);
    const int X = full_extent ? (int)vol.w : (int)(vol.w / 8) * 8;
    const int Y = full_extent ? (int)vol.h : (int)(vol.h / 8) * 8;
    const int Z = full_extent ? (int)vol.d : (int)(vol.d / 8) * 8;
    uint64_t n = 0;
    for (int z = 0; z < Z; ++z)
        for (int y = 0; y < Y; ++y)
            for (int x = 0; x < X; ++x) {
                const float3 P_c = T_cw * vol.VoxelPositionInUnits(x, y, z);
                const float2 p_c = K.Project(P_c);
                if (!depth.InBounds(p_c, 2)) continue;
                const float md = depth.GetBilinear<float>(p_c);
                const float3 mdn = make_float3(normals.GetBilinear<float4>(p_c));
                const float costheta = dot(mdn, P_c) / -length(P_c);
                const float sd = costheta * (md - P_c.z);
                const float w = costheta * 1.0f / P_c.z;
                if (sd <= -trunc_dist) continue;
                if (std::isfinite(md) && std::isfinite(w) && costheta > mincostheta) {
                    SDF_t s(clamp(sd, -trunc_dist, trunc_dist), w);
                    s += vol(x, y, z);
                    s.LimitWeight(max_w);
                    vol(x, y, z) = s;
                    ++n;
                }
            }
    return n;
}

// Trilinear sample and gradient at a world position (BoundedVolume.h:93-106)
float ref_trilinear(const kfo_volume* pv, const float pos[3])
{
    return mkvol(pv).GetUnitsTrilinearClamped(make_float3(pos[0], pos[1], pos[2]));
}
void ref_backward_diff(const kfo_volume* pv, const float pos[3], float out[3])
{
    const float3 g = mkvol(pv).GetUnitsBackwardDiffDxDyDz(make_float3(pos[0], pos[1], pos[2]));
    out[0] = g.x; out[1] = g.y; out[2] = g.z;
}

// Ray march with the reference's samplers.  Outputs depth (NaN = miss) and the
// camera-frame normal; ref_raycast_shade adds the PhongShade image.
void ref_raycast_geom(const kfo_image* pdepth, const kfo_image* pnorm, const kfo_volume* pv, const float* t,
                      const float* k, float near, float far, float trunc_dist, int subpix)
{
    HVol vol = mkvol(pv);
    HImgF imgdepth = imf(pdepth);
    HImgF4 norm = imf4(pnorm);
    const Mat<float, 3, 4> T_wc = mkT(t);
    const ImageIntrinsics K = mkK(k);
    for (int v = 0; v < (int)imgdepth.h; ++v)
        for (int u = 0; u < (int)imgdepth.w; ++u) {
            const float3 c_w = SE3Translation(T_wc);
            const float3 ray_c = K.Unproject(u, v);
            const float3 ray_w = mulSO3(T_wc, ray_c);
            const float3 ta = (vol.bbox.Min() - c_w) / ray_w;
            const float3 tb = (vol.bbox.Max() - c_w) / ray_w;
            const float3 tmin = fminf(ta, tb);
            const float3 tmax = fmaxf(ta, tb);
            const float max_tmin = fmaxf(fmaxf(fmaxf(tmin.x, tmin.y), tmin.z), near);
            const float min_tmax = fminf(fminf(fminf(tmax.x, tmax.y), tmax.z), far);
            float depth = 0.0f;
            if (max_tmin < min_tmax) {
                float lambda = max_tmin;
                float last_sdf = InvalidValue<float>::Value();
                const float min_step = vol.VoxelSizeUnits().x;
                float step = 0;
                while (lambda < min_tmax) {
                    const float sdf = vol.GetUnitsTrilinearClamped(c_w + lambda * ray_w);
                    if (sdf <= 0) {
                        if (last_sdf > 0) {
                            if (subpix) lambda = lambda + step * sdf / (last_sdf - sdf);
                            depth = lambda;
                        }
                        break;
                    }
                    step = sdf > 0 ? fmaxf(sdf, min_step) : trunc_dist;
                    lambda += step;
                    last_sdf = sdf;
                }
            }
            if (depth > 0) {
                const float3 g = vol.GetUnitsBackwardDiffDxDyDz(c_w + depth * ray_w);
                const float len = length(g);
                const float3 n_w = len > 0 ? g / len : make_float3(0, 0, 1);
                imgdepth(u, v) = depth;
                norm(u, v) = make_float4(mulSO3inv(T_wc, n_w), 1);
            } else {
                imgdepth(u, v) = InvalidValue<float>::Value();
                norm(u, v) = make_float4(0, 0, 0, 0);
            }
        }
}

void ref_depth_to_vbo(const kfo_image* pvbo, const kfo_image* pd, const float* k, float scale)
{
    HImgF4 vbo = imf4(pvbo);
    HImgF d = imf(pd);
    const ImageIntrinsics K = mkK(k);
    for (int v = 0; v < (int)vbo.h; ++v)
        for (int u = 0; u < (int)vbo.w; ++u) {
            const float3 P = K.Unproject(u, v, scale * d(u, v));
            vbo(u, v) = make_float4(P.x, P.y, P.z, 1);
        }
}

void ref_bilateral_f32(const kfo_image* pout, const kfo_image* pin, float gs, float gr, int size, float minval, int use_minval)
{
    bilateral_t<float>(pout, pin, gs, gr, size, minval, use_minval);
}
void ref_bilateral_u16(const kfo_image* pout, const kfo_image* pin, float gs, float gr, int size, unsigned short minval)
{
    bilateral_t<unsigned short>(pout, pin, gs, gr, size, minval, 1);
}
void ref_bilateral_u8(const kfo_image* pout, const kfo_image* pin, float gs, float gr, int size)
{
    bilateral_t<unsigned char>(pout, pin, gs, gr, size, 0, 0);
}

// KernNormalsFromVbo (cu_normals.cu:12-38) on the reference's float4 operator-, make_float3, length, make_float4
void ref_normals_from_vbo(const kfo_image* pn, const kfo_image* pv)
{
    HImgF4 dN = imf4(pn), dV = imf4(pv);
    for (int v = 0; v < (int)dN.h; ++v)
        for (int u = 0; u < (int)dN.w; ++u) {
            if (u + 1 < (int)dN.w && v + 1 < (int)dN.h) {
                const float4 Vc = dV(u, v);
                const float4 Vr = dV(u + 1, v);
                const float4 Vu = dV(u, v + 1);
                const float4 a = Vr - Vc;
                const float4 b = Vu - Vc;
                const float3 axb = make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
                const float magaxb = length(axb);
                dN(u, v) = make_float4(-axb.x / magaxb, -axb.y / magaxb, -axb.z / magaxb, 1);
            } else {
                dN(u, v) = make_float4(0, 0, 0, 0);
            }
        }
}

// The reference's PhongShade, compiled from its own lines (see the top of this file)
float ref_phong_shade(const float p_c[3], const float n_c[3])
{
    return PhongShade(make_float3(p_c[0], p_c[1], p_c[2]), make_float3(n_c[0], n_c[1], n_c[2]));
}

// The shade image of KernRaycastSdf (cu_raycast.cu:90-101) from the depth / normal images ref_raycast_geom produced:
// p_c = depth * K.Unproject(u, v), img = PhongShade(p_c, n_c) on hits, 0 on misses.
void ref_raycast_shade(const kfo_image* pimg, const kfo_image* pdepth, const kfo_image* pnorm, const float* k)
{
    HImgF img = imf(pimg), imgdepth = imf(pdepth);
    HImgF4 norm = imf4(pnorm);
    const ImageIntrinsics K = mkK(k);
    for (int v = 0; v < (int)img.h; ++v)
        for (int u = 0; u < (int)img.w; ++u) {
            const float depth = imgdepth(u, v);
            if (depth > 0) {
                const float3 ray_c = K.Unproject(u, v);
                const float3 p_c = depth * ray_c;
                img(u, v) = PhongShade(p_c, make_float3(norm(u, v)));
            } else {
                img(u, v) = 0;
            }
        }
}

// KernSdfSphere (cu_sdffusion.cu:175-187) over the (dim/8)*8 extents of its launch (:189-195)
void ref_sdf_sphere(const kfo_volume* pv, const float center[3], float r)
{
    HVol vol = mkvol(pv);
    const float3 c = make_float3(center[0], center[1], center[2]);
    const int X = (int)(vol.w / 8) * 8, Y = (int)(vol.h / 8) * 8, Z = (int)(vol.d / 8) * 8;
    for (int z = 0; z < Z; ++z)
        for (int y = 0; y < Y; ++y)
            for (int x = 0; x < X; ++x) {
                const float3 pos = vol.VoxelPositionInUnits(x, y, z);
                const float dist = length(pos - c);
                const float sdf = dist - r;
                vol(x, y, z) = SDF_t(sdf);
            }
}

// KernElementwiseScaleBias<float,float,float> (cu_operations.cu:39-49) on the reference's InBounds and ConvertPixel
void ref_elementwise_scale_bias_f32(const kfo_image* pb, const kfo_image* pa, float s, float offset)
{
    HImgF b = imf(pb), a = imf(pa);
    for (int y = 0; y < (int)b.h; ++y)
        for (int x = 0; x < (int)b.w; ++x)
            if (b.InBounds(x, y)) {
                const float v1 = ConvertPixel<float, float>(a(x, y));
                b(x, y) = ConvertPixel<float, float>(s * v1 + offset);
            }
}

// KernBoxHalfIgnoreInvalid<float,float,float> (cu_resample.cu:89-110) on the reference's InvalidValue<float>
void ref_box_half_ignore_invalid_f32(const kfo_image* pout, const kfo_image* pin)
{
    HImgF out = imf(pout), in = imf(pin);
    for (unsigned y = 0; y < out.h; ++y)
        for (unsigned x = 0; x < out.w; ++x) {
            const float* tl = &in(2 * x, 2 * y);
            const float* bl = &in(2 * x, 2 * y + 1);
            const float v1 = *tl;
            const float v2 = *(tl + 1);
            const float v3 = *bl;
            const float v4 = *(bl + 1);
            int n = 0;
            float sum = 0;
            if (InvalidValue<float>::IsValid(v1)) { sum += v1; n++; }
            if (InvalidValue<float>::IsValid(v2)) { sum += v2; n++; }
            if (InvalidValue<float>::IsValid(v3)) { sum += v3; n++; }
            if (InvalidValue<float>::IsValid(v4)) { sum += v4; n++; }
            out(x, y) = n > 0 ? (float)(sum / n) : InvalidValue<float>::Value();
        }
}

void ref_fit_to_frustum(float lo[3], float hi[3], const float* t, float w, float h, const float* k,
                        float near, float far)
{
    BoundingBox bb(mkT(t), w, h, mkK(k), near, far);
    lo[0] = bb.Min().x; lo[1] = bb.Min().y; lo[2] = bb.Min().z;
    hi[0] = bb.Max().x; hi[1] = bb.Max().y; hi[2] = bb.Max().z;
}

void ref_sub_bounding_volume(kfo_volume* out, const kfo_volume* pv, const float rmin[3], const float rmax[3])
{
    HVol vol = mkvol(pv);
    HVol s = vol.SubBoundingVolume(BoundingBox(make_float3(rmin[0], rmin[1], rmin[2]),
                                               make_float3(rmax[0], rmax[1], rmax[2])));
    out->pitch = s.pitch; out->ptr = s.ptr; out->w = s.w; out->h = s.h;
    out->img_pitch = s.img_pitch; out->d = s.d;
    out->boxmin[0] = s.bbox.Min().x; out->boxmin[1] = s.bbox.Min().y; out->boxmin[2] = s.bbox.Min().z;
    out->boxmax[0] = s.bbox.Max().x; out->boxmax[1] = s.bbox.Max().y; out->boxmax[2] = s.bbox.Max().z;
}

void ref_se3_inverse(float o[12], const float* t)
{
    const Mat<float, 3, 4> r = SE3inv(mkT(t));
    for (int i = 0; i < 12; ++i) o[i] = r.m[i];
}

void ref_intrinsics_level(float o[4], const float* k, int level)
{
    const ImageIntrinsics r = mkK(k)[level];
    o[0] = r.fu; o[1] = r.fv; o[2] = r.u0; o[3] = r.v0;
}

void ref_sdf_accumulate(float val, float w, float old_val, float old_w, float max_w, float out[2])
{
    SDF_t s(val, w);
    s += SDF_t(old_val, old_w);
    s.LimitWeight(max_w);
    out[0] = s.val; out[1] = s.w;
}

// Projective point-to-plane ICP: the body of KernPoseRefinementProjectiveIcpPointPlane
// (cu_model_refinement.cu:541-593) with every arithmetic step a call into the reference's headers
// (Mat*float4, dn, InBounds, GetNearestNeighbour, float3-float4, dot, SE3gen*mul, LSReweightTukey,
// OuterProduct, mul_aTb) and the sums taken with LeastSquaresSystem::operator+= in ReducePutBlock's order
// (LeastSquareSum.h:71-85; that header itself needs thrust and is not included).  Block geometry as
// InitDimFromOutputImage(dPl,16,16); blocks are then summed in the order documented in kfx_oracle.c.
void ref_icp_point_plane(const kfo_image* pPl, const kfo_image* pPr, const kfo_image* pNr, const float* kt, const float* trl,
                         float c, const kfo_image* pdbg, kfo_lss6* out, kfo_lss6* block_sums)
{
    typedef LeastSquaresSystem<float, 6> LSS;
    static_assert(sizeof(LSS) == sizeof(kfo_lss6), "LeastSquaresSystem<float,6> layout");
    HImgF4 dPl = imf4(pPl), dPr = imf4(pPr), dNr = imf4(pNr);
    HImgF4 dDebug = pdbg ? imf4(pdbg) : HImgF4();
    const Mat<float, 3, 4> KT_lr = mkT(kt), T_rl = mkT(trl);
    const unsigned bx = Gcd<unsigned>(dPl.w, 16), by = Gcd<unsigned>(dPl.h, 16);
    const unsigned gx = dPl.w / bx, gy = dPl.h / by, n = bx * by;
    LSS partial[256];
    for (int i = 0; i < 256; ++i) partial[i].SetZero();
    for (unsigned bj = 0; bj < gy; ++bj)
        for (unsigned bi = 0; bi < gx; ++bi) {
            LSS sReduce[256];
            for (unsigned ty = 0; ty < by; ++ty)
                for (unsigned tx = 0; tx < bx; ++tx) {
                    const unsigned u = bi * bx + tx, v = bj * by + ty;
                    LSS& sum = sReduce[ty * bx + tx];
                    sum.SetZero();
                    const float4 Pr = dPr(u, v);
                    const float4 Nr = dNr(u, v);
                    const float3 KPl = KT_lr * Pr;
                    const float2 pl = dn(KPl);
                    float4 dbg;
                    if (std::isfinite(Pr.z) && Nr.w == 1.0f && dPl.InBounds(pl, 3)) {
                        const float4 _Pl = dPl.GetNearestNeighbour(pl);
                        if (std::isfinite(_Pl.z)) {
                            const float3 _Pr = T_rl * _Pl;
                            const float3 Dr = _Pr - Pr;
                            const float DrDotNr = dot(Dr, Nr);
                            const float y = DrDotNr;
                            const Mat<float, 1, 6> Jr = {
                                -dot(SE3gen0mul(_Pr), Nr), -dot(SE3gen1mul(_Pr), Nr), -dot(SE3gen2mul(_Pr), Nr),
                                -dot(SE3gen3mul(_Pr), Nr), -dot(SE3gen4mul(_Pr), Nr), -dot(SE3gen5mul(_Pr), Nr)};
                            const float w = (1.0f / Pr.z) * LSReweightTukey(y, c);
                            sum.JTJ = OuterProduct(Jr, w);
                            sum.JTy = mul_aTb(Jr, y * w);
                            sum.obs = 1;
                            sum.sqErr = y * y;
                            const float db = fabs(y);
                            dbg = make_float4(db, db, db, 1);
                        } else {
                            dbg = make_float4(0, 0, 1, 1);
                        }
                    } else {
                        dbg = make_float4(1, 0, 0, 1);
                    }
                    if (pdbg) dDebug(u, v) = dbg;
                }
            for (unsigned S = n / 2; S > 0; S >>= 1)
                for (unsigned tid = 0; tid < S; ++tid) sReduce[tid] += sReduce[tid + S];
            const unsigned bid = bj * gx + bi;
            if (block_sums) memcpy(&block_sums[bid], &sReduce[0], sizeof(LSS));
            partial[bid % 256] += sReduce[0];
        }
    for (unsigned S = 128; S > 0; S >>= 1)
        for (unsigned tid = 0; tid < S; ++tid) partial[tid] += partial[tid + S];
    memcpy(out, &partial[0], sizeof(LSS));
}

// Colour fusion: the body of the colour KernSdfFuse (cu_sdffusion.cu:70-118), every arithmetic step a
// reference header call (Image<uchar3>::GetBilinear<float3> -> sampling.h lerp, ConvertPixel<float,float3>,
// the double "/ 255.0", SDF_t::operator+=, LimitWeight).  Extents of the (16,16) launch with its z loop.
uint64_t ref_sdf_fuse_color(const kfo_volume* pv, const kfo_volume* pc, const kfo_image* pd, const kfo_image* pn, const float* t,
                            const float* k, const kfo_image* pimg, const float* tiw, const float* kimg, float trunc_dist,
                            float max_w, float mincostheta, int full_extent)
{
    HVol vol = mkvol(pv);
    Volume<float, TargetHost, DontManage> cv0((float*)pc->ptr, pc->w, pc->h, pc->d, pc->pitch, pc->img_pitch);
    BoundedVolume<float, TargetHost, DontManage> colorVol(cv0, BoundingBox(make_float3(pc->boxmin[0], pc->boxmin[1], pc->boxmin[2]),
                                                                        make_float3(pc->boxmax[0], pc->boxmax[1], pc->boxmax[2])));
    HImgF depth = imf(pd);
    HImgF4 normals = imf4(pn);
    Image<uchar3, TargetHost, DontManage> img((uchar3*)pimg->ptr, pimg->w, pimg->h, pimg->pitch);
    const Mat<float, 3, 4> T_cw = mkT(t), T_iw = mkT(tiw);
    const ImageIntrinsics K = mkK(k), Kimg = mkK(kimg);
    const int X = full_extent ? (int)vol.w : (int)(vol.w / 16) * 16;
    const int Y = full_extent ? (int)vol.h : (int)(vol.h / 16) * 16;
    uint64_t n = 0;
    for (int z = 0; z < (int)vol.d; ++z)
        for (int y = 0; y < Y; ++y)
            for (int x = 0; x < X; ++x) {
                const float3 P_w = vol.VoxelPositionInUnits(x, y, z);
                const float3 P_c = T_cw * P_w;
                const float2 p_c = K.Project(P_c);
                const float3 P_i = T_iw * P_w;
                const float2 p_i = Kimg.Project(P_i);
                if (depth.InBounds(p_c, 2) && img.InBounds(p_i, 2)) {
                    const float vd = P_c.z;
                    const float md = depth.GetBilinear<float>(p_c);
                    const float3 mdn = make_float3(normals.GetBilinear<float4>(p_c));
                    const float c = ConvertPixel<float, float3>(img.GetBilinear<float3>(p_i)) / 255.0;
                    const float costheta = dot(mdn, P_c) / -length(P_c);
                    const float sd = costheta * (md - vd);
                    const float w = costheta * 1.0f / vd;
                    if (sd <= -trunc_dist) {
                    } else {
                        if (std::isfinite(md) && std::isfinite(w) && costheta > mincostheta) {
                            const SDF_t curvol = vol(x, y, z);
                            SDF_t sdf(clamp(sd, -trunc_dist, trunc_dist), w);
                            sdf += curvol;
                            sdf.LimitWeight(max_w);
                            vol(x, y, z) = sdf;
                            colorVol(x, y, z) = (w * c + colorVol(x, y, z) * curvol.w) / (w + curvol.w);
                            ++n;
                        }
                    }
                }
            }
    return n;
}

// BoundedVolume<float>::GetUnitsTrilinearClamped, the colour sample of the colour raycast (cu_raycast.cu:172)
float ref_color_trilinear(const kfo_volume* pc, const float pos[3])
{
    Volume<float, TargetHost, DontManage> cv0((float*)pc->ptr, pc->w, pc->h, pc->d, pc->pitch, pc->img_pitch);
    BoundedVolume<float, TargetHost, DontManage> colorVol(cv0, BoundingBox(make_float3(pc->boxmin[0], pc->boxmin[1], pc->boxmin[2]),
                                                                        make_float3(pc->boxmax[0], pc->boxmax[1], pc->boxmax[2])));
    return colorVol.GetUnitsTrilinearClamped(make_float3(pos[0], pos[1], pos[2]));
}

// Volume persistence: the reference's own host writer SavePXM(std::ofstream&, Volume<T,TargetHost,Manage>&)
// (extra/SavePPM.h:46-58) behind the two bbox lines its BoundedVolume overload emits first (:79-87; that
// overload itself takes a device volume and copies it with the CUDA runtime, so it cannot run here).
int ref_save_pxm(const char* path, const kfo_volume* pv, int elem_bytes)
{
    std::ofstream bFile(path, std::ios::out | std::ios::binary);
    const BoundingBox bbox(make_float3(pv->boxmin[0], pv->boxmin[1], pv->boxmin[2]), make_float3(pv->boxmax[0], pv->boxmax[1], pv->boxmax[2]));
    bFile << bbox.boxmin.x << " " << bbox.boxmin.y << " " << bbox.boxmin.z << std::endl;
    bFile << bbox.boxmax.x << " " << bbox.boxmax.y << " " << bbox.boxmax.z << std::endl;
    if (elem_bytes == 8) {
        Volume<SDF_t, TargetHost, DontManage> v((SDF_t*)pv->ptr, pv->w, pv->h, pv->d, pv->pitch, pv->img_pitch);
        SavePXM<SDF_t, DontManage>(bFile, v);
    } else if (elem_bytes == 4) {
        Volume<float, TargetHost, DontManage> v((float*)pv->ptr, pv->w, pv->h, pv->d, pv->pitch, pv->img_pitch);
        SavePXM<float, DontManage>(bFile, v);
    } else {
        return -1;
    }
    return 0;
}

// The reference's marching-cubes case tables (MarchingCubesTables.h:61-348), exposed so that tests can compare
// the topology of this repo's independently derived tables with them case by case.
void ref_mc_tables(int edge_flags[256], int tris[256 * 16])
{
    for (int i = 0; i < 256; ++i) {
        edge_flags[i] = aiCubeEdgeFlags[i];
        for (int j = 0; j < 16; ++j) tris[i * 16 + j] = a2iTriangleConnectionTable[i][j];
    }
}

// SdfDistance (cu_sdffusion.cu:200-217) and the analytic renderers' per-pixel arithmetic (cu_raycast.cu:202-310), every
// step a reference header call (Unproject, float*float3, Mat*float3, GetUnitsTrilinearClamped, mulSO3, mulSE3inv,
// Plane_b_from_a, fminf/fmaxf(float3), dot, length); PhongShade lives in the .cu and is not covered.
void ref_sdf_distance(const kfo_image* pdist, const kfo_image* pdepth, const kfo_volume* pv, const float* t, const float* k)
{
    HImgF dist = imf(pdist), depth = imf(pdepth);
    HVol vol = mkvol(pv);
    const Mat<float, 3, 4> T_wc = mkT(t);
    const ImageIntrinsics K = mkK(k);
    for (int v = 0; v < (int)depth.h; ++v)
        for (int u = 0; u < (int)depth.w; ++u) {
            const float z = depth(u, v);
            const float3 p_c = z * K.Unproject(u, v);
            const float3 p_w = T_wc * p_c;
            dist(u, v) = vol.GetUnitsTrilinearClamped(p_w);
        }
}
void ref_analytic_depths(const kfo_image* pbox, const kfo_image* psph, const kfo_image* ppl, const float* t, const float* k,
                         const float* bmin, const float* bmax, const float* center, float r, const float* n_w)
{
    HImgF ibox = imf(pbox), isph = imf(psph), ipl = imf(ppl);
    const Mat<float, 3, 4> T_wc = mkT(t);
    const ImageIntrinsics K = mkK(k);
    const BoundingBox bbox(make_float3(bmin[0], bmin[1], bmin[2]), make_float3(bmax[0], bmax[1], bmax[2]));
    const float3 center_c = mulSE3inv(T_wc, make_float3(center[0], center[1], center[2]));
    const float3 n_c = Plane_b_from_a(T_wc, make_float3(n_w[0], n_w[1], n_w[2]));
    for (int v = 0; v < (int)ibox.h; ++v)
        for (int u = 0; u < (int)ibox.w; ++u) {
            const float3 c_w = SE3Translation(T_wc);
            const float3 ray_c = K.Unproject(u, v);
            const float3 ray_w = mulSO3(T_wc, ray_c);
            const float3 tminbound = (bbox.Min() - c_w) / ray_w;
            const float3 tmaxbound = (bbox.Max() - c_w) / ray_w;
            const float3 tmin = fminf(tminbound, tmaxbound);
            const float3 tmax = fmaxf(tminbound, tmaxbound);
            const float max_tmin = fmaxf(fmaxf(tmin.x, tmin.y), tmin.z);
            const float min_tmax = fminf(fminf(tmax.x, tmax.y), tmax.z);
            ibox(u, v) = (max_tmin < min_tmax) ? max_tmin : InvalidValue<float>::Value();
            const float ldotc = dot(ray_c, center_c);
            const float lsq = dot(ray_c, ray_c);
            const float csq = dot(center_c, center_c);
            isph(u, v) = (ldotc - sqrtf(ldotc * ldotc - lsq * (csq - r * r))) / lsq;
            ipl(u, v) = -1 / dot(n_c, ray_c);
        }
}

// ColourVbo's per-pixel body (cu_depth_tools.cu:91-110) with the reference's Mat product, InBounds, GetBilinear<float3>
// and the float -> unsigned char narrowing of make_uchar4
void ref_colour_vbo(const kfo_image* pid, const kfo_image* pvbo, const kfo_image* prgb, const float* kt)
{
    Image<uchar4, TargetHost, DontManage> dId((uchar4*)pid->ptr, pid->w, pid->h, pid->pitch);
    HImgF4 dPd = imf4(pvbo);
    Image<uchar3, TargetHost, DontManage> dIc((uchar3*)prgb->ptr, prgb->w, prgb->h, prgb->pitch);
    const Mat<float, 3, 4> KT_cd = mkT(kt);
    for (int v = 0; v < (int)dId.h; ++v)
        for (int u = 0; u < (int)dId.w; ++u) {
            const float4 Pd4 = dPd(u, v);
            const Mat<float, 4, 1> Pd = {Pd4.x, Pd4.y, Pd4.z, 1};
            const Mat<float, 3, 1> KPc = KT_cd * Pd;
            const Mat<float, 2, 1> pc = {KPc(0) / KPc(2), KPc(1) / KPc(2)};
            uchar4 Id;
            if (dIc.InBounds(pc(0), pc(1), 1)) {
                const float3 c = dIc.GetBilinear<float3>(pc(0), pc(1));
                Id = make_uchar4(c.x, c.y, c.z, 255);
            } else {
                Id = make_uchar4(0, 0, 0, 0);
            }
            dId(u, v) = Id;
        }
}

// TextureDepth, single-keyframe kernel body (cu_depth_tools.cu:131-148) and the per-keyframe terms of the blended kernel
// (:178-191) with the reference's ImageKeyframe / ImageTransformProject / Unproject / mulSO3 / GetBilinear<float3>.
// Blended form: color starts at zero here (uninitialised in the reference).
void ref_texture_depth(const kfo_image* pout, const kfo_keyframe* kfs, int n_kf, const kfo_image* pdepth, const kfo_image* pnorm,
                       const kfo_image* pphong, const float* t, const float* k)
{
    HImgF4 img = imf4(pout), norm = imf4(pnorm);
    HImgF depth = imf(pdepth);
    const Mat<float, 3, 4> T_wd = mkT(t);
    const ImageIntrinsics Kdepth = mkK(k);
    ImageKeyframe<uchar3> kf[10];
    for (int i = 0; i < 10; ++i) {
        kf[i].img.ptr = 0;
        if (i < n_kf && kfs[i].img.ptr) {
            kf[i].K = mkK(kfs[i].K);
            kf[i].T_iw = mkT(kfs[i].T_iw);
            kf[i].img.ptr = (uchar3*)kfs[i].img.ptr;
            kf[i].img.pitch = kfs[i].img.pitch;
            kf[i].img.w = kfs[i].img.w;
            kf[i].img.h = kfs[i].img.h;
        }
    }
    for (int v = 0; v < (int)img.h; ++v)
        for (int u = 0; u < (int)img.w; ++u) {
            const float d = depth(u, v);
            const float4 N_d = norm(u, v);
            const float3 N_w = mulSO3(T_wd, N_d);
            const float3 P_d = Kdepth.Unproject(u, v, d);
            const float3 P_w = T_wd * P_d;
            if (!pphong) {
                const float2 p_kf = kf[0].Project(P_w);
                const float3 N_c = mulSO3(kf[0].T_iw, N_w);
                if (kf[0].img.InBounds(p_kf, 2) && dot(N_c, make_float3(0, 0, 1)) < -0.2) {
                    const float3 color = (1.0f / 255.0f) * kf[0].img.GetBilinear<float3>(p_kf);
                    img(u, v) = make_float4(color, 1);
                } else {
                    img(u, v) = make_float4(0, 0, 0, 1);
                }
            } else {
                HImgF phong = imf(pphong);
                float w = 0;
                float3 color = make_float3(0, 0, 0);
                for (int i = 0; i < 10 && kf[i].img.ptr; ++i) {
                    const float3 P_kf = kf[i].T_iw * P_w;
                    const float2 p_kf = kf[i].K.Project(P_kf);
                    const float3 N_c = mulSO3(kf[i].T_iw, N_w);
                    const float ndot = dot(N_c, P_kf) / -length(P_kf);
                    if (kf[i].img.InBounds(p_kf, 2) && ndot > 0.1 && P_kf.z > 0) {
                        color += (ndot / 255.0f) * kf[i].img.GetBilinear<float3>(p_kf);
                        w += ndot;
                    }
                }
                if (w == 0) {
                    w = 1;
                    color = make_float3(phong(u, v));
                }
                img(u, v) = make_float4(color / w, 1);
            }
        }
}

}
