// analytic.hip -- the remaining entry points of the reference's cu_raycast.h / cu_sdffusion.h: the analytic
// renderers RaycastBox / RaycastSphere / RaycastPlane (src/cu_raycast.cu:202-310; used by the reference's examples
// to produce synthetic depth) and SdfDistance (src/cu_sdffusion.cu:200-225: the TSDF value at every pixel's
// back-projected depth).  One pixel per lane, 64 x 4 pixel workgroups, IEEE arithmetic in the reference's order.
#include "kfx_device.h"
#include "sampling.h"

namespace kfx {

struct PixParams {
    unsigned char* dptr;   // Image<float> depth (in/out)
    unsigned char* iptr;   // Image<float> shade (may be null)
    size_t dpitch, ipitch;
    int w, h;
    Pose T;
    Intr K;
    V3 a, b;               // box min / max, or sphere centre (camera frame) / plane normal (camera frame)
    float r;
};

// PhongShade (cu_raycast.cu:14-28)
__device__ __forceinline__ float phong_shade(const V3 p_c, const V3 n_c)
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

__device__ __forceinline__ bool pixel(const PixParams& p, int& u, int& v)
{
    u = blockIdx.x * 64 + (threadIdx.x & 63);
    v = blockIdx.y * 4 + (threadIdx.x >> 6);
    return u < p.w && v < p.h;
}
__device__ __forceinline__ V3 unproject(const Intr& K, int u, int v) { return v3(((float)u - K.u0) / K.fu, ((float)v - K.v0) / K.fv, 1.0f); }

__global__ __launch_bounds__(256) void k_raycast_box(const PixParams p)
{
    int u, v;
    if (!pixel(p, u, v)) return;
    const V3 c_w = v3(p.T.m[3], p.T.m[7], p.T.m[11]);
    const V3 ray_w = so3_mul(p.T, unproject(p.K, u, v));
    const V3 ta = div_cw(p.a - c_w, ray_w), tb = div_cw(p.b - c_w, ray_w);
    const float max_tmin = fmaxf(fmaxf(fminf(ta.x, tb.x), fminf(ta.y, tb.y)), fminf(ta.z, tb.z));
    const float min_tmax = fminf(fminf(fmaxf(ta.x, tb.x), fmaxf(ta.y, tb.y)), fmaxf(ta.z, tb.z));
    *(reinterpret_cast<float*>(p.dptr + (size_t)v * p.dpitch) + u) = (max_tmin < min_tmax) ? max_tmin : __builtin_nanf("");
}

__global__ __launch_bounds__(256) void k_raycast_sphere(const PixParams p)
{
    int u, v;
    if (!pixel(p, u, v)) return;
    const V3 ray_c = unproject(p.K, u, v);
    const V3 center_c = p.a;
    const float ldotc = dot(ray_c, center_c);
    const float lsq = dot(ray_c, ray_c);
    const float csq = dot(center_c, center_c);
    const float depth = (ldotc - sqrtf(ldotc * ldotc - lsq * (csq - p.r * p.r))) / lsq;
    float* pd = reinterpret_cast<float*>(p.dptr + (size_t)v * p.dpitch) + u;
    const float prev = *pd;
    if (depth > 0 && (depth < prev || !isfinite(prev))) {
        *pd = depth;
        if (p.iptr) {
            const V3 p_c = ray_c * depth;
            const V3 n_c = p_c - center_c;
            *(reinterpret_cast<float*>(p.iptr + (size_t)v * p.ipitch) + u) = phong_shade(p_c, div_s(n_c, length(n_c)));
        }
    }
}

__global__ __launch_bounds__(256) void k_raycast_plane(const PixParams p)
{
    int u, v;
    if (!pixel(p, u, v)) return;
    const V3 ray_c = unproject(p.K, u, v);
    const V3 n_c = p.a;
    const float depth = -1 / dot(n_c, ray_c);
    float* pd = reinterpret_cast<float*>(p.dptr + (size_t)v * p.dpitch) + u;
    const float prev = *pd;
    if (depth > 0 && (depth < prev || !isfinite(prev))) {
        if (p.iptr) *(reinterpret_cast<float*>(p.iptr + (size_t)v * p.ipitch) + u) = phong_shade(ray_c * depth, div_s(n_c, length(n_c)));
        *pd = depth;
    }
}

struct DistParams {
    VolView vol;
    V3 size, dims1, hi2;
    V3 inv_size;
    int fastdiv, off32;
    unsigned char *optr, *dptr;
    size_t opitch, dpitch;
    int w, h;
    Pose T;
    Intr K;
};

__global__ __launch_bounds__(256) void k_sdf_distance(const DistParams p)
{
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= p.w || v >= p.h) return;
    const float z = *(reinterpret_cast<const float*>(p.dptr + (size_t)v * p.dpitch) + u);
    const V3 p_c = unproject(p.K, u, v) * z;
    const V3 p_w = se3_mul(p.T, p_c);
    *(reinterpret_cast<float*>(p.optr + (size_t)v * p.opitch) + u) = trilinear<RayF32>(p, p_w);
}

static int pix_params(PixParams& p, const kfx_image* imgd, const kfx_image* img, const float T_wc[12], const float K[4], const char* what)
{
    if (!imgd || !imgd->ptr || !T_wc || !K) return set_error(KFX_E_NULL, what);
    if (imgd->pitch < imgd->w * 4 || (((uintptr_t)imgd->ptr | imgd->pitch) & 3)) return set_error(KFX_E_SHAPE, what);
    p.dptr = (unsigned char*)imgd->ptr;
    p.dpitch = imgd->pitch;
    p.w = (int)imgd->w;
    p.h = (int)imgd->h;
    p.iptr = nullptr;
    p.ipitch = 0;
    if (img && img->ptr) {
        if (img->w < imgd->w || img->h < imgd->h || img->pitch < imgd->w * 4 || (((uintptr_t)img->ptr | img->pitch) & 3)) return set_error(KFX_E_SHAPE, what);
        p.iptr = (unsigned char*)img->ptr;
        p.ipitch = img->pitch;
    }
    for (int i = 0; i < 12; ++i) p.T.m[i] = T_wc[i];
    p.K = Intr{K[0], K[1], K[2], K[3]};
    p.a = p.b = V3{0.f, 0.f, 0.f};
    p.r = 0.f;
    return 0;
}

} // namespace kfx

using namespace kfx;

// RaycastBox(imgd, T_wc, K, bbox) (cu_raycast.cu:202-240): entry depth of every pixel's ray into the box, NaN on a miss
extern "C" int kfx_raycast_box(const kfx_image* imgd, const float T_wc[12], const float K[4], const float boxmin[3], const float boxmax[3],
                               kfx_stream stream)
{
    PixParams p;
    if (!boxmin || !boxmax) return set_error(KFX_E_NULL, "RaycastBox: null argument");
    if (int e = pix_params(p, imgd, nullptr, T_wc, K, "RaycastBox: image")) return e;
    if (p.w == 0 || p.h == 0) return 0;
    p.a = V3{boxmin[0], boxmin[1], boxmin[2]};
    p.b = V3{boxmax[0], boxmax[1], boxmax[2]};
    hipLaunchKernelGGL(k_raycast_box, dim3(ceil_div(p.w, 64), ceil_div(p.h, 4)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_raycast_box");
}

// RaycastSphere(imgd, img, T_wc, K, center, r) (cu_raycast.cu:246-279): nearer-than-existing sphere hits overwrite
// imgd (and img, when given, with the Phong shade); center_c = mulSE3inv(T_wc, center) on the host as the reference does
extern "C" int kfx_raycast_sphere(const kfx_image* imgd, const kfx_image* img, const float T_wc[12], const float K[4], const float center[3],
                                  float r, kfx_stream stream)
{
    PixParams p;
    if (!center) return set_error(KFX_E_NULL, "RaycastSphere: null argument");
    if (int e = pix_params(p, imgd, img, T_wc, K, "RaycastSphere: image")) return e;
    if (p.w == 0 || p.h == 0) return 0;
    const float* T = T_wc;
    const float ax = center[0] - T[3], ay = center[1] - T[7], az = center[2] - T[11]; // MatUtils.h:192-200
    p.a = V3{T[0] * ax + T[4] * ay + T[8] * az, T[1] * ax + T[5] * ay + T[9] * az, T[2] * ax + T[6] * ay + T[10] * az};
    p.r = r;
    hipLaunchKernelGGL(k_raycast_sphere, dim3(ceil_div(p.w, 64), ceil_div(p.h, 4)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_raycast_sphere");
}

// RaycastPlane(imgd, img, T_wc, K, n_w) (cu_raycast.cu:285-310): plane n.x = -1; n_c = Plane_b_from_a(T_wc, n_w)
// (MatUtils.h:474-488) on the host; the reference bounds the launch by img and always writes it, so img is required
extern "C" int kfx_raycast_plane(const kfx_image* imgd, const kfx_image* img, const float T_wc[12], const float K[4], const float n_w[3],
                                 kfx_stream stream)
{
    PixParams p;
    if (!n_w || !img || !img->ptr) return set_error(KFX_E_NULL, "RaycastPlane: null argument");
    if (int e = pix_params(p, imgd, img, T_wc, K, "RaycastPlane: image")) return e;
    if (p.w == 0 || p.h == 0) return 0;
    const float* T = T_wc;
    const float dn = T[3] * n_w[0] + T[7] * n_w[1] + T[11] * n_w[2] + 1.0f;
    p.a = V3{(T[0] * n_w[0] + T[4] * n_w[1] + T[8] * n_w[2]) / dn, (T[1] * n_w[0] + T[5] * n_w[1] + T[9] * n_w[2]) / dn,
             (T[2] * n_w[0] + T[6] * n_w[1] + T[10] * n_w[2]) / dn};
    hipLaunchKernelGGL(k_raycast_plane, dim3(ceil_div(p.w, 64), ceil_div(p.h, 4)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_raycast_plane");
}

// SdfDistance(dist, depth, vol, T_wc, K, trunc_distance) (cu_sdffusion.cu:200-225): dist(u,v) = trilinear TSDF at
// T_wc * (depth(u,v) * Unproject(u,v)); trunc_distance is unused by the reference kernel as well
extern "C" int kfx_sdf_distance(const kfx_image* dist, const kfx_image* depth, const kfx_volume* vol, const float T_wc[12], const float K[4],
                                float trunc_distance, kfx_stream stream)
{
    (void)trunc_distance;
    if (!dist || !depth || !vol || !dist->ptr || !depth->ptr || !vol->ptr || !T_wc || !K) return set_error(KFX_E_NULL, "SdfDistance: null argument");
    if (depth->w == 0 || depth->h == 0) return 0;
    if (dist->w < depth->w || dist->h < depth->h || dist->pitch < depth->w * 4 || depth->pitch < depth->w * 4)
        return set_error(KFX_E_SHAPE, "SdfDistance: image sizes");
    if (vol->w < 2 || vol->h < 2 || vol->d < 2 || vol->pitch < vol->w * 8 || vol->img_pitch < vol->pitch * (vol->h - 1) + vol->w * 8)
        return set_error(KFX_E_SHAPE, "SdfDistance: volume");
    if ((((uintptr_t)dist->ptr | dist->pitch | (uintptr_t)depth->ptr | depth->pitch) & 3) || (((uintptr_t)vol->ptr | vol->pitch | vol->img_pitch) & 7))
        return set_error(KFX_E_ALIGN, "SdfDistance: alignment");
    DistParams p;
    set_geometry(p, vol);
    p.optr = (unsigned char*)dist->ptr; p.opitch = dist->pitch;
    p.dptr = (unsigned char*)depth->ptr; p.dpitch = depth->pitch;
    p.w = (int)depth->w; p.h = (int)depth->h;
    for (int i = 0; i < 12; ++i) p.T.m[i] = T_wc[i];
    p.K = Intr{K[0], K[1], K[2], K[3]};
    hipLaunchKernelGGL(k_sdf_distance, dim3(ceil_div(p.w, 64), ceil_div(p.h, 4)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("kfx_sdf_distance");
}
