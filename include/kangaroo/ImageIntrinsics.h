// ImageIntrinsics.h -- pinhole camera {fu, fv, u0, v0}, 16 bytes (reference
// include/kangaroo/ImageIntrinsics.h:51-200): Project :87-91, Unproject :109-131, pyramid level :137-142.
#pragma once

#include <kangaroo/Image.h>
#include <kangaroo/MatUtils.h>

namespace roo
{

struct ImageIntrinsics
{
    KANGAROO_HD ImageIntrinsics() : fu(0), fv(0), u0(0), v0(0) {}
    KANGAROO_HD ImageIntrinsics(float fu_, float fv_, float u0_, float v0_) : fu(fu_), fv(fv_), u0(u0_), v0(v0_) {}
    KANGAROO_HD ImageIntrinsics(float f, float u0_, float v0_) : fu(f), fv(f), u0(u0_), v0(v0_) {}
    template<typename T, typename Target, typename Manage>
    KANGAROO_HD ImageIntrinsics(float f, const Image<T, Target, Manage>& img)
        : fu(f), fv(f), u0(img.w / 2.0f - 0.5), v0(img.h / 2.0f - 0.5) {}

    KANGAROO_HD float2 Project(const float3 P_c) const { return make_float2(u0 + fu * P_c.x / P_c.z, v0 + fv * P_c.y / P_c.z); }
    KANGAROO_HD float2 Project(float x, float y, float z) const { return make_float2(u0 + fu * x / z, v0 + fv * y / z); }
    KANGAROO_HD float2 operator*(float3 P_c) const { return Project(P_c); }

    KANGAROO_HD float3 Unproject(float u, float v) const { return make_float3((u - u0) / fu, (v - v0) / fv, 1); }
    KANGAROO_HD float3 Unproject(const float2 p_c) const { return Unproject(p_c.x, p_c.y); }
    KANGAROO_HD float3 Unproject(float u, float v, float z) const { return make_float3(z * (u - u0) / fu, z * (v - v0) / fv, z); }
    KANGAROO_HD float3 Unproject(const float2 p_c, float z) const { return Unproject(p_c.x, p_c.y, z); }

    // intrinsics of level l of a power-of-two pyramid (pixel centres at integer coordinates)
    KANGAROO_HD ImageIntrinsics operator[](int l) const
    {
        const float scale = 1.0f / (1 << l);
        return ImageIntrinsics(scale * fu, scale * fv, scale * (u0 + 0.5f) - 0.5f, scale * (v0 + 0.5f) - 0.5f);
    }

    float fu, fv, u0, v0;
};

}
