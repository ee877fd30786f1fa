// BoundingBox.h -- axis-aligned box {float3 boxmin, boxmax}, 24 bytes (reference
// include/kangaroo/BoundingBox.h:11-164), including the frustum fit the application uses to pick
// the region of the volume in view (FitToFrustum :72-96).
#pragma once

#include <iostream>
#include <limits>

#include <kangaroo/ImageIntrinsics.h>
#include <kangaroo/MatUtils.h>

namespace roo
{

struct BoundingBox
{
    KANGAROO_HD BoundingBox() {}
    KANGAROO_HD BoundingBox(const BoundingBox& b) : boxmin(b.boxmin), boxmax(b.boxmax) {}
    KANGAROO_HD BoundingBox(const float3 lo, const float3 hi) : boxmin(lo), boxmax(hi) {}
    BoundingBox(const Mat<float, 3, 4> T_wc, float w, float h, float fu, float fv, float u0, float v0, float near, float far)
    {
        FitToFrustum(T_wc, w, h, fu, fv, u0, v0, near, far);
    }
    BoundingBox(const Mat<float, 3, 4> T_wc, float w, float h, ImageIntrinsics K, float near, float far)
    {
        FitToFrustum(T_wc, w, h, K.fu, K.fv, K.u0, K.v0, near, far);
    }

    KANGAROO_HD float3& Min() { return boxmin; }
    KANGAROO_HD float3 Min() const { return boxmin; }
    KANGAROO_HD float3& Max() { return boxmax; }
    KANGAROO_HD float3 Max() const { return boxmax; }
    KANGAROO_HD float3 Size() const { return sub(boxmax, boxmin); }
    KANGAROO_HD float3 Center() const { return add(boxmin, div_by(sub(boxmax, boxmin), 2.0f)); }

    void Clear()
    {
        const float big = std::numeric_limits<float>::max();
        boxmin = make_float3(big, big, big);
        boxmax = make_float3(-big, -big, -big);
    }
    void Insert(const float3 p)
    {
        boxmax = max3(p, boxmax);
        boxmin = min3(p, boxmin);
    }
    void Insert(const BoundingBox& b)
    {
        boxmin = min3(b.boxmin, boxmin);
        boxmax = max3(b.boxmax, boxmax);
    }
    void Intersect(const BoundingBox& b)
    {
        boxmin = max3(b.boxmin, boxmin);
        boxmax = min3(b.boxmax, boxmax);
    }
    KANGAROO_HD void Enlarge(float3 scale)
    {
        const float3 c = Center();
        const float3 s = Size();
        const float3 half = div_by(make_float3(scale.x * s.x, scale.y * s.y, scale.z * s.z), 2.0f);
        boxmin = sub(c, half);
        boxmax = add(c, half);
    }

    // box around the 8 corners of the view frustum between the near and far planes
    void FitToFrustum(const Mat<float, 3, 4> T_wc, float w, float h, float fu, float fv, float u0, float v0, float near, float far)
    {
        Clear();
        const float3 c_w = SE3Translation(T_wc);
        const float us[2] = {0.f, w}, vs[2] = {0.f, h};
        float3 rays[4];
        for (int j = 0; j < 2; ++j)
            for (int i = 0; i < 2; ++i) rays[2 * j + i] = mulSO3(T_wc, make_float3((us[i] - u0) / fu, (vs[j] - v0) / fv, 1));
        const float dist[2] = {near, far};
        for (int k = 0; k < 2; ++k)
            for (int i = 0; i < 4; ++i) Insert(add(c_w, scaled(rays[i], dist[k])));
    }
    void FitToFrustum(const Mat<float, 3, 4> T_wc, float w, float h, ImageIntrinsics K, float near, float far)
    {
        FitToFrustum(T_wc, w, h, K.fu, K.fv, K.u0, K.v0, near, far);
    }

    float3 boxmin;
    float3 boxmax;
};

inline std::ostream& operator<<(std::ostream& os, const BoundingBox& b)
{
    os << "(" << b.boxmin.x << "," << b.boxmin.y << "," << b.boxmin.z << ") - (" << b.boxmax.x << "," << b.boxmax.y << "," << b.boxmax.z << ")";
    return os;
}

}
