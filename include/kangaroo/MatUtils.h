// MatUtils.h -- rigid-transform helpers on roo::Mat<float,3,4> (reference include/kangaroo/MatUtils.h:
// operator* :117-125, mulSO3 :147-155, mulSO3inv :177-185, mulSE3 / mulSE3inv :187-200, SE3inv :202-214,
// SE3Translation :216-220, Plane_b_from_a :474-488).  Left-to-right sums, as the kernels compute them.
#pragma once

#include <kangaroo/Mat.h>
#include <kangaroo/VecMath.h>

namespace roo
{

KANGAROO_HD inline float3 operator*(const Mat<float, 3, 4>& T_ba, const float3& p_a)
{
    return make_float3(T_ba(0, 0) * p_a.x + T_ba(0, 1) * p_a.y + T_ba(0, 2) * p_a.z + T_ba(0, 3),
                       T_ba(1, 0) * p_a.x + T_ba(1, 1) * p_a.y + T_ba(1, 2) * p_a.z + T_ba(1, 3),
                       T_ba(2, 0) * p_a.x + T_ba(2, 1) * p_a.y + T_ba(2, 2) * p_a.z + T_ba(2, 3));
}
KANGAROO_HD inline float3 mulSE3(const Mat<float, 3, 4>& T_ab, const float3& r_a) { return T_ab * r_a; }

KANGAROO_HD inline float3 mulSO3(const Mat<float, 3, 4>& T_ab, const float3& r_a)
{
    return make_float3(T_ab(0, 0) * r_a.x + T_ab(0, 1) * r_a.y + T_ab(0, 2) * r_a.z,
                       T_ab(1, 0) * r_a.x + T_ab(1, 1) * r_a.y + T_ab(1, 2) * r_a.z,
                       T_ab(2, 0) * r_a.x + T_ab(2, 1) * r_a.y + T_ab(2, 2) * r_a.z);
}

KANGAROO_HD inline float3 mulSO3inv(const Mat<float, 3, 4>& T_ba, const float3& r_a)
{
    return make_float3(T_ba(0, 0) * r_a.x + T_ba(1, 0) * r_a.y + T_ba(2, 0) * r_a.z,
                       T_ba(0, 1) * r_a.x + T_ba(1, 1) * r_a.y + T_ba(2, 1) * r_a.z,
                       T_ba(0, 2) * r_a.x + T_ba(1, 2) * r_a.y + T_ba(2, 2) * r_a.z);
}

KANGAROO_HD inline float3 mulSE3inv(const Mat<float, 3, 4>& T_ab, const float3& r_a)
{
    const float dx = r_a.x - T_ab(0, 3), dy = r_a.y - T_ab(1, 3), dz = r_a.z - T_ab(2, 3);
    return make_float3(T_ab(0, 0) * dx + T_ab(1, 0) * dy + T_ab(2, 0) * dz,
                       T_ab(0, 1) * dx + T_ab(1, 1) * dy + T_ab(2, 1) * dz,
                       T_ab(0, 2) * dx + T_ab(1, 2) * dy + T_ab(2, 2) * dz);
}

KANGAROO_HD inline Mat<float, 3, 4> SE3inv(const Mat<float, 3, 4>& T_ba)
{
    Mat<float, 3, 4> T_ab;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) T_ab(r, c) = T_ba(c, r);
    for (int r = 0; r < 3; ++r)
        T_ab(r, 3) = -(T_ab(r, 0) * T_ba(0, 3) + T_ab(r, 1) * T_ba(1, 3) + T_ab(r, 2) * T_ba(2, 3));
    return T_ab;
}

KANGAROO_HD inline float3 SE3Translation(const Mat<float, 3, 4>& T_ba)
{
    return make_float3(T_ba(0, 3), T_ba(1, 3), T_ba(2, 3));
}

// plane n.x = -1 expressed in frame a, re-expressed in frame b
KANGAROO_HD inline float3 Plane_b_from_a(const Mat<float, 3, 4> T_ab, const float3 n_a)
{
    const float s = T_ab(0, 3) * n_a.x + T_ab(1, 3) * n_a.y + T_ab(2, 3) * n_a.z + 1.0f;
    return make_float3((T_ab(0, 0) * n_a.x + T_ab(1, 0) * n_a.y + T_ab(2, 0) * n_a.z) / s,
                       (T_ab(0, 1) * n_a.x + T_ab(1, 1) * n_a.y + T_ab(2, 1) * n_a.z) / s,
                       (T_ab(0, 2) * n_a.x + T_ab(1, 2) * n_a.y + T_ab(2, 2) * n_a.z) / s);
}

}
