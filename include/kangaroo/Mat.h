// Mat.h -- small dense row-major matrix POD, roo::Mat<P,R,C> (reference include/kangaroo/Mat.h:33-163).
// Only what the volumetric path uses: element access, fill, copy.  48 bytes for Mat<float,3,4>,
// passed by value / by address to the C ABI as float[12].
#pragma once

#include <cmath>
#include <cstddef>

#include <kangaroo/platform.h>

namespace roo
{

template<typename P, unsigned R, unsigned C = 1>
struct Mat
{
    KANGAROO_HD P operator()(int r, int c) const { return m[r * C + c]; }
    KANGAROO_HD P& operator()(int r, int c) { return m[r * C + c]; }
    KANGAROO_HD P operator()(int i) const { return m[i]; }
    KANGAROO_HD P& operator()(int i) { return m[i]; }
    KANGAROO_HD P operator[](int i) const { return m[i]; }
    KANGAROO_HD P& operator[](int i) { return m[i]; }
    KANGAROO_HD unsigned Rows() const { return R; }
    KANGAROO_HD unsigned Cols() const { return C; }

    template<typename P2> KANGAROO_HD void operator=(const Mat<P2, R, C>& rhs)
    {
        for (unsigned i = 0; i < R * C; ++i) m[i] = (P)rhs.m[i];
    }
    KANGAROO_HD void Fill(P v)
    {
        for (unsigned i = 0; i < R * C; ++i) m[i] = v;
    }
    KANGAROO_HD void SetZero() { Fill(0); }
    KANGAROO_HD P Length() const
    {
        P s = 0;
        for (unsigned i = 0; i < R * C; ++i) s += m[i] * m[i];
        return std::sqrt(s);
    }

    P m[R * C];
};

template<typename P, unsigned R, unsigned C> KANGAROO_HD inline Mat<P, R, C> MatZero()
{
    Mat<P, R, C> z;
    z.SetZero();
    return z;
}

// [ I | 0 ] for the 3x4 poses used throughout
KANGAROO_HD inline Mat<float, 3, 4> SE3Identity()
{
    Mat<float, 3, 4> T;
    T.SetZero();
    T(0, 0) = T(1, 1) = T(2, 2) = 1.0f;
    return T;
}

}
