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

// N x N symmetric matrix, unique elements only, lower triangle in row-major order
// (reference Mat.h:353-447).  Converts to the full Mat<PT,N,N>.
template<typename P, unsigned N>
struct SymMat
{
    static const unsigned int unique = N * (N + 1) / 2;

    template<typename PT> KANGAROO_HD operator Mat<PT, N, N>() const
    {
        Mat<PT, N, N> full;
        unsigned i = 0;
        for (unsigned r = 0; r < N; ++r)
            for (unsigned c = 0; c <= r; ++c) {
                const PT e = (PT)m[i++];
                full(r, c) = e;
                full(c, r) = e;
            }
        return full;
    }
    KANGAROO_HD void SetZero()
    {
        for (unsigned i = 0; i < unique; ++i) m[i] = 0;
    }
    template<typename P2> KANGAROO_HD void operator+=(const SymMat<P2, N>& rhs)
    {
        for (unsigned i = 0; i < unique; ++i) m[i] += rhs.m[i];
    }
    KANGAROO_HD void operator*=(const P w)
    {
        for (unsigned i = 0; i < unique; ++i) m[i] *= w;
    }

    P m[unique];
};

// Normal equations of a least-squares problem in N unknowns (reference Mat.h:483-520).
// LeastSquaresSystem<float,6> is 116 bytes and layout-identical to kfx_lss6.
template<typename P, unsigned N>
struct LeastSquaresSystem
{
    Mat<P, N, 1> JTy;
    SymMat<P, N> JTJ;
    P sqErr;
    unsigned obs;

    KANGAROO_HD void SetZero()
    {
        JTJ.SetZero();
        JTy.SetZero();
        sqErr = 0;
        obs = 0;
    }
    template<typename P2> KANGAROO_HD void operator+=(const LeastSquaresSystem<P2, N>& rhs)
    {
        for (unsigned i = 0; i < N; ++i) JTy.m[i] += rhs.JTy.m[i];
        JTJ += rhs.JTJ;
        sqErr += rhs.sqErr;
        obs += rhs.obs;
    }
};

// [ I | 0 ] for the 3x4 poses used throughout
KANGAROO_HD inline Mat<float, 3, 4> SE3Identity()
{
    Mat<float, 3, 4> T;
    T.SetZero();
    T(0, 0) = T(1, 1) = T(2, 2) = 1.0f;
    return T;
}

}
