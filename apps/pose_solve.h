// pose_solve.h -- the double-precision host step of the tracking loop (reference application,
// applications/kinectfusion/main.cpp:312-333).  The reference uses Eigen::FullPivLU and Sophus::SE3d::exp
// (external libraries, not in its tree); the same published algorithms are written out here for 6x6 / 3x3
// systems: LU with complete pivoting (rank by Eigen's default threshold eps * n * max|pivot|, free
// variables of a rank-deficient system = 0) and the closed-form SE(3) exponential.
// Same functions as kangaroo_amd/tracking.py.
#pragma once

#include <cmath>
#include <limits>
#include <utility>

namespace posesolve
{

struct SE3d {
    double R[3][3];
    double t[3];
    SE3d()
    {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) R[i][j] = i == j ? 1.0 : 0.0;
            t[i] = 0.0;
        }
    }
    SE3d operator*(const SE3d& o) const
    {
        SE3d r;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) r.R[i][j] = R[i][0] * o.R[0][j] + R[i][1] * o.R[1][j] + R[i][2] * o.R[2][j];
            r.t[i] = R[i][0] * o.t[0] + R[i][1] * o.t[1] + R[i][2] * o.t[2] + t[i];
        }
        return r;
    }
    SE3d inverse() const
    {
        SE3d r;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) r.R[i][j] = R[j][i];
        for (int i = 0; i < 3; ++i) r.t[i] = -(r.R[i][0] * t[0] + r.R[i][1] * t[1] + r.R[i][2] * t[2]);
        return r;
    }
    // row-major 3x4 [R | t] in float (the implicit Eigen -> roo::Mat<float,3,4> conversion of the application)
    template<typename M34> M34 matrix3x4() const
    {
        M34 m;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) m(i, j) = (float)R[i][j];
            m(i, 3) = (float)t[i];
        }
        return m;
    }
};

// x = FullPivLU(A).solve(b), A is n x n row-major (n <= 6)
template<int n> inline void FullPivLuSolve(const double* A, const double* b, double* x)
{
    double lu[n][n];
    int rows[n], cols[n];
    for (int i = 0; i < n; ++i) {
        rows[i] = cols[i] = i;
        for (int j = 0; j < n; ++j) lu[i][j] = A[i * n + j];
    }
    int nonzero = n;
    double maxpivot = 0.0;
    for (int k = 0; k < n; ++k) {
        int pr = k, pc = k;
        double biggest = 0.0;
        for (int i = k; i < n; ++i)
            for (int j = k; j < n; ++j)
                if (std::fabs(lu[i][j]) > biggest) { biggest = std::fabs(lu[i][j]); pr = i; pc = j; }
        if (biggest == 0.0) { nonzero = k; break; }
        if (biggest > maxpivot) maxpivot = biggest;
        if (pr != k) {
            for (int j = 0; j < n; ++j) std::swap(lu[k][j], lu[pr][j]);
            std::swap(rows[k], rows[pr]);
        }
        if (pc != k) {
            for (int i = 0; i < n; ++i) std::swap(lu[i][k], lu[i][pc]);
            std::swap(cols[k], cols[pc]);
        }
        for (int i = k + 1; i < n; ++i) {
            lu[i][k] /= lu[k][k];
            for (int j = k + 1; j < n; ++j) lu[i][j] -= lu[i][k] * lu[k][j];
        }
    }
    const double thresh = std::numeric_limits<double>::epsilon() * n * maxpivot;
    int rank = 0;
    for (int i = 0; i < nonzero; ++i) rank += std::fabs(lu[i][i]) > thresh ? 1 : 0;
    for (int i = 0; i < n; ++i) x[i] = 0.0;
    if (rank == 0) return;
    double c[n], y[n];
    for (int i = 0; i < n; ++i) {
        c[i] = b[rows[i]];
        for (int j = 0; j < i; ++j) c[i] -= lu[i][j] * c[j];
        y[i] = 0.0;
    }
    for (int i = rank - 1; i >= 0; --i) {
        double s = c[i];
        for (int j = i + 1; j < rank; ++j) s -= lu[i][j] * y[j];
        y[i] = s / lu[i][i];
    }
    for (int i = 0; i < n; ++i) x[cols[i]] = y[i];
}

// exp of (upsilon, omega): R = I + sin(t)/t W + (1-cos t)/t^2 W^2, p = (I + (1-cos t)/t^2 W + (t-sin t)/t^3 W^2) upsilon
inline SE3d Exp(const double x[6], bool rotation_only = false)
{
    const double* w = x + 3;
    const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = std::sqrt(t2);
    const double W[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
    double W2[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) W2[i][j] = W[i][0] * W[0][j] + W[i][1] * W[1][j] + W[i][2] * W[2][j];
    const bool small = t < 1e-10;
    const double a = small ? 1.0 : std::sin(t) / t, b = small ? 0.5 : (1.0 - std::cos(t)) / t2,
                 c = small ? 1.0 / 6.0 : (t - std::sin(t)) / (t2 * t);
    SE3d T;
    for (int i = 0; i < 3; ++i) {
        double p = 0.0;
        for (int j = 0; j < 3; ++j) {
            T.R[i][j] = (i == j ? 1.0 : 0.0) + a * W[i][j] + b * W2[i][j];
            p += ((i == j ? 1.0 : 0.0) + b * W[i][j] + c * W2[i][j]) * x[j];
        }
        T.t[i] = rotation_only ? 0.0 : p;
    }
    return T;
}

} // namespace posesolve
