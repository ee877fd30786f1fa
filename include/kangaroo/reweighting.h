// reweighting.h -- robust-estimation weights w(r, c) of the iteratively reweighted least-squares solvers
// (reference include/kangaroo/reweighting.h:5-35: LSReweightSq / L1 / Huber / Tukey / Cauchy).  r is a residual, c the
// estimator's scale; the ICP kernel (kangaroo_amd/csrc/icp.hip) applies the Tukey weight with the same operation order.
#pragma once

#include <cmath>

#include <kangaroo/platform.h>

namespace roo
{
namespace robust
{
enum Estimator { Quadratic, AbsoluteValue, Huber, TukeyBiweight, Cauchy };

// One definition for all estimators; every branch is a compile-time constant at the call sites below.
template<Estimator E>
KANGAROO_HD inline float Weight(const float r, const float c)
{
    const float magnitude = std::fabs(r);
    if (E == AbsoluteValue) return 1.0f / magnitude;
    if (E == Huber) return magnitude <= c ? 1.0f : c / magnitude;
    const float scaled = r / c;                    // residual in units of the scale
    if (E == TukeyBiweight) {
        const float bell = 1.0f - scaled * scaled; // biweight: bell squared inside the cut-off, nothing outside
        return magnitude <= c ? bell * bell : 0.0f;
    }
    if (E == Cauchy) return 1.0f / (1.0f + scaled * scaled);
    return 1;                                      // Quadratic: ordinary least squares
}
}

KANGAROO_HD inline float LSReweightSq(float r, float c) { return robust::Weight<robust::Quadratic>(r, c); }
KANGAROO_HD inline float LSReweightL1(float r, float c) { return robust::Weight<robust::AbsoluteValue>(r, c); }
KANGAROO_HD inline float LSReweightHuber(float r, float c) { return robust::Weight<robust::Huber>(r, c); }
KANGAROO_HD inline float LSReweightTukey(float r, float c) { return robust::Weight<robust::TukeyBiweight>(r, c); }
KANGAROO_HD inline float LSReweightCauchy(float r, float c) { return robust::Weight<robust::Cauchy>(r, c); }

}
