// reweighting.h -- robust-estimation weight functions (reference include/kangaroo/reweighting.h:5-35).
// w(r, c) multiplies a residual's contribution to the normal equations; c is the scale of the estimator.
#pragma once

#include <cmath>

#include <kangaroo/platform.h>

namespace roo
{

KANGAROO_HD inline float LSReweightSq(float /*r*/, float /*c*/) { return 1; }

KANGAROO_HD inline float LSReweightL1(float r, float /*c*/) { return 1.0f / std::fabs(r); }

KANGAROO_HD inline float LSReweightHuber(float r, float c)
{
    const float a = std::fabs(r);
    return a <= c ? 1.0f : c / a;
}

// (1 - (r/c)^2)^2 inside the cut-off, 0 outside: the weight the ICP kernel applies (icp.hip)
KANGAROO_HD inline float LSReweightTukey(float r, float c)
{
    const float q = r / c;
    const float t = 1.0f - q * q;
    return std::fabs(r) <= c ? t * t : 0.0f;
}

KANGAROO_HD inline float LSReweightCauchy(float r, float c)
{
    const float q = r / c;
    return 1.0f / (1.0f + q * q);
}

}
