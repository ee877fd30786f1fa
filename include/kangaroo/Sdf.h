// Sdf.h -- the TSDF cell, roo::SDF_t {float val; float w;}, 8 bytes, 8-byte aligned
// (reference include/kangaroo/Sdf.h:11-36).  operator+= is the weighted running average the
// fusion kernel applies: `new += old` keeps `new` untouched while old.w <= 0 (the (NaN, 0)
// "never observed" state, quirk Q3).
#pragma once

#include <kangaroo/VecMath.h>

namespace roo
{

struct alignas(8) SDF_t
{
    KANGAROO_HD SDF_t() {}
    KANGAROO_HD SDF_t(float v) : val(v), w(1) {}
    KANGAROO_HD SDF_t(float v, float weight) : val(v), w(weight) {}

    KANGAROO_HD operator float() const { return val; }
    KANGAROO_HD void Clamp(float lo, float hi) { val = clamp(val, lo, hi); }
    KANGAROO_HD void LimitWeight(float max_weight) { w = fminf(w, max_weight); }
    KANGAROO_HD void operator+=(const SDF_t& rhs)
    {
        if (rhs.w > 0) {
            val = (w * val + rhs.w * rhs.val);
            w += rhs.w;
            val /= w;
        }
    }

    float val;
    float w;
};

KANGAROO_HD inline SDF_t operator+(const SDF_t& lhs, const SDF_t& rhs)
{
    SDF_t r = lhs;
    r += rhs;
    return r;
}

}
