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

// ---- fp16 cell (BASELINE config C5: 2048^3 TSDF = 32 GiB) -----------------------------------
// {half val; half w;}, 4 bytes.  Follows the reference's commented-out half SDF_t (Sdf.h:38-62):
// the running average rounds every intermediate to half (round-to-nearest-even).  Conversions are
// done with integer arithmetic so that the header needs no half type from the host compiler.
namespace half_bits
{
KANGAROO_HD inline unsigned short from_float(float f)
{
    union { float f; unsigned u; } in;
    in.f = f;
    const unsigned sign = (in.u >> 16) & 0x8000u;
    const unsigned x = in.u & 0x7fffffffu;
    if (x > 0x7f800000u) return (unsigned short)(sign | 0x7e00u);             // NaN
    if (x >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u);            // rounds to inf (>= 65520)
    if (x < 0x33000001u) return (unsigned short)sign;                         // rounds to zero (<= 2^-25)
    const int e = (int)(x >> 23) - 127;
    unsigned mant = (x & 0x7fffffu) | 0x800000u;
    int shift, hexp;
    if (e < -14) { shift = 13 + (-14 - e); hexp = 0; } else { shift = 13; hexp = e + 15; }
    unsigned h = mant >> shift;
    const unsigned rem = mant & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (h & 1u))) ++h;                    // round to nearest even
    // h holds the implicit bit for normals: adding (hexp - 1) << 10 folds it into the exponent, and a
    // mantissa carry propagates into the exponent the same way
    const unsigned out = hexp > 0 ? h + ((unsigned)(hexp - 1) << 10) : h;
    return (unsigned short)(sign | out);
}
KANGAROO_HD inline float to_float(unsigned short h)
{
    const unsigned sign = ((unsigned)h & 0x8000u) << 16;
    unsigned e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    union { float f; unsigned u; } out;
    if (e == 0x1f) out.u = sign | 0x7f800000u | (m << 13);
    else if (e == 0) {
        if (m == 0) out.u = sign;
        else {
            int k = 0;
            while (!(m & 0x400u)) { m <<= 1; ++k; }
            out.u = sign | ((unsigned)(127 - 15 + 1 - k) << 23) | ((m & 0x3ffu) << 13);
        }
    } else out.u = sign | ((e + 127 - 15) << 23) | (m << 13);
    return out.f;
}
KANGAROO_HD inline float q(float f) { return to_float(from_float(f)); }
}

struct alignas(4) SDF_h
{
    KANGAROO_HD SDF_h() {}
    KANGAROO_HD SDF_h(float v) : val(half_bits::from_float(v)), w(half_bits::from_float(1.0f)) {}
    KANGAROO_HD SDF_h(float v, float weight) : val(half_bits::from_float(v)), w(half_bits::from_float(weight)) {}

    KANGAROO_HD operator float() const { return half_bits::to_float(val); }
    KANGAROO_HD float Weight() const { return half_bits::to_float(w); }
    KANGAROO_HD void LimitWeight(float max_weight) { w = half_bits::from_float(fminf(half_bits::to_float(w), max_weight)); }
    KANGAROO_HD void operator+=(const SDF_h& rhs)
    {
        using namespace half_bits;
        if (to_float(rhs.w) > 0) {
            val = from_float(to_float(w) * to_float(val) + to_float(rhs.w) * to_float(rhs.val));
            w = from_float(to_float(w) + to_float(rhs.w));
            val = from_float(to_float(val) / to_float(w));
        }
    }

    unsigned short val;
    unsigned short w;
};

KANGAROO_HD inline SDF_t operator+(const SDF_t& lhs, const SDF_t& rhs)
{
    SDF_t r = lhs;
    r += rhs;
    return r;
}

}
