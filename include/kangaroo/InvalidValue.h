// InvalidValue.h -- the "no measurement" marker of each pixel type (reference
// include/kangaroo/InvalidValue.h:15-82): NaN for float, 0 for (unsigned) char, -1 for int.
#pragma once

#include <cmath>
#include <limits>

#include <kangaroo/platform.h>

namespace roo
{

template<typename T> struct InvalidValue;

template<> struct InvalidValue<float> {
    KANGAROO_HD static float Value() { return std::numeric_limits<float>::quiet_NaN(); }
    KANGAROO_HD static bool IsValid(float v) { return v - v == 0.0f; /* finite */ }
};

template<> struct InvalidValue<char> {
    KANGAROO_HD static char Value() { return 0; }
    KANGAROO_HD static bool IsValid(unsigned char v) { return !v; }
};

template<> struct InvalidValue<unsigned char> {
    KANGAROO_HD static unsigned char Value() { return 0; }
    KANGAROO_HD static bool IsValid(unsigned char v) { return !v; }
};

template<> struct InvalidValue<int> {
    KANGAROO_HD static int Value() { return -1; }
    KANGAROO_HD static bool IsValid(int v) { return v >= 0; }
};

}
