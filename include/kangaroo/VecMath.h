// VecMath.h -- the small amount of float3/float4 arithmetic the containers need, on HIP's native
// vector types.  Semantics follow what the reference's kernels rely on:
// lerp(a,b,t) = a + t*(b-a); clamp(f,a,b) = max(a, min(f,b)); vector / vector is a true
// componentwise division while vector / scalar multiplies by the reciprocal.
// HIP_vector_type already supplies +, -, * and componentwise /; only what is missing is added.
#pragma once

#include <cmath>

#include <kangaroo/platform.h>

namespace roo
{

typedef unsigned int uint;

KANGAROO_HD inline float lerp(float a, float b, float t) { return a + t * (b - a); }
KANGAROO_HD inline float3 lerp(float3 a, float3 b, float t) { return make_float3(lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t)); }
KANGAROO_HD inline float4 lerp(float4 a, float4 b, float t) { return make_float4(lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t), lerp(a.w, b.w, t)); }

KANGAROO_HD inline float clamp(float f, float lo, float hi) { return fmaxf(lo, fminf(f, hi)); }
KANGAROO_HD inline int clamp(int f, int lo, int hi) { return f < lo ? lo : (f > hi ? hi : f); }

KANGAROO_HD inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
KANGAROO_HD inline float length(float3 v) { return sqrtf(dot(v, v)); }
KANGAROO_HD inline float3 scaled(float3 a, float s) { return make_float3(a.x * s, a.y * s, a.z * s); }
// vector / scalar: one reciprocal, three multiplies
KANGAROO_HD inline float3 div_by(float3 a, float s) { const float inv = 1.0f / s; return scaled(a, inv); }
// vector / vector: three true divisions
KANGAROO_HD inline float3 div_cw(float3 a, float3 b) { return make_float3(a.x / b.x, a.y / b.y, a.z / b.z); }
KANGAROO_HD inline float3 sub(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
KANGAROO_HD inline float3 add(float3 a, float3 b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
KANGAROO_HD inline float3 min3(float3 a, float3 b) { return make_float3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
KANGAROO_HD inline float3 max3(float3 a, float3 b) { return make_float3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
KANGAROO_HD inline float3 xyz(float4 a) { return make_float3(a.x, a.y, a.z); }

}

// The reference keeps its vector helpers at global scope (include/kangaroo/CUDA_SDK/cutil_math.h) and applications call them
// unqualified (main.cpp:221: length(vol.VoxelSizeUnits())): the ones applications use are visible there too.
using roo::length;
using roo::dot;
