// cu_operations.h -- roo::ElementwiseScaleBias<Tout,Tin,Tup> (reference include/kangaroo/cu_operations.h,
// src/cu_operations.cu:39-57): b = s*a + offset.  Instantiated for <float,float,float>, the millimetre ->
// metre conversion of the KinectFusion frame loop (applications/kinectfusion/main.cpp:208).
#pragma once

#include <kangaroo/Image.h>
#include <kangaroo/launch_utils.h>
#include <kangaroo/platform.h>

namespace roo
{

template<typename Tout, typename Tin, typename Tup>
KANGAROO_EXPORT
void ElementwiseScaleBias(Image<Tout> b, const Image<Tin> a, float s, Tup offset=0);

template<> inline void ElementwiseScaleBias(Image<float> b, const Image<float> a, float s, float offset)
{
    GpuNoteStatus(kfx_elementwise_scale_bias_f32(b.abi(), a.abi(), s, offset, 0));
}

}
