// cu_bilateral.h -- roo::BilateralFilter<To,Ti> with the reference's signatures
// (include/kangaroo/cu_bilateral.h:9-19) for the instantiations the reference exports
// (src/cu_bilateral.cu:52-53, 103-104): <float,float>, <float,unsigned char> without a validity
// threshold; <float,float>, <float,unsigned short> with one.
#pragma once

#include <kfx_extras.h>   // the operators of this header beyond the KinectFusion path
#include <kangaroo/Image.h>
#include <kangaroo/launch_utils.h>
#include <kangaroo/platform.h>

namespace roo
{

template<typename To, typename Ti>
KANGAROO_EXPORT
void BilateralFilter(Image<To> dOut, const Image<Ti> dIn, float gs, float gr, uint size);

template<typename To, typename Ti>
KANGAROO_EXPORT
void BilateralFilter(Image<To> dOut, const Image<Ti> dIn, float gs, float gr, uint size, Ti minval);

// joint bilateral filter with a guide image (reference cu_bilateral.h:21-25; src/cu_bilateral.cu:110-155,
// instantiated for <float,float,unsigned char> and <float,float,float>)
template<typename To, typename Ti, typename Ti2>
KANGAROO_EXPORT
void BilateralFilter(Image<To> dOut, const Image<Ti> dIn, const Image<Ti2> dImg, float gs, float gr, float gc, uint size);

template<> inline void BilateralFilter(Image<float> dOut, const Image<float> dIn, const Image<unsigned char> dImg, float gs, float gr, float gc, uint size)
{
    GpuNoteStatus(kfx_bilateral_guided_u8(dOut.abi(), dIn.abi(), dImg.abi(), gs, gr, gc, size, 0));
}
template<> inline void BilateralFilter(Image<float> dOut, const Image<float> dIn, const Image<float> dImg, float gs, float gr, float gc, uint size)
{
    GpuNoteStatus(kfx_bilateral_guided_f32(dOut.abi(), dIn.abi(), dImg.abi(), gs, gr, gc, size, 0));
}

template<> inline void BilateralFilter(Image<float> dOut, const Image<float> dIn, float gs, float gr, uint size)
{
    GpuNoteStatus(kfx_bilateral_f32(dOut.abi(), dIn.abi(), gs, gr, size, 0.f, 0, 0));
}
template<> inline void BilateralFilter(Image<float> dOut, const Image<unsigned char> dIn, float gs, float gr, uint size)
{
    GpuNoteStatus(kfx_bilateral_u8(dOut.abi(), dIn.abi(), gs, gr, size, 0));
}
template<> inline void BilateralFilter(Image<float> dOut, const Image<float> dIn, float gs, float gr, uint size, float minval)
{
    GpuNoteStatus(kfx_bilateral_f32(dOut.abi(), dIn.abi(), gs, gr, size, minval, 1, 0));
}
template<> inline void BilateralFilter(Image<float> dOut, const Image<unsigned short> dIn, float gs, float gr, uint size, unsigned short minval)
{
    GpuNoteStatus(kfx_bilateral_u16(dOut.abi(), dIn.abi(), gs, gr, size, minval, 0));
}

}
