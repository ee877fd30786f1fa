// cu_depth_tools.h -- roo::DepthToVbo<T> with the reference's signatures
// (include/kangaroo/cu_depth_tools.h:19-27), instantiated for float and unsigned short depth
// (src/cu_depth_tools.cu:216-217), and the header's small per-pixel tools Disp2Depth, FilterBadKinectData and
// ColourVbo.  Not provided: TextureDepth (keyframe texturing of the GUI's view mode, SURVEY: out of scope).
#pragma once

#include <kangaroo/Image.h>
#include <kangaroo/ImageIntrinsics.h>
#include <kangaroo/Mat.h>
#include <kangaroo/launch_utils.h>
#include <kangaroo/platform.h>

namespace roo
{

// reference cu_depth_tools.h:11-17 (kernels cu_depth_tools.cu:15-53)
KANGAROO_EXPORT inline
void Disp2Depth(Image<float> dIn, const Image<float> dOut, float fu, float fBaseline, float fMinDisp = 0.0)
{
    GpuNoteStatus(kfx_disp2depth(dIn.abi(), dOut.abi(), fu, fBaseline, fMinDisp, 0));
}

KANGAROO_EXPORT inline
void FilterBadKinectData(Image<float> dFiltered, Image<unsigned short> dKinectDepth)
{
    GpuNoteStatus(kfx_filter_bad_kinect_u16(dFiltered.abi(), dKinectDepth.abi(), 0));
}

KANGAROO_EXPORT inline
void FilterBadKinectData(Image<float> dFiltered, Image<float> dKinectDepth)
{
    GpuNoteStatus(kfx_filter_bad_kinect_f32(dFiltered.abi(), dKinectDepth.abi(), 0));
}

template<typename T>
KANGAROO_EXPORT
void DepthToVbo( Image<float4> dVbo, const Image<T> dKinectDepth, ImageIntrinsics K, float scale = 1.0f);

template<> inline void DepthToVbo( Image<float4> dVbo, const Image<float> dKinectDepth, ImageIntrinsics K, float scale)
{
    GpuNoteStatus(kfx_depth_to_vbo_f32(dVbo.abi(), dKinectDepth.abi(), &K.fu, scale, 0));
}
template<> inline void DepthToVbo( Image<float4> dVbo, const Image<unsigned short> dKinectDepth, ImageIntrinsics K, float scale)
{
    GpuNoteStatus(kfx_depth_to_vbo_u16(dVbo.abi(), dKinectDepth.abi(), &K.fu, scale, 0));
}

template<typename T>
inline void DepthToVbo( Image<float4> dVbo, const Image<T> dKinectDepth, float fu, float fv, float u0, float v0, float scale = 1.0f)
{
    DepthToVbo<T>(dVbo, dKinectDepth, ImageIntrinsics(fu,fv,u0,v0), scale);
}

// reference cu_depth_tools.h:30 (kernel cu_depth_tools.cu:86-119)
KANGAROO_EXPORT inline
void ColourVbo(Image<uchar4> dId, const Image<float4> dPd, const Image<uchar3> dIc, const Mat<float,3,4> KT_cd )
{
    GpuNoteStatus(kfx_colour_vbo(dId.abi(), dPd.abi(), dIc.abi(), KT_cd.m, 0));
}

}
