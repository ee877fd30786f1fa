// cu_depth_tools.h -- roo::DepthToVbo<T> with the reference's signatures
// (include/kangaroo/cu_depth_tools.h:19-27), instantiated for float and unsigned short depth
// (src/cu_depth_tools.cu:216-217).
#pragma once

#include <kangaroo/Image.h>
#include <kangaroo/ImageIntrinsics.h>
#include <kangaroo/launch_utils.h>
#include <kangaroo/platform.h>

namespace roo
{

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

}
