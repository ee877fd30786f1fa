// cu_depth_tools.h -- roo::DepthToVbo<T> with the reference's signatures
// (include/kangaroo/cu_depth_tools.h:19-27), instantiated for float and unsigned short depth
// (src/cu_depth_tools.cu:216-217), and the header's small per-pixel tools Disp2Depth, FilterBadKinectData and
// ColourVbo and TextureDepth (ImageKeyframe is declared here too).
#pragma once

#include <kfx_extras.h>   // the operators of this header beyond the KinectFusion path
#include <kangaroo/Image.h>
#include <kangaroo/ImageIntrinsics.h>
#include <kangaroo/Mat.h>
#include <kangaroo/MatUtils.h>
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

// Addition next to the reference API: DepthToVbo<float> followed by NormalsFromVbo in one launch; dVbo and dN hold exactly
// what the two separate calls write (kfx_depth_to_vbo_normals_f32).
inline void DepthToVboNormals( Image<float4> dVbo, Image<float4> dN, const Image<float> dKinectDepth, ImageIntrinsics K, float scale = 1.0f)
{
    GpuNoteStatus(kfx_depth_to_vbo_normals_f32(dVbo.abi(), dN.abi(), dKinectDepth.abi(), &K.fu, scale, 0));
}

// BoxReduceIgnoreInvalid(depth) followed by DepthToVbo + NormalsFromVbo on every level (main.cpp:211-218) as ONE launch (addition
// beside the reference API; the 2 * Levels - 1 separate launches are latency-sized): depth[0] is the input, K[l]
// (ImageIntrinsics::operator[]) the intrinsics of level l.  Same images.  Up to four levels; a level of size zero ends the chain as in BoxReduceIgnoreInvalid.
template<unsigned Levels, typename PyrD, typename PyrV>
inline void DepthPyramidVboNormals(PyrD& depth, PyrV& vbo, PyrV& nrm, const ImageIntrinsics& K, float scale = 1.0f)
{
    static_assert(Levels >= 1 && Levels <= 4, "DepthPyramidVboNormals: one to four levels");
    kfx_image d[Levels], v[Levels], n[Levels];
    float k[4 * Levels];
    int levels = 0;
    for (unsigned l = 0; l < Levels && depth[l].w != 0 && depth[l].h != 0; ++l, ++levels) {
        d[l] = *depth[l].abi(); v[l] = *vbo[l].abi(); n[l] = *nrm[l].abi();
        const ImageIntrinsics Kl = K[(int)l];
        k[4 * l] = Kl.fu; k[4 * l + 1] = Kl.fv; k[4 * l + 2] = Kl.u0; k[4 * l + 3] = Kl.v0;
    }
    if (levels > 0) GpuNoteStatus(kfx_depth_pyramid_vbo_normals_f32(d, v, n, k, levels, scale, 0));
}

// roo::ImageKeyframe<T> (reference ImageKeyframe.h:10-14 over ImageTransformProject, ImageIntrinsics.h:202-212): a camera
// {K, T_iw} with its image; 96 bytes, layout-identical to kfx_keyframe.
struct ImageTransformProject
{
    float2 Project(const float3 P_w) const
    {
        const float3 P_i = T_iw * P_w;
        return make_float2(K.u0 + K.fu * P_i.x / P_i.z, K.v0 + K.fv * P_i.y / P_i.z);
    }
    ImageIntrinsics K;
    Mat<float,3,4> T_iw;
};
template<typename T, typename Target = TargetDevice, typename Management = DontManage>
struct ImageKeyframe : public ImageTransformProject
{
    Image<T, TargetDevice> img;
};
static_assert(sizeof(ImageKeyframe<uchar3>) == sizeof(kfx_keyframe), "ImageKeyframe<uchar3> must match kfx_keyframe");

// reference cu_depth_tools.h:33-38 (kernels cu_depth_tools.cu:123-207), instantiated as the reference does: <float4,uchar3>
template<typename Tout, typename Tin>
KANGAROO_EXPORT
void TextureDepth(Image<Tout> img, const ImageKeyframe<Tin> kf, const Image<float> depth, const Image<float4> norm, const Mat<float,3,4> T_wd, ImageIntrinsics Kdepth);

template<> inline void TextureDepth(Image<float4> img, const ImageKeyframe<uchar3> kf, const Image<float> depth, const Image<float4> norm, const Mat<float,3,4> T_wd, ImageIntrinsics Kdepth)
{
    GpuNoteStatus(kfx_texture_depth(img.abi(), reinterpret_cast<const kfx_keyframe*>(&kf), 1, depth.abi(), norm.abi(), nullptr, T_wd.m, &Kdepth.fu, 0));
}

template<typename Tout, typename Tin, size_t N>   // (Mat's row count is unsigned: N is never deduced, callers name it, main.cpp:269)
KANGAROO_EXPORT
void TextureDepth(Image<Tout> img, const Mat<ImageKeyframe<Tin>,N> kfs, const Image<float> depth, const Image<float4> norm, const Image<float> phong, const Mat<float,3,4> T_wd, ImageIntrinsics Kdepth);

template<> inline void TextureDepth<float4,uchar3,10>(Image<float4> img, const Mat<ImageKeyframe<uchar3>,10> kfs, const Image<float> depth, const Image<float4> norm, const Image<float> phong, const Mat<float,3,4> T_wd, ImageIntrinsics Kdepth)
{
    GpuNoteStatus(kfx_texture_depth(img.abi(), reinterpret_cast<const kfx_keyframe*>(kfs.m), 10, depth.abi(), norm.abi(), phong.abi(), T_wd.m, &Kdepth.fu, 0));
}

// reference cu_depth_tools.h:30 (kernel cu_depth_tools.cu:86-119)
KANGAROO_EXPORT inline
void ColourVbo(Image<uchar4> dId, const Image<float4> dPd, const Image<uchar3> dIc, const Mat<float,3,4> KT_cd )
{
    GpuNoteStatus(kfx_colour_vbo(dId.abi(), dPd.abi(), dIc.abi(), KT_cd.m, 0));
}

}
