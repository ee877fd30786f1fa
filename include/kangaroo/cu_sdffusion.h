// cu_sdffusion.h -- roo::SdfFuse / SdfReset / SdfSphere with the reference's signatures
// (include/kangaroo/cu_sdffusion.h:13-26), forwarding to the gfx950 kernels behind include/kfx.h.
#pragma once

#include <kfx_extras.h>   // the operators of this header beyond the KinectFusion path
#include <kangaroo/BoundedVolume.h>
#include <kangaroo/Image.h>
#include <kangaroo/ImageIntrinsics.h>
#include <kangaroo/Mat.h>
#include <kangaroo/Sdf.h>
#include <kangaroo/launch_utils.h>
#include <kangaroo/platform.h>

namespace roo
{

KANGAROO_EXPORT inline
void SdfFuse(BoundedVolume<SDF_t> vol, Image<float> depth, Image<float4> norm, Mat<float,3,4> T_cw, ImageIntrinsics K, float trunc_dist, float maxw, float mincostheta )
{
    GpuCheckStatus(kfx_sdf_fuse(vol.abi(), depth.abi(), norm.abi(), T_cw.m, &K.fu, trunc_dist, maxw, mincostheta, 0, 0));
}

KANGAROO_EXPORT inline
void SdfReset(BoundedVolume<SDF_t> vol, float trunc_dist)
{
    GpuCheckStatus(kfx_sdf_reset(vol.abi(), trunc_dist, 0));
}

// colour fusion (reference cu_sdffusion.h:16-22, kernel cu_sdffusion.cu:70-138): colorVol holds grey levels in
// [0,1]; img is the RGB frame of a camera at T_iw with intrinsics Kimg
KANGAROO_EXPORT inline
void SdfFuse(
    BoundedVolume<SDF_t> vol, BoundedVolume<float> colorVol,
    Image<float> depth, Image<float4> norm, Mat<float,3,4> T_cw, ImageIntrinsics K,
    Image<uchar3> img, Mat<float,3,4> T_iw, ImageIntrinsics Kimg,
    float trunc_dist, float max_w, float mincostheta
)
{
    GpuCheckStatus(kfx_sdf_fuse_color(vol.abi(), colorVol.abi(), depth.abi(), norm.abi(), T_cw.m, &K.fu, img.abi(), T_iw.m, &Kimg.fu,
                                      trunc_dist, max_w, mincostheta, 0, 0));
}

// SdfReset(BoundedVolume<float>) fills with 0.5 (cu_sdffusion.cu:166-169)
KANGAROO_EXPORT inline
void SdfReset(BoundedVolume<float> vol)
{
    GpuCheckStatus(kfx_color_reset(vol.abi(), 0));
}

// TSDF value at every pixel's back-projected depth (reference cu_sdffusion.h:30-31, kernel cu_sdffusion.cu:200-225)
KANGAROO_EXPORT inline
void SdfDistance(Image<float> dist, Image<float> depth, BoundedVolume<SDF_t> vol, const Mat<float,3,4> T_wc, ImageIntrinsics K, float trunc_distance)
{
    GpuCheckStatus(kfx_sdf_distance(dist.abi(), depth.abi(), vol.abi(), T_wc.m, &K.fu, trunc_distance, 0));
}

// fp16-cell overloads (config C5)
KANGAROO_EXPORT inline
void SdfFuse(BoundedVolume<SDF_h> vol, Image<float> depth, Image<float4> norm, Mat<float,3,4> T_cw, ImageIntrinsics K, float trunc_dist, float maxw, float mincostheta )
{
    GpuCheckStatus(kfx_sdf_fuse_h(vol.abi(), depth.abi(), norm.abi(), T_cw.m, &K.fu, trunc_dist, maxw, mincostheta, 0, 0));
}

KANGAROO_EXPORT inline
void SdfReset(BoundedVolume<SDF_h> vol, float trunc_dist)
{
    GpuCheckStatus(kfx_sdf_reset_h(vol.abi(), trunc_dist, 0));
}

KANGAROO_EXPORT inline
void SdfSphere(BoundedVolume<SDF_h> vol, float3 center, float r)
{
    const float c[3] = {center.x, center.y, center.z};
    GpuCheckStatus(kfx_sdf_sphere_h(vol.abi(), c, r, 0));
}

KANGAROO_EXPORT inline
void SdfSphere(BoundedVolume<SDF_t> vol, float3 center, float r)
{
    const float c[3] = {center.x, center.y, center.z};
    GpuCheckStatus(kfx_sdf_sphere(vol.abi(), c, r, 0));
}

}
