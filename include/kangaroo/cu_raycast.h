// cu_raycast.h -- roo::RaycastSdf with the reference's signature (include/kangaroo/cu_raycast.h:13-14).
#pragma once

#include <kfx_extras.h>   // the operators of this header beyond the KinectFusion path
#include <kangaroo/BoundedVolume.h>
#include <kangaroo/BoundingBox.h>
#include <kangaroo/Image.h>
#include <kangaroo/ImageIntrinsics.h>
#include <kangaroo/Mat.h>
#include <kangaroo/Sdf.h>
#include <kangaroo/launch_utils.h>
#include <kangaroo/platform.h>

namespace roo
{

KANGAROO_EXPORT inline
void RaycastSdf(Image<float> depth, Image<float4> norm, Image<float> img, const BoundedVolume<SDF_t> vol, const Mat<float,3,4> T_wc, ImageIntrinsics K, float near, float far, float trunc_dist, bool subpix = true)
{
    GpuCheckStatus(kfx_raycast_sdf(depth.abi(), norm.abi(), img.abi(), vol.abi(), T_wc.m, &K.fu, near, far, trunc_dist, subpix ? 1 : 0, 0));
}


// Addition next to the reference API: the per-level RaycastSdf calls of the tracking loop
// (applications/kinectfusion/main.cpp:280-288) as one launch -- same images, bit for bit, but the levels' marches
// overlap (kfx_raycast_sdf_levels).  depth / norm / img / K are arrays of n entries.
KANGAROO_EXPORT inline
void RaycastSdfLevels(const Image<float>* depth, const Image<float4>* norm, const Image<float>* img, unsigned n, const BoundedVolume<SDF_t> vol, const Mat<float,3,4> T_wc, const ImageIntrinsics* K, float near, float far, float trunc_dist, bool subpix = true, const Image<float4>* vbo = 0)
{
    const kfx_image *d[8], *nn[8], *im[8], *vb[8];
    float k[32];
    if (n > 8) GpuCheckStatus(KFX_E_RANGE);
    for (unsigned l = 0; l < n && l < 8; ++l) {
        d[l] = depth[l].abi(); nn[l] = norm[l].abi(); im[l] = img[l].abi();
        vb[l] = vbo ? vbo[l].abi() : 0;   // vbo[l] receives DepthToVbo(vbo[l], depth[l], K[l]) from the same launch
        k[4 * l] = K[l].fu; k[4 * l + 1] = K[l].fv; k[4 * l + 2] = K[l].u0; k[4 * l + 3] = K[l].v0;
    }
    GpuCheckStatus(kfx_raycast_sdf_levels((int)n, d, nn, im, vbo ? vb : 0, vol.abi(), T_wc.m, k, near, far, trunc_dist, subpix ? 1 : 0, 0));
}

// colour overload (reference cu_raycast.h:16-17, kernel cu_raycast.cu:119-196): img = colour volume sampled at the hit
KANGAROO_EXPORT inline
void RaycastSdf(Image<float> depth, Image<float4> norm, Image<float> img, const BoundedVolume<SDF_t> vol, const BoundedVolume<float> colorVol, const Mat<float,3,4> T_wc, ImageIntrinsics K, float near, float far, float trunc_dist, bool subpix = true)
{
    GpuCheckStatus(kfx_raycast_sdf_color(depth.abi(), norm.abi(), img.abi(), vol.abi(), colorVol.abi(), T_wc.m, &K.fu, near, far, trunc_dist, subpix ? 1 : 0, 0));
}

// analytic renderers (reference cu_raycast.h:19-26, kernels cu_raycast.cu:202-310)
KANGAROO_EXPORT inline
void RaycastBox(Image<float> depth, const Mat<float,3,4> T_wc, ImageIntrinsics K, const BoundingBox bbox )
{
    const float3 lo = bbox.Min(), hi = bbox.Max();
    const float a[3] = {lo.x, lo.y, lo.z}, b[3] = {hi.x, hi.y, hi.z};
    GpuCheckStatus(kfx_raycast_box(depth.abi(), T_wc.m, &K.fu, a, b, 0));
}

KANGAROO_EXPORT inline
void RaycastSphere(Image<float> depth, Image<float> img, const Mat<float,3,4> T_wc, ImageIntrinsics K, float3 center, float r)
{
    const float c[3] = {center.x, center.y, center.z};
    GpuCheckStatus(kfx_raycast_sphere(depth.abi(), img.abi(), T_wc.m, &K.fu, c, r, 0));
}

KANGAROO_EXPORT inline
void RaycastPlane(Image<float> depth, Image<float> img, const Mat<float,3,4> T_wc, ImageIntrinsics K, const float3 n_w )
{
    const float n[3] = {n_w.x, n_w.y, n_w.z};
    GpuCheckStatus(kfx_raycast_plane(depth.abi(), img.abi(), T_wc.m, &K.fu, n, 0));
}

// fp16-cell overload (config C5)
KANGAROO_EXPORT inline
void RaycastSdf(Image<float> depth, Image<float4> norm, Image<float> img, const BoundedVolume<SDF_h> vol, const Mat<float,3,4> T_wc, ImageIntrinsics K, float near, float far, float trunc_dist, bool subpix = true)
{
    GpuCheckStatus(kfx_raycast_sdf_h(depth.abi(), norm.abi(), img.abi(), vol.abi(), T_wc.m, &K.fu, near, far, trunc_dist, subpix ? 1 : 0, 0));
}

}
