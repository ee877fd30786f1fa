/*
 * kfx_extras.h -- operators of the reference's cu_bilateral.h / cu_depth_tools.h / cu_raycast.h / cu_sdffusion.h that are NOT on
 * the KinectFusion hot path (SURVEY.md 2 marks them out of scope; they complete the five headers the path's operators live in so
 * that an application including <kangaroo/kangaroo.h> finds every declaration).  Same library (libkfx.so), same conventions as
 * kfx.h, bit-exact against the oracle like the rest (tests/test_gpu_parity.py); kept apart so that kfx.h is the path and nothing else.
 */
#ifndef KFX_EXTRAS_H
#define KFX_EXTRAS_H

#include "kfx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* BilateralFilter(dOut, dIn, dImg, gs, gr, gc, size) (cu_bilateral.cu:110-155): joint bilateral filter of a float image
 * with a float / unsigned char guide image (third weight exp(-(guide difference)^2 / 2 gc^2)); sumw == 0 keeps the input. */
int kfx_bilateral_guided_f32(const kfx_image* out, const kfx_image* in, const kfx_image* guide, float gs, float gr, float gc,
                             unsigned size, kfx_stream stream);
int kfx_bilateral_guided_u8(const kfx_image* out, const kfx_image* in, const kfx_image* guide, float gs, float gr, float gc,
                            unsigned size, kfx_stream stream);

/* ---- the small per-pixel tools of cu_depth_tools.h -----------------------------------------------------
 * kfx_disp2depth:           Disp2Depth(dIn, dOut, fu, fBaseline, fMinDisp) (cu_depth_tools.cu:15-30)
 * kfx_filter_bad_kinect_*:  FilterBadKinectData(dFiltered, dKinectDepth) for float / unsigned short readings (:32-53)
 * kfx_colour_vbo:           ColourVbo(dId, dPd, dIc, KT_cd) (:86-119): uchar4 colour per vertex from an Image<uchar3> */
int kfx_disp2depth(const kfx_image* in, const kfx_image* out, float fu, float baseline, float min_disp, kfx_stream stream);
int kfx_filter_bad_kinect_f32(const kfx_image* out, const kfx_image* in, kfx_stream stream);
int kfx_filter_bad_kinect_u16(const kfx_image* out, const kfx_image* in, kfx_stream stream);
int kfx_colour_vbo(const kfx_image* id, const kfx_image* vbo, const kfx_image* rgb, const float KT_cd[12], kfx_stream stream);

/* roo::ImageKeyframe<uchar3> (ImageKeyframe.h:10-14 over ImageIntrinsics.h:202-212): {ImageIntrinsics K; Mat<float,3,4> T_iw;
 * Image<uchar3> img}, 96 bytes, same field order. */
typedef struct kfx_keyframe {
    float K[4];
    float T_iw[12];
    kfx_image img;
} kfx_keyframe;
/* TextureDepth (cu_depth_tools.cu:123-207): colour every pixel of a rendered depth / normal image from RGB keyframes.
 * phong == NULL, n_kf == 1: TextureDepth<float4,uchar3>(img, kf, depth, norm, T_wd, Kdepth) -- the keyframe's colour where the
 *   point projects inside it and faces it (camera-frame normal z < -0.2), else black.
 * phong != NULL, n_kf <= 10: TextureDepth<float4,uchar3,10>(img, kfs, depth, norm, phong, T_wd, Kdepth) -- keyframes blended by
 *   the cosine between normal and viewing ray (> 0.1, in front of the keyframe), the Phong image where none applies; a keyframe
 *   with img.ptr == NULL ends the list.  (The reference accumulates into an uninitialised float3; it is zero here.) */
int kfx_texture_depth(const kfx_image* img, const kfx_keyframe* kfs, int n_kf, const kfx_image* depth, const kfx_image* norm,
                      const kfx_image* phong, const float T_wd[12], const float Kdepth[4], kfx_stream stream);

/* ---- the rest of cu_raycast.h / cu_sdffusion.h ---------------------------------------------------------
 * kfx_raycast_box:    RaycastBox(imgd, T_wc, K, bbox) (cu_raycast.cu:202-240) -- entry depth into the box, NaN on a miss.
 * kfx_raycast_sphere: RaycastSphere(imgd, img, T_wc, K, center, r) (:246-279) -- a sphere hit nearer than the depth already
 *                     in imgd (or where that is not finite) overwrites imgd and, if img->ptr, img with its Phong shade.
 * kfx_raycast_plane:  RaycastPlane(imgd, img, T_wc, K, n_w) (:285-310) -- the plane n_w . x = -1, same overwrite rule.
 * kfx_sdf_distance:   SdfDistance(dist, depth, vol, T_wc, K, trunc_distance) (cu_sdffusion.cu:200-225) -- the trilinear
 *                     TSDF value at every pixel's back-projected depth point. */
int kfx_raycast_box(const kfx_image* imgd, const float T_wc[12], const float K[4], const float boxmin[3], const float boxmax[3],
                    kfx_stream stream);
int kfx_raycast_sphere(const kfx_image* imgd, const kfx_image* img, const float T_wc[12], const float K[4], const float center[3],
                       float r, kfx_stream stream);
int kfx_raycast_plane(const kfx_image* imgd, const kfx_image* img, const float T_wc[12], const float K[4], const float n_w[3],
                      kfx_stream stream);
int kfx_sdf_distance(const kfx_image* dist, const kfx_image* depth, const kfx_volume* vol, const float T_wc[12], const float K[4],
                     float trunc_distance, kfx_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* KFX_EXTRAS_H */
