/*
 * kfx_oracle.h -- CPU restatement of the KinectFusion volumetric hot path of
 * arpg/Kangaroo.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle: a plain-C, IEEE-fp32, no-FMA restatement of the
 * reference's algorithm, each function citing the reference file:line it
 * follows (paths relative to the reference tree).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (kangaroo_amd/, include/) never links, imports or calls it.
 *
 * Pinning status: the reference ships no tests / golden vectors for this path
 * (SURVEY.md section 4).  The helper arithmetic is pinned against the
 * reference's own headers compiled in place (oracle/_ref, see
 * oracle/ref_harness.cpp); the CUDA kernels themselves (*.cu) need nvcc and
 * are unbuildable here, so the kernel-level control flow is restated by
 * reading and is otherwise "parity unpinned".  See DESIGN.md section 3.
 *
 * Semantics: CUDA *device* semantics for fminf/fmaxf (IEEE minNum/maxNum, C99
 * fminf/fmaxf) -- the reference's host fallbacks in cutil_math.h:55-63 are
 * ternaries and differ only when an operand is NaN.
 */
#ifndef KFX_ORACLE_H
#define KFX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order as roo::Image (include/kangaroo/Image.h:617-620). */
typedef struct kfo_image {
    size_t pitch; /* bytes per row */
    void*  ptr;
    size_t w;
    size_t h;
} kfo_image;

/* Same field order as roo::BoundedVolume (Volume.h:363-369 + BoundedVolume.h:168,
 * BoundingBox.h:163-164). Element = SDF_t {float val; float w;} (Sdf.h:34-35). */
typedef struct kfo_volume {
    size_t pitch;     /* bytes per x-row */
    void*  ptr;
    size_t w;
    size_t h;
    size_t img_pitch; /* bytes per z-slice */
    size_t d;
    float  boxmin[3];
    float  boxmax[3];
} kfo_volume;

typedef struct kfo_raycast_stats {
    uint64_t rays;        /* rays whose segment intersects the box */
    uint64_t steps;       /* total trilinear samples taken */
    uint64_t hits;        /* rays with depth > 0 */
} kfo_raycast_stats;

/* cu_bilateral.cu:13-41 (use_minval=0) and :59-92 (use_minval=1), <float,float> */
void kfo_bilateral_f32(const kfo_image* out, const kfo_image* in, float gs, float gr,
                       int size, float minval, int use_minval, int nthreads);
/* cu_bilateral.cu:59-92 <float,unsigned short> */
void kfo_bilateral_u16(const kfo_image* out, const kfo_image* in, float gs, float gr,
                       int size, unsigned short minval, int nthreads);
/* cu_bilateral.cu:13-41 <float,unsigned char> */
void kfo_bilateral_u8(const kfo_image* out, const kfo_image* in, float gs, float gr,
                      int size, int nthreads);
/* cu_depth_tools.cu:59-78 <float> / <unsigned short> */
void kfo_depth_to_vbo_f32(const kfo_image* vbo, const kfo_image* depth, const float K[4], float scale);
void kfo_depth_to_vbo_u16(const kfo_image* vbo, const kfo_image* depth, const float K[4], float scale);
/* cu_normals.cu:12-45 */
void kfo_normals_from_vbo(const kfo_image* nrm, const kfo_image* vbo);
/* cu_sdffusion.cu:153-164 (Volume.h:343-356: contiguous span incl. pitch padding) */
void kfo_sdf_reset(const kfo_volume* vol, float trunc);
/* cu_sdffusion.cu:175-195 */
void kfo_sdf_sphere(const kfo_volume* vol, const float center[3], float r);
/* cu_sdffusion.cu:16-61. full_extent=0 reproduces the reference's w/8,h/8,d/8
 * integer-division grid (quirk Q1); returns number of voxels updated. */
uint64_t kfo_sdf_fuse(const kfo_volume* vol, const kfo_image* depth, const kfo_image* norm,
                      const float T_cw[12], const float K[4], float trunc, float max_w,
                      float mincostheta, int full_extent, int nthreads);
/* Z-slab of a larger volume (multi-GPU partition): `vol` holds planes [z_offset, z_offset + d) of a volume
 * with full_d planes spanning [full_zmin, full_zmax]; z positions use the full volume's expression. */
typedef struct kfo_slab {
    size_t full_d;
    size_t z_offset;
    float  full_zmin;
    float  full_zmax;
} kfo_slab;
uint64_t kfo_sdf_fuse_slab(const kfo_volume* vol, const kfo_slab* slab, const kfo_image* depth, const kfo_image* norm,
                           const float T_cw[12], const float K[4], float trunc, float max_w,
                           float mincostheta, int full_extent, int nthreads);
/* colour fusion / raycast: cvol is a BoundedVolume<float> (4-byte cells), img an Image<uchar3>.
 * cu_sdffusion.cu:70-138, :166-169; cu_raycast.cu:119-196 */
void kfo_color_reset(const kfo_volume* cvol);
uint64_t kfo_sdf_fuse_color(const kfo_volume* vol, const kfo_volume* cvol, const kfo_image* depth, const kfo_image* normals,
                            const float T_cw[12], const float K[4], const kfo_image* img, const float T_iw[12], const float Kimg[4],
                            float trunc_dist, float max_w, float mincostheta, int full_extent, int nthreads);
void kfo_raycast_sdf_color(const kfo_image* depth, const kfo_image* norm, const kfo_image* img, const kfo_volume* vol,
                           const kfo_volume* cvol, const float T_wc[12], const float K[4], float near, float far, float trunc,
                           int subpix, int nthreads);

/* ImageKeyframe<uchar3> and TextureDepth (cu_depth_tools.cu:123-207); phong == NULL: single-keyframe form */
typedef struct kfo_keyframe {
    float K[4];
    float T_iw[12];
    kfo_image img;
} kfo_keyframe;
void kfo_texture_depth(const kfo_image* out, const kfo_keyframe* kfs, int n_kf, const kfo_image* depth, const kfo_image* norm,
                       const kfo_image* phong, const float T_wd[12], const float Kdepth[4]);

/* cu_bilateral.cu:110-155 */
void kfo_bilateral_guided(const kfo_image* out, const kfo_image* in, const kfo_image* guide, int guide_is_u8, float gs, float gr, float gc,
                          int size);

/* cu_depth_tools.cu:15-53, :86-119 */
void kfo_disp2depth(const kfo_image* in, const kfo_image* out, float fu, float baseline, float min_disp);
void kfo_filter_bad_kinect(const kfo_image* out, const kfo_image* in, int in_is_u16);
void kfo_colour_vbo(const kfo_image* id, const kfo_image* vbo, const kfo_image* rgb, const float KT_cd[12]);

/* cu_sdffusion.cu:200-225 */
void kfo_sdf_distance(const kfo_image* dist, const kfo_image* depth, const kfo_volume* vol, const float T_wc[12], const float K[4]);

/* MarchingCubes.h:43-143 in SaveMesh's loop nest (:226-232); case tables supplied by the caller */
uint64_t kfo_marching_cubes(const kfo_volume* vol, const kfo_volume* cvol, const unsigned char* ntris, const unsigned short* emask,
                            const signed char* tris, int tri_stride, float* verts, float* norms, float* colors);

/* roo::LeastSquaresSystem<float,6> (Mat.h:483-520) */
typedef struct kfo_lss6 {
    float JTy[6];
    float JTJ[21]; /* lower triangle, row-major (Mat.h:353-365) */
    float sqErr;
    unsigned obs;
} kfo_lss6;
/* cu_model_refinement.cu:541-608 + LeastSquareSum.h:71-85; block_sums (may be NULL): one system per 16x16 block */
void kfo_icp_block_dims(size_t w, size_t h, unsigned out[4]);
void kfo_icp_point_plane(const kfo_image* Pl, const kfo_image* Pr, const kfo_image* Nr, const float KT_lr[12],
                         const float T_rl[12], float c, const kfo_image* debug, kfo_lss6* out, kfo_lss6* block_sums);

/* One round of the exact multi-GPU march (state carried across Z-slabs); see kfx_oracle.c */
void kfo_raycast_sdf_slab(float* state, int init, const kfo_volume* vol, const kfo_slab* slab, int own_lo, int own_hi,
                          int w, int h, const float T_wc[12], const float K[4], float near, float far,
                          float trunc_dist, int subpix);
/* cu_raycast.cu:14-113 */
void kfo_raycast_sdf(const kfo_image* depth, const kfo_image* norm, const kfo_image* img,
                     const kfo_volume* vol, const float T_wc[12], const float K[4],
                     float near, float far, float trunc, int subpix, int nthreads,
                     kfo_raycast_stats* stats);
/* fp16-cell variants (BASELINE config C5): volume cells are {half val; half w;} (4 bytes), the
 * arithmetic of the reference's commented-out half SDF_t (Sdf.h:38-62). */
uint64_t kfo_sdf_fuse_h(const kfo_volume* vol, const kfo_image* depth, const kfo_image* norm,
                        const float T_cw[12], const float K[4], float trunc, float max_w,
                        float mincostheta, int full_extent, int nthreads);
void kfo_raycast_sdf_h(const kfo_image* depth, const kfo_image* norm, const kfo_image* img,
                       const kfo_volume* vol, const float T_wc[12], const float K[4],
                       float near, float far, float trunc, int subpix, int nthreads,
                       kfo_raycast_stats* stats);
void kfo_sdf_reset_h(const kfo_volume* vol, float trunc);
void kfo_sdf_sphere_h(const kfo_volume* vol, const float center[3], float r);

/* Like kfo_raycast_sdf but also marks every distinct voxel the rays touch
 * (trilinear corners + normal stencil) in `bitmap` (1 bit per voxel, index
 * (z*h + y)*w + x), for the algorithmic-bytes figure of SURVEY 8(d). */
void kfo_raycast_sdf_touch(const kfo_image* depth, const kfo_image* norm, const kfo_image* img,
                           const kfo_volume* vol, const float T_wc[12], const float K[4],
                           float near, float far, float trunc, int subpix,
                           uint8_t* bitmap, kfo_raycast_stats* stats);

/* Analytic depth renderers, cu_raycast.cu:202-310 */
void kfo_raycast_box(const kfo_image* depth, const float T_wc[12], const float K[4],
                     const float boxmin[3], const float boxmax[3]);
void kfo_raycast_sphere(const kfo_image* depth, const kfo_image* img /* may have ptr NULL */,
                        const float T_wc[12], const float K[4], const float center[3], float r);
void kfo_raycast_plane(const kfo_image* depth, const kfo_image* img, const float T_wc[12],
                       const float K[4], const float n_w[3]);

/* Synthetic scenes of SURVEY 8(d) (not reference code): depth in metres, NaN = invalid.
 * scene 0 = S_room (box interior x,y in +-0.9, back wall z=3.8, sphere c=(0,0,3) r=0.5),
 * scene 1 = S_full (flat wall z = 5.95). Rays are cast from pose T_wc (camera->world). */
void kfo_render_scene(const kfo_image* depth, int scene, const float T_wc[12], const float K[4]);

/* BoundedVolume::SubBoundingVolume (BoundedVolume.h:137-165) and
 * BoundingBox::FitToFrustum (BoundingBox.h:72-96): host-side ROI helpers. */
void kfo_fit_to_frustum(float boxmin[3], float boxmax[3], const float T_wc[12], float w, float h,
                        const float K[4], float near, float far);
void kfo_sub_bounding_volume(kfo_volume* out, const kfo_volume* vol, const float rmin[3],
                             const float rmax[3]);

/* SE3inv (MatUtils.h:202-214) */
void kfo_se3_inverse(float out[12], const float T[12]);

/* Point-wise helpers (exposed so tests can pin them against the reference headers):
 * GetUnitsTrilinearClamped (BoundedVolume.h:93-98), GetUnitsBackwardDiffDxDyDz (:100-106),
 * SDF_t::operator+= + LimitWeight (Sdf.h:22-32), ImageIntrinsics::operator[] (ImageIntrinsics.h:137-142),
 * VoxelPositionInUnits (BoundedVolume.h:115-125). */
float kfo_trilinear(const kfo_volume* vol, const float pos_w[3]);
void kfo_gradient(const kfo_volume* vol, const float pos_w[3], float out[3]);
void kfo_sdf_accumulate(float val, float w, float old_val, float old_w, float max_w, float out[2]);
float kfo_phong_shade(const float p_c[3], const float n_c[3]); /* PhongShade, cu_raycast.cu:14-28 */
void kfo_intrinsics_level(float out[4], const float K[4], int level);
void kfo_voxel_position(const kfo_volume* vol, int x, int y, int z, float out[3]);

/* cu_operations.cu:39-57 <float,float,float>; cu_resample.cu:89-120 <float,float,float> */
void kfo_elementwise_scale_bias_f32(const kfo_image* b, const kfo_image* a, float s, float offset);
void kfo_box_half_ignore_invalid_f32(const kfo_image* out, const kfo_image* in);

int kfo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
