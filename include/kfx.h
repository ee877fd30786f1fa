/*
 * kfx.h -- C ABI of libkfx: the MI355X (gfx950) KinectFusion volumetric path.
 *
 * This is the drop-in boundary for the hot path of arpg/Kangaroo
 * (BilateralFilter -> DepthToVbo -> NormalsFromVbo -> SdfFuse -> RaycastSdf).
 * The reference has no FFI: its boundary is a set of C++ free functions in
 * namespace roo taking pitched view structs by value.  Each
 * entry point below replaces one of those functions; the header-only wrappers
 * in include/kangaroo/cu_*.h keep the reference signatures verbatim and forward
 * here.  Plain pointers and sizes only -- no C++ or torch types.
 *
 * Conventions
 *  - kfx_image / kfx_volume have exactly the memory layout of roo::Image<T> and
 *    roo::BoundedVolume<T> (reference include/kangaroo/Image.h:617-620,
 *    Volume.h:363-369, BoundedVolume.h:168, BoundingBox.h:163-164), so a
 *    roo:: view can be passed by address without conversion.
 *  - All image/volume pointers are DEVICE pointers (HBM).  Views are non-owning
 *    (reference Memory.h:152-164 "DontManage"); the library never allocates or
 *    frees behind the caller's back and keeps no scratch state.
 *  - Poses are row-major 3x4 float (roo::Mat<float,3,4>, Mat.h:33-163);
 *    intrinsics are {fu, fv, u0, v0} (ImageIntrinsics.h:51-200).
 *  - `stream` is a hipStream_t (NULL = default stream).  Launches are
 *    asynchronous, ordered on that stream, like the reference's default-stream
 *    launches (launch_utils.h:8: no sync).
 *  - Return value: 0 = ok; > 0 = hipError_t from the launch; < 0 = argument
 *    error (KFX_E_*).  Never exits the process: the roo:: wrappers map non-zero
 *    to the reference's print-and-exit(-1) convention (launch_utils.h:29-47).
 *  - Default arithmetic (KFX_MATH_EXACT) is IEEE binary32, no FMA contraction,
 *    correctly rounded div and sqrt, in the reference's operation order: results
 *    are bit-identical to the CPU oracle for every op except the bilateral filter,
 *    whose __expf is a hardware approximation as in the reference.
 *    kfx_set_math_mode(KFX_MATH_FAST) switches kfx_sdf_fuse[_h] to hardware
 *    rcp/rsq + FMA (the regime of the reference's own -use_fast_math build).
 */
#ifndef KFX_H
#define KFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KFX_VERSION_MAJOR 0
#define KFX_VERSION_MINOR 1

/* argument errors */
#define KFX_E_NULL      (-1) /* null struct or data pointer */
#define KFX_E_SHAPE     (-2) /* inconsistent / zero / too small dimensions */
#define KFX_E_ALIGN     (-3) /* pointer or pitch not aligned for the element type */
#define KFX_E_RANGE     (-4) /* parameter out of supported range */
#define KFX_E_NODEVICE  (-5) /* no HIP device */
#define KFX_E_TIMEOUT   (-6) /* a point-to-point leg of the in-process transport found no matching partner in time (kfx_slab.h) */

/* roo::Image<T,Target,Management>: {size_t pitch; T* ptr; size_t w; size_t h;} */
typedef struct kfx_image {
    size_t pitch; /* bytes per row */
    void*  ptr;
    size_t w;
    size_t h;
} kfx_image;

/* roo::BoundedVolume<T,...>: Volume {pitch, ptr, w, h, img_pitch, d} + BoundingBox {boxmin, boxmax}.
 * For the TSDF volume T = SDF_t {float val; float w;} (8 bytes, Sdf.h:11-36), x fastest. */
typedef struct kfx_volume {
    size_t pitch;     /* bytes per x-row */
    void*  ptr;
    size_t w;
    size_t h;
    size_t img_pitch; /* bytes per z-slice */
    size_t d;
    float  boxmin[3];
    float  boxmax[3];
} kfx_volume;

typedef void* kfx_stream; /* hipStream_t */

/* flags for kfx_sdf_fuse */
#define KFX_FUSE_FULL_EXTENT 1u /* also integrate the trailing dim%8 voxels the reference's
                                   integer-division grid skips (cu_sdffusion.cu:57-59) */

#define KFX_FUSE_SLAB_EXTENT 2u /* kfx_sdf_fuse_slab*: integrate exactly the voxels the reference integrates on the WHOLE
                                   volume -- x / y extents (dim/8)*8, and the local planes below (full_d/8)*8 -- whatever
                                   the slab's own plane count is */

/* ---- the hot path ------------------------------------------------------------ */

/* roo::SdfFuse(BoundedVolume<SDF_t>, Image<float>, Image<float4>, Mat<float,3,4> T_cw,
 *              ImageIntrinsics, float trunc_dist, float maxw, float mincostheta)
 * reference: include/kangaroo/cu_sdffusion.h:13-14, src/cu_sdffusion.cu:16-61 */
int kfx_sdf_fuse(const kfx_volume* vol, const kfx_image* depth, const kfx_image* norm,
                 const float T_cw[12], const float K[4], float trunc_dist, float max_w,
                 float mincostheta, unsigned flags, kfx_stream stream);

/* Z-slab of a larger volume (multi-GPU partition): `vol` holds planes
 * [z_offset, z_offset + vol->d) of a volume with `full_d` planes whose box spans [full_zmin, full_zmax]
 * in z (x/y extents and box are those of `vol`).  Voxel positions are evaluated with the FULL volume's
 * expression (BoundedVolume.h:115-125), so a slab is integrated bit-identically to the same planes of the
 * monolithic volume.  Pass KFX_FUSE_FULL_EXTENT when vol->d is not a multiple of 8. */
typedef struct kfx_slab {
    size_t full_d;
    size_t z_offset;
    float  full_zmin;
    float  full_zmax;
} kfx_slab;
int kfx_sdf_fuse_slab(const kfx_volume* vol, const kfx_slab* slab, const kfx_image* depth, const kfx_image* norm,
                      const float T_cw[12], const float K[4], float trunc_dist, float max_w,
                      float mincostheta, unsigned flags, kfx_stream stream);

/* Diagnostics (no reference counterpart): number of voxels kfx_sdf_fuse would update for
 * this frame -- the same projection / lookup / predicate (cu_sdffusion.cu:22-44) evaluated
 * without touching the volume data.  *d_count is a DEVICE uint64 the caller zeroes; the count
 * is added to it.  bench.py uses it for the algorithmic-bytes figure 16 B x N_updated. */
int kfx_sdf_fuse_count(const kfx_volume* vol, const kfx_image* depth, const kfx_image* norm,
                       const float T_cw[12], const float K[4], float trunc_dist,
                       float mincostheta, unsigned flags, unsigned long long* d_count,
                       kfx_stream stream);

/* Diagnostics (no reference counterpart): the march of kfx_raycast_sdf (fp32 cells) for a w x h image with every cell it reads
 * -- the eight of each sample, the gradient stencil of each hit -- marked in d_bitmap (one bit per voxel, dense x-fastest
 * index, ceil(w_vol h_vol d_vol / 32) words, zeroed by the caller).  Added to the DEVICE counters d_counters[4]: samples taken,
 * rays that enter the box, hits, distinct voxels touched (U).  bench.py prices RaycastSdf's algorithmic bytes with it:
 * 8 B x U + 24 B x w h.  Writes no image. */
int kfx_raycast_sdf_count(const kfx_volume* vol, unsigned w, unsigned h, const float T_wc[12], const float K[4], float near, float far,
                          float trunc_dist, int subpix, unsigned* d_bitmap, unsigned long long* d_counters, kfx_stream stream);
/* ... for half cells (kfx_raycast_sdf_h; config C5: the 2048^3 bitmap is 1 GiB) */
int kfx_raycast_sdf_count_h(const kfx_volume* vol, unsigned w, unsigned h, const float T_wc[12], const float K[4], float near, float far,
                            float trunc_dist, int subpix, unsigned* d_bitmap, unsigned long long* d_counters, kfx_stream stream);

/* roo::RaycastSdf(Image<float> depth, Image<float4> norm, Image<float> img,
 *                 const BoundedVolume<SDF_t>, const Mat<float,3,4> T_wc, ImageIntrinsics,
 *                 float near, float far, float trunc_dist, bool subpix)
 * reference: include/kangaroo/cu_raycast.h:13-14, src/cu_raycast.cu:14-113 */
int kfx_raycast_sdf(const kfx_image* depth, const kfx_image* norm, const kfx_image* img,
                    const kfx_volume* vol, const float T_wc[12], const float K[4], float near,
                    float far, float trunc_dist, int subpix, kfx_stream stream);

/* The per-level RaycastSdf calls of the tracking loop (applications/kinectfusion/main.cpp:280-288: one call per
 * pyramid level with ICP iterations, same model, same pose, intrinsics K[l]) as ONE launch.  depth / norm / img are
 * arrays of n_levels image pointers, K holds n_levels x 4 floats {fu, fv, u0, v0}; n_levels <= 8.  Every image is
 * bit-identical to what its own kfx_raycast_sdf call writes; the levels' marches overlap instead of running one
 * after the other (a coarse level takes as long as the full-resolution one: DESIGN.md 5.2).  No counterpart in the
 * reference API -- an addition next to it; the roo:: wrapper is RaycastSdfLevels (include/kangaroo/cu_raycast.h).
 * vbo: NULL, or n_levels pointers (each may be NULL) to Image<float4> vertex maps that receive DepthToVbo<float>(vbo[l],
 * depth[l], K[l]) -- the call that follows each RaycastSdf in the application (main.cpp:286) -- from the same launch. */
int kfx_raycast_sdf_levels(int n_levels, const kfx_image* const* depth, const kfx_image* const* norm, const kfx_image* const* img,
                           const kfx_image* const* vbo, const kfx_volume* vol, const float T_wc[12], const float* K, float near, float far,
                           float trunc_dist, int subpix, kfx_stream stream);
int kfx_raycast_sdf_levels_h(int n_levels, const kfx_image* const* depth, const kfx_image* const* norm, const kfx_image* const* img,
                             const kfx_image* const* vbo, const kfx_volume* vol, const float T_wc[12], const float* K, float near, float far,
                             float trunc_dist, int subpix, kfx_stream stream);

/* roo::BilateralFilter<float,float>(Image<float>, Image<float>, gs, gr, size[, minval])
 * reference: include/kangaroo/cu_bilateral.h:9-19, src/cu_bilateral.cu:13-53 (use_minval=0),
 * :59-104 (use_minval=1) */
int kfx_bilateral_f32(const kfx_image* out, const kfx_image* in, float gs, float gr, unsigned size,
                      float minval, int use_minval, kfx_stream stream);
/* roo::BilateralFilter<float,unsigned short>(..., unsigned short minval): cu_bilateral.cu:104 */
int kfx_bilateral_u16(const kfx_image* out, const kfx_image* in, float gs, float gr, unsigned size,
                      unsigned short minval, kfx_stream stream);
/* roo::BilateralFilter<float,unsigned char>(...): cu_bilateral.cu:53 */
int kfx_bilateral_u8(const kfx_image* out, const kfx_image* in, float gs, float gr, unsigned size,
                     kfx_stream stream);

/* roo::DepthToVbo<float|unsigned short>(Image<float4>, Image<T>, ImageIntrinsics, float scale)
 * reference: include/kangaroo/cu_depth_tools.h:19-27, src/cu_depth_tools.cu:59-78,216-217 */
int kfx_depth_to_vbo_f32(const kfx_image* vbo, const kfx_image* depth, const float K[4], float scale,
                         kfx_stream stream);
int kfx_depth_to_vbo_u16(const kfx_image* vbo, const kfx_image* depth, const float K[4], float scale,
                         kfx_stream stream);

/* roo::NormalsFromVbo(Image<float4> dN, Image<float4> dV)
 * reference: include/kangaroo/cu_normals.h:9-10, src/cu_normals.cu:12-45 */
int kfx_normals_from_vbo(const kfx_image* nrm, const kfx_image* vbo, kfx_stream stream);

/* Frame pre-amble of the application around the path:
 * roo::ElementwiseScaleBias<float,float,float>(b, a, s, offset): b = s*a + offset (millimetres -> metres,
 *   applications/kinectfusion/main.cpp:208) -- reference: src/cu_operations.cu:39-57
 * roo::BoxHalfIgnoreInvalid<float,float,float>(out, in): 2x2 mean of the finite samples, NaN if none; one level
 *   of roo::BoxReduceIgnoreInvalid (main.cpp:211) -- reference: src/cu_resample.cu:89-120, reduce.h:48-59 */
int kfx_elementwise_scale_bias_f32(const kfx_image* out, const kfx_image* in, float s, float offset, kfx_stream stream);
int kfx_box_half_ignore_invalid_f32(const kfx_image* out, const kfx_image* in, kfx_stream stream);

/* roo::SdfReset(BoundedVolume<SDF_t>, float trunc_dist): fills (trunc_dist, 0) over the
 * contiguous span ptr .. RowPtr(h-1,d-1)+w including pitch padding
 * reference: cu_sdffusion.h:20, src/cu_sdffusion.cu:153-164, Volume.h:343-356 */
int kfx_sdf_reset(const kfx_volume* vol, float trunc_dist, kfx_stream stream);

/* roo::SdfSphere(BoundedVolume<SDF_t>, float3 center, float r)
 * reference: cu_sdffusion.h:26, src/cu_sdffusion.cu:175-195 */
int kfx_sdf_sphere(const kfx_volume* vol, const float center[3], float r, kfx_stream stream);

/* ---- fp16 TSDF (BASELINE config C5) ---------------------------------------------------------
 * Same operators on a volume of roo::SDF_h {half val; half w;} cells (4 bytes, include/kangaroo/Sdf.h):
 * the arithmetic of the reference's commented-out half SDF_t (Sdf.h:38-62), every intermediate of the
 * running average rounded to half (round-to-nearest-even).  2048^3 = 32 GiB fits one MI355X. */
int kfx_sdf_fuse_h(const kfx_volume* vol, const kfx_image* depth, const kfx_image* norm,
                   const float T_cw[12], const float K[4], float trunc_dist, float max_w,
                   float mincostheta, unsigned flags, kfx_stream stream);
int kfx_raycast_sdf_h(const kfx_image* depth, const kfx_image* norm, const kfx_image* img,
                      const kfx_volume* vol, const float T_wc[12], const float K[4], float near,
                      float far, float trunc_dist, int subpix, kfx_stream stream);
int kfx_sdf_reset_h(const kfx_volume* vol, float trunc_dist, kfx_stream stream);
int kfx_sdf_sphere_h(const kfx_volume* vol, const float center[3], float r, kfx_stream stream);

/* ---- device allocator: roo::TargetDevice (Memory.h:59-84) ----------------------- */

/* AllocatePitchedMem: rows padded to a multiple of 256 bytes (512 B coalescing
 * segments stay aligned); *pitch receives the row stride in bytes. */
int kfx_alloc_pitched(void** dev_ptr, size_t* pitch, size_t width_bytes, size_t rows);
int kfx_free(void* dev_ptr);
/* roo::TargetHost (Memory.h:32-57): page-locked host memory, pitch = width (no padding). */
int kfx_alloc_host(void** host_ptr, size_t bytes);
int kfx_free_host(void* host_ptr);
/* 2-D copies behind Image::CopyFrom / MemcpyFromHost / MemcpyToHost (Image.h:174-213).
 * kind: 0 host->host, 1 host->device, 2 device->host, 3 device->device, 4 default (stream null: blocking, like cudaMemcpy2D in the
 * reference; else asynchronous on `stream`); 5 device->device enqueued on `stream` whatever it is -- the null stream too */
int kfx_memcpy_2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes,
                  size_t rows, int kind, kfx_stream stream);
int kfx_stream_synchronize(kfx_stream stream);

/* DepthToVbo<float>(vbo, depth, K, scale) followed by NormalsFromVbo(nrm, vbo) in one launch (no reference counterpart:
 * both are launch-latency sized at VGA).  The neighbour vertices a normal needs are recomputed from `depth` with
 * DepthToVbo's expression, so vbo and nrm hold exactly what the two separate entry points write. */
int kfx_depth_to_vbo_normals_f32(const kfx_image* vbo, const kfx_image* nrm, const kfx_image* depth, const float K[4], float scale,
                                 kfx_stream stream);

/* The frame's depth pyramid with its vertex and normal maps in one launch (no reference counterpart: the 2 * levels - 1 launches
 * of roo::BoxReduceIgnoreInvalid (reduce.h:48-59) and of DepthToVbo + NormalsFromVbo per level -- main.cpp:211-218 -- are
 * launch-latency sized).  depth[0] is the input; depth[1 .. levels) are written, each at most half the size of the level
 * before; vbo[l] / nrm[l] are the maps of level l (of depth[l]'s size); K holds `levels` x {fu, fv, u0, v0}.  1 <= levels <= 4.
 * Every output holds exactly what the per-level entry points write. */
int kfx_depth_pyramid_vbo_normals_f32(const kfx_image* depth, const kfx_image* vbo, const kfx_image* nrm, const float* K, int levels,
                                      float scale, kfx_stream stream);

/* (The operators of the reference's five headers that SURVEY.md 2 marks out of the path's scope -- the joint bilateral filter, Disp2Depth,
 * FilterBadKinectData, ColourVbo, TextureDepth, RaycastBox / Sphere / Plane, SdfDistance -- live in the same library behind
 * include/kfx_extras.h.) */

/* ---- colour fusion / colour raycast ------------------------------------------
 * `colorvol` is a roo::BoundedVolume<float> (4-byte grey cells in [0,1], same struct, same dims as `vol`);
 * `img` a roo::Image<uchar3> (3-byte pixels).
 * kfx_sdf_fuse_color: SdfFuse(vol, colorVol, depth, norm, T_cw, K, img, T_iw, Kimg, ...) (cu_sdffusion.cu:70-138) --
 *   voxels whose projections fall inside both images get the SDF update of kfx_sdf_fuse and
 *   colour = (w*c + colour*w_old) / (w + w_old), c = bilinear RGB mean / 255.  Extents as the reference's launch:
 *   x, y truncated to multiples of 16, every z slice (KFX_FUSE_FULL_EXTENT lifts the truncation).  Follows
 *   kfx_set_math_mode like kfx_sdf_fuse.
 * kfx_raycast_sdf_color: RaycastSdf(depth, norm, img, vol, colorVol, ...) (cu_raycast.cu:119-196) -- as
 *   kfx_raycast_sdf, but img = trilinear sample of the colour volume at the hit instead of the Phong shade.
 * kfx_color_reset: SdfReset(BoundedVolume<float>) = Fill(0.5) including pitch padding (cu_sdffusion.cu:166-169). */
int kfx_sdf_fuse_color(const kfx_volume* vol, const kfx_volume* colorvol, const kfx_image* depth, const kfx_image* norm,
                       const float T_cw[12], const float K[4], const kfx_image* img, const float T_iw[12], const float Kimg[4],
                       float trunc_dist, float max_w, float mincostheta, unsigned flags, kfx_stream stream);
int kfx_raycast_sdf_color(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, const kfx_volume* vol,
                          const kfx_volume* colorvol, const float T_wc[12], const float K[4], float near, float far,
                          float trunc_dist, int subpix, kfx_stream stream);
int kfx_color_reset(const kfx_volume* colorvol, kfx_stream stream);

/* ---- mesh extraction ----------------------------------------------------------
 * Marching cubes over the (w-1)(h-1)(d-1) cubes of a BoundedVolume<SDF_t>, as roo::SaveMesh performs it on the host
 * (MarchingCubes.h:43-143, loop nest :226-232), in two device passes:
 *   kfx_mc_count: counts[(x*(h-1) + y)*(d-1) + z] = triangles of cube (x,y,z) (0..5; 0 when a corner is not finite);
 *   kfx_mc_emit:  for the n_active cubes listed in cube_index (linear indices as above, ascending = emission order)
 *                 with tri_offset[i] = number of triangles emitted before cube i (exclusive prefix sum of counts at
 *                 that cube), writes for every triangle three vertices (verts / norms: 3 floats each; colors: 4
 *                 floats, only when `colorvol` has every dimension >= 8, the reference's IsValid) to slot
 *                 3*tri_offset[i] + ..., i.e. in the reference's emission order.
 * All pointers are device memory; the caller does the prefix sum and the compaction (kangaroo_amd/mesh.py uses
 * torch.cumsum / torch.nonzero, include/kangaroo/MarchingCubes.h a host loop over the one-byte counts). */
int kfx_mc_count(const kfx_volume* vol, unsigned char* counts, kfx_stream stream);
int kfx_mc_emit(const kfx_volume* vol, const kfx_volume* colorvol, const long long* cube_index, const unsigned* tri_offset,
                long long n_active, float* verts, float* norms, float* colors, kfx_stream stream);

/* ---- projective point-to-plane ICP ------------------------------------------
 * roo::LeastSquaresSystem<float,6> (Mat.h:483-520): JTy, the 21 unique elements of the symmetric JTJ in
 * row-major lower-triangle order (Mat.h:353-365), the squared error and the observation count. */
typedef struct kfx_lss6 {
    float JTy[6];
    float JTJ[21];
    float sqErr;
    unsigned obs;
} kfx_lss6;

/* roo::PoseRefinementProjectiveIcpPointPlane (cu_model_refinement.cu:541-608, decl cu_model_refinement.h:58-64).
 * For every pixel of the model-view vertex map dPr with a valid normal (dNr.w == 1): project it with KT_lr
 * into the live vertex map dPl (nearest neighbour, 3-pixel border), residual y = (T_rl * Pl - Pr) . Nr, Jacobian
 * -(gen_i * _Pr) . Nr, weight Tukey(y, c) / Pr.z; the per-pixel systems are summed (per 16x16 block in the
 * reference's tree order, then over the blocks in a fixed order) and returned in *out (host memory).
 * `workspace`: device bytes, pitch*h >= (w/gcd(w,16)) * (h/gcd(h,16)) * sizeof(kfx_lss6); `debug` (float4, may
 * be NULL) receives the reference's per-pixel debug colours.  Blocks until the result is in *out. */
int kfx_icp_point_plane(const kfx_image* Pl, const kfx_image* Pr, const kfx_image* Nr, const float KT_lr[12],
                        const float T_rl[12], float c, const kfx_image* workspace, const kfx_image* debug,
                        kfx_lss6* out, kfx_stream stream);

/* The whole coarse-to-fine refinement loop of the reference application (main.cpp:301-337) on the device: per level
 * (given COARSEST FIRST) and iteration the ICP system is summed, solved in float64 on the device (weak prior,
 * complete-pivoting LU spread over one wave, SE(3) exponential -- the algorithms of kangaroo_amd/tracking.py, the same operations in
 * the same order) and K*T_lp / T_lp^-1 are
 * left in device memory for the next evaluation; the host synchronises once at the end instead of once per
 * iteration.  `workspace`: >= (largest level's blocks * 116, rounded up to 256) + 512 bytes, 8-byte aligned.
 * Results: T_lp (row-major 3x4, float64), rmse / obs of the last evaluation, tracking_good = rmse < max_rmse. */
typedef struct kfx_icp_level {
    kfx_image Pl, Pr, Nr;  /* live vertex map, model vertex map, model normals of this level */
    float K[4];            /* the level's intrinsics fu, fv, u0, v0 */
    int iterations;
    int rotation_only;     /* the application solves the coarsest level for rotation only */
} kfx_icp_level;
int kfx_icp_refine(const kfx_icp_level* levels, int n_levels, float c, float max_rmse, const kfx_image* workspace,
                   const kfx_image* debug, double T_lp[12], float* rmse, unsigned* obs, int* tracking_good, kfx_stream stream);
/* kfx_icp_refine with a hook: `enqueue_more(user)` is called once, after the whole refinement and the read-back of its result
 * have been enqueued and before the calling thread waits -- for the read-back only, not for the stream.  Work the hook
 * enqueues on `stream` (the next frame's pre-amble: it does not depend on the pose) runs while the thread wakes up, so the
 * device is not idle between the refinement and what the caller launches with the pose (SdfFuse).  Same results. */
int kfx_icp_refine_then(const kfx_icp_level* levels, int n_levels, float c, float max_rmse, const kfx_image* workspace,
                        const kfx_image* debug, double T_lp[12], float* rmse, unsigned* obs, int* tracking_good,
                        void (*enqueue_more)(void* user), void* user, kfx_stream stream);

/* The pose update that follows the refinement (main.cpp:337 and :345): T_wl_out = T_wl * T_lp^-1 and, if T_cw_out is given,
 * (float) T_wl_out^-1 -- rigid row-major 3 x 4 transforms, float64 products written out term by term on the host (what SE3d of
 * apps/pose_solve.h computes).  T_wl_out may alias T_wl. */
int kfx_pose_step(const double T_wl[12], const double T_lp[12], double T_wl_out[12], float T_cw_out[12]);

/* ---- multi-GPU raycast composite (no reference counterpart) ------------------------
 * Per-pixel glue around the two collectives of kangaroo_amd/pipeline.py::SlabPipeline.composite:
 *   pack:   key[v*w+u] = (bits(depth or +inf) << 8) | rank                    then all_reduce(MIN, key)
 *   select: payload[(v*w+u)*4 ..] = this rank won ? {n.x,n.y,n.z, shade} : 0   then all_reduce(SUM, payload)
 *   unpack: depth = key's depth (NaN if no rank hit), norm = (payload xyz, hit ? 1 : 0), img = payload w
 * key (w*h int64) and payload (w*h*KFX_COMPOSITE_PAYLOAD float, 16-byte aligned) are dense device buffers owned by the caller. */
#define KFX_COMPOSITE_PAYLOAD 4
int kfx_composite_pack(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, long long* key, int rank, kfx_stream stream);
int kfx_composite_select(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, const long long* key,
                         float* payload, int rank, kfx_stream stream);
int kfx_composite_unpack(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, const long long* key,
                         const float* payload, kfx_stream stream);

/* The direct-send composite (xGMI is a full mesh of point-to-point links: every pair of GPUs has its own).  Rank j owns strip j of
 * the image -- pixels [j S, (j + 1) S) in row-major order, S = kfx_composite_strip_pixels(w, h, world) (the last strip is padded) --
 * and a buffer of strips is world x KFX_COMPOSITE_STRIP_PLANES x S floats: per strip the planes {depth (+inf: miss), n.x, n.y, n.z,
 * shade}.
 *   pack:   send = this rank's images cut into strips           then all-to-all: strip j goes to rank j (recv[r] = rank r's copy)
 *   merge:  merged (KFX_COMPOSITE_STRIP_PLANES x S) = per pixel the copy with the nearest depth, the lowest rank on ties -- the
 *           winner of the key's minimum above                   then all-gather of `merged` (or a gather to one rank)
 *   unpack: the gathered strips back into depth (NaN: miss) / norm (w = hit ? 1 : 0) / img
 * Per rank and phase 20 B x (world - 1) / world x w h bytes leave over world - 1 links at once.  rank_stride: floats between rank r's
 * strip and rank r + 1's in a buffer of strips (0: dense, KFX_COMPOSITE_STRIP_PLANES x S) -- several images (pyramid levels) can
 * share one buffer, and one pair of collectives, with the strips of a rank side by side. */
#define KFX_COMPOSITE_STRIP_PLANES 5
size_t kfx_composite_strip_pixels(size_t w, size_t h, int world);
int kfx_composite_strips_pack(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, float* send, size_t rank_stride, int world,
                              kfx_stream stream);
int kfx_composite_strips_merge(const float* recv, float* merged, size_t strip_pixels, size_t rank_stride, int world, kfx_stream stream);
int kfx_composite_strips_unpack(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, const float* strips, size_t rank_stride,
                                int world, kfx_stream stream);

/* Exact multi-GPU march ("exact variant").  One round of one rank: rays whose current sample
 * falls into a trilinear base cell this rank owns, [own_lo, own_hi) (global plane indices), are advanced until
 * they hit, leave the volume, or step into another rank's cells; lambda / last_sdf / delta travel in `state`
 * (KFX_RAY_STATE_PLANES dense planes of h*w floats: 0 lambda, 1 last_sdf, 2 delta, 3 status {0 marching, 1 hit,
 * 2 miss, 3 hit awaiting its normal}, 4 touched in this call, 5-7 normal, 8 shade; a hit's depth is its lambda).  `vol` holds planes [slab->z_offset, slab->z_offset + vol->d) of the
 * full volume described by `slab`.  Every sample is taken at the position and from the cells of the
 * single-volume march (cu_raycast.cu:58-81), so the final images are bit-identical to kfx_raycast_sdf on the
 * whole volume.  The host merges the per-rank states between rounds (kangaroo_amd/pipeline.py). */
#define KFX_RAY_STATE_PLANES 9
int kfx_raycast_sdf_slab(float* state, int init, const kfx_volume* vol, const kfx_slab* slab, int own_lo, int own_hi,
                         int w, int h, const float T_wc[12], const float K[4], float near, float far,
                         float trunc_dist, int subpix, kfx_stream stream);
/* The same round on a range of image rows [v0, v1), with the state kept per row-tile (the tile-pipelined hand-over of
 * kfx_slab_raycast_exact_tiled, kfx_slab.h): tile t = rows [t R, (t + 1) R), R = rows_per_tile; its march planes 0-3 lie at
 * state + (t * 4 + k) * plane_stride, its normal / shade planes at result + (t * 4 + k) * plane_stride (plane_stride >= R w
 * pixels), so that the march state of one tile -- 16 bytes per pixel; plane 4 is not kept -- is one contiguous message.  fin (optional, dense w h ints): set to 1
 * where this call gives a ray its final status; an initialising call clears it first, and with claim_misses sets it for rays that
 * never enter the box (one rank answers for those).  adopt_lo / adopt_hi (optional): snapshots of the same rays received from
 * the two neighbour ranks, march planes as above -- of ONE tile ([4][plane_stride], the tile of rows v0 .. v1) or, with
 * layout_flags & 1, of all tiles: before marching, a ray takes a neighbour's snapshot when it is newer than its own (a final or
 * hit status is later than "marching", a larger lambda later than a smaller) and still under way.  layout_flags & 2: the tile state
 * is PACKED into three planes -- lambda, last_sdf, and a marching ray's delta (positive) or -status of any other ray -- at
 * state + (t * 3 + k) * plane_stride, the snapshots alike: 12 bytes per ray and hop; needs trunc_dist > 0.
 * layout_flags & 4: a hit's normal is evaluated wherever the three planes of its gradient stencil are stored (ghost planes included),
 * not only by the rank that owns the stencil's base plane. */
int kfx_raycast_sdf_slab_tiles(float* state, float* result, size_t plane_stride, int rows_per_tile, int v0, int v1, int init, int* fin,
                               int claim_misses, const float* adopt_lo, const float* adopt_hi, int layout_flags, const kfx_volume* vol,
                               const kfx_slab* slab, int own_lo, int own_hi, int w, int h, const float T_wc[12], const float K[4], float near,
                               float far, float trunc_dist, int subpix, kfx_stream stream);
/* fp16-cell (roo::SDF_h) variants of the slab entry points: config C5 spread over several GPUs */
int kfx_sdf_fuse_slab_h(const kfx_volume* vol, const kfx_slab* slab, const kfx_image* depth, const kfx_image* norm,
                        const float T_cw[12], const float K[4], float trunc_dist, float max_w, float mincostheta,
                        unsigned flags, kfx_stream stream);
int kfx_raycast_sdf_slab_h(float* state, int init, const kfx_volume* vol, const kfx_slab* slab, int own_lo, int own_hi,
                           int w, int h, const float T_wc[12], const float K[4], float near, float far,
                           float trunc_dist, int subpix, kfx_stream stream);
int kfx_raycast_state_to_images(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, const float* state,
                                kfx_stream stream);

/* ---- brick summary: RaycastSdf without touching uniformly free or never-observed space (addition) ---------------
 * The reference's march (cu_raycast.cu:58-81) reads eight cells at every step, ~100 dependent HBM misses per ray, most
 * of them in space the TSDF knows to be empty.  A kfx_sdf_summary keeps, per 8 x 8 x 8 cells, the range of the stored
 * values; the tracked SdfFuse maintains it as a by-product (one workgroup owns a summary brick: wave shuffles + LDS, no
 * atomics).  The tracked RaycastSdf builds two-bit class tables from it (per 16^3 and 32^3 cells: every cell holds trunc_dist
 * / every cell is NaN / every cell is one or the other), stages them in LDS, derives the same classes for 64^3 and 128^3 cells
 * there, and takes the reference's own steps through such entries without reading the volume:
 *   exact numerics: only cells bit-equal to trunc_dist (or NaN) qualify -- the images equal kfx_raycast_sdf bit for bit;
 *   fast numerics:  also cells within a relative 1e-5 of it (observed free space: the running average of +trunc drifts
 *                   by a few ulp per frame) -- depth within the fast-mode tolerance of the exact march.
 * Where less than a quarter of the volume qualifies (the count the last table build published; KFX_RAYCAST_SUMMARY=1 / -1
 * overrides) the tracked call runs the plain march: same images.
 * The summary describes the volume it was created for; views of that volume (SubBoundingVolume) may be passed to the
 * tracked calls.  Anything else that writes the volume (copies, kfx_sdf_sphere, untracked kfx_sdf_fuse) must be followed
 * by kfx_sdf_summary_invalidate.  A tracked SdfFuse whose view does not start on multiples of 8 cells, or that takes the
 * untiled kernel, invalidates the summary itself (correct, no skipping until the next reset). */
typedef struct kfx_sdf_summary kfx_sdf_summary;
int kfx_sdf_summary_create(kfx_sdf_summary** out, const kfx_volume* vol);
int kfx_sdf_summary_destroy(kfx_sdf_summary* s);
int kfx_sdf_summary_invalidate(kfx_sdf_summary* s, kfx_stream stream);
int kfx_sdf_reset_tracked(const kfx_volume* vol, kfx_sdf_summary* s, float trunc_dist, kfx_stream stream);
int kfx_sdf_fuse_tracked(const kfx_volume* vol, kfx_sdf_summary* s, const kfx_image* depth, const kfx_image* norm,
                         const float T_cw[12], const float K[4], float trunc_dist, float max_w, float mincostheta,
                         unsigned flags, kfx_stream stream);
int kfx_raycast_sdf_tracked(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, const kfx_volume* vol,
                            kfx_sdf_summary* s, const float T_wc[12], const float K[4], float near, float far,
                            float trunc_dist, int subpix, kfx_stream stream);
/* Diagnostics: kfx_raycast_sdf_count for the march kfx_raycast_sdf_tracked would run (the class-table march, or the plain one
 * where the tracked call falls back to it).  d_counters[6], added to: samples taken, rays that enter the box, hits, distinct
 * voxels read (U), table look-ups, bytes of class tables a workgroup stages (0: plain march).  bench.py prices the table
 * march's algorithmic bytes with it: 8 B x U + table bytes + 24 B x w h. */
int kfx_raycast_sdf_count_tracked(const kfx_volume* vol, kfx_sdf_summary* s, unsigned w, unsigned h, const float T_wc[12], const float K[4],
                                  float near, float far, float trunc_dist, int subpix, unsigned* d_bitmap, unsigned long long* d_counters,
                                  kfx_stream stream);
/* kfx_raycast_sdf_levels (several renderings of the model in one launch) with the summary consulted in every march */
int kfx_raycast_sdf_levels_tracked(int n_levels, const kfx_image* const* depth, const kfx_image* const* norm, const kfx_image* const* img,
                                   const kfx_image* const* vbo, const kfx_volume* vol, kfx_sdf_summary* s, const float T_wc[12],
                                   const float* K, float near, float far, float trunc_dist, int subpix, kfx_stream stream);

/* Recompute the summary from what the volume holds (one pass over the parent volume: 8 B x cells of reads): the way back to a
 * summary that describes the volume after writers that do not track (untracked kfx_sdf_fuse, copies, LoadPXM, kfx_sdf_sphere)
 * -- every brick gets the exact range of its valued cells and the exact state, instead of kfx_sdf_summary_invalidate's
 * "nothing is known".  fp32 cells. */
int kfx_sdf_summary_rebuild(kfx_sdf_summary* s, kfx_stream stream);

/* ---- one frame of the application's loop as ONE call (no reference counterpart in the operator API) ----------------
 * The per-frame sequence of applications/kinectfusion/main.cpp:200-356 for a stream with known poses, enqueued by the
 * library itself: BilateralFilter(filtered, raw) -> DepthToVbo(vbo, filtered, K) -> NormalsFromVbo(normals, vbo)
 * (main.cpp:209-215) -> SdfFuse(vol, filtered, normals, T_cw, ...) (main.cpp:345-356) -> RaycastSdf(ray_*, vol, T_wc, ...)
 * (main.cpp:286) -- exactly the launches the separate entry points above make, in that order on `stream`, so the images and
 * the volume are bit-identical to calling them one by one; what it saves is the host time between launches (a frame is
 * ~0.45 ms of GPU work at 512^3: an interpreter that issues seven calls per frame can fall behind it).
 * A kfx_frame owns nothing but its optional brick summary and its timing events: volume and images are the caller's views.
 *   kfx_frame_set_track(f, 1): SdfFuse / RaycastSdf go through kfx_sdf_fuse_tracked / kfx_raycast_sdf_tracked with a summary
 *     the frame creates on first use and (re)builds from the volume's contents (kfx_sdf_summary_rebuild); 0: the plain pair.
 *   kfx_frame_step(f, raw, T_wc, T_cw, parts, stream): raw NULL = cfg.raw; T_cw NULL = inverse of T_wc (R^T, -R^T t in double,
 *     rounded once); parts = KFX_FRAME_* bits, 0 = all.
 *   Timing: with cfg.timing_slots = n > 0 every step records four device events on `stream` (before the preprocess, before
 *     SdfFuse, after SdfFuse, after RaycastSdf) in a ring of the last n frames.  kfx_frame_timings waits for the last of the
 *     frames asked for and returns KFX_FRAME_TIMING_FIELDS floats per frame, in ms: preprocess, SdfFuse, RaycastSdf (tracked:
 *     the class-table build included), the whole frame, and the period = start of this frame to the start of the next one (NaN
 *     for the most recent frame; measured between the first event a frame records and the same event of the next frame) -- what a
 *     frames/s figure is made of.  Frames that have left the ring: KFX_E_RANGE. */
typedef struct kfx_frame kfx_frame;
typedef struct kfx_frame_config {
    kfx_volume vol;                          /* BoundedVolume<SDF_t> (fp32 cells) */
    kfx_image raw, filtered, vbo, normals;   /* depth in (metres), BilateralFilter out, DepthToVbo out, NormalsFromVbo out */
    kfx_image ray_depth, ray_norm, ray_img;  /* RaycastSdf outputs */
    float K[4];
    float bilateral_gs, bilateral_gr, bilateral_minval;   /* main.cpp:149-151,209 */
    unsigned bilateral_size;
    float near, far, trunc_dist, max_w, mincostheta;      /* main.cpp:80-81,155-158,221 */
    unsigned fuse_flags;                     /* KFX_FUSE_* */
    int timing_slots;                        /* 0: no events */
} kfx_frame_config;
#define KFX_FRAME_PREPROCESS 1u
#define KFX_FRAME_FUSE       2u
#define KFX_FRAME_RAYCAST    4u
#define KFX_FRAME_TIMING_FIELDS 5
#define KFX_FRAME_EVENTS_ALL  15u  /* before the preprocess, before SdfFuse, after SdfFuse, after RaycastSdf */
#define KFX_FRAME_EVENTS_FUSE  6u  /* the two around SdfFuse: its window and the frame period (before-SdfFuse to before-SdfFuse) */
int kfx_frame_create(kfx_frame** out, const kfx_frame_config* cfg);
int kfx_frame_destroy(kfx_frame* f);
int kfx_frame_reset(kfx_frame* f, kfx_stream stream);             /* SdfReset(vol, NaN) (main.cpp:229), summary set to match */
int kfx_frame_set_track(kfx_frame* f, int on, kfx_stream stream);
int kfx_frame_get_track(const kfx_frame* f);
kfx_sdf_summary* kfx_frame_summary(kfx_frame* f);                 /* the frame's own summary (NULL before the first set_track(1)) */
long long kfx_frame_count(const kfx_frame* f);                    /* frames stepped so far = index of the next frame */
int kfx_frame_step(kfx_frame* f, const kfx_image* raw, const float T_wc[12], const float* T_cw, unsigned parts, kfx_stream stream);
int kfx_frame_timings(kfx_frame* f, long long first_frame, int n_frames, float* ms);
/* Which of the four events the following steps record (KFX_FRAME_EVENTS_*; 0: none).  An event is a marker between two launches
 * of the stream and is not free: all four cost 2.7 % of a 0.42 ms frame (measured), so a loop that is itself being timed records
 * the two around SdfFuse only.  Fields that need an event the frame did not record come back as NaN. */
int kfx_frame_set_timing(kfx_frame* f, unsigned mask);

/* ---- numerics mode --------------------------------------------------------------- */
/* KFX_MATH_EXACT (default): IEEE fp32, no FMA contraction, correctly rounded div/sqrt, reference
 * operation order -- bit-identical to the CPU oracle.  KFX_MATH_FAST: hardware rcp/rsq (1 ulp),
 * FMA, shared reciprocals -- the regime of the reference's own build (-use_fast_math,
 * CMakeLists.txt:141); within the stated tolerance (TSDF L-inf < 1e-4), not bit-exact.  The running
 * average of fp32 cells (Sdf.h:25-32) is evaluated as old + (new - old) * (w / (w + old.w)): the same
 * value up to rounding, and a cell that is handed the value it holds keeps it bit for bit (free space
 * stays at +trunc over any number of frames, which the class tables of kfx_sdf_summary rely on).
 * Process-global; initial value from the environment variable KFX_MATH=exact|fast.
 * Currently affects kfx_sdf_fuse and kfx_raycast_sdf. */
#define KFX_MATH_EXACT 0
#define KFX_MATH_FAST  1
int kfx_set_math_mode(int mode); /* returns the previous mode, or KFX_E_RANGE */
int kfx_get_math_mode(void);

/* ---- diagnostics --------------------------------------------------------------- */
const char* kfx_last_error_string(void); /* thread-local, never NULL */
const char* kfx_error_name(int code);    /* hipGetErrorString for >0, KFX_E_* names for <0 */
int kfx_version(void);                   /* major*100 + minor */
/* A digest of the sources the kernels of a family ("fuse": SdfFuse, "raycast": RaycastSdf and the class tables) were compiled
 * from ("" for an unknown family).  Committed measurements of a kernel (profiles/..._pmc_traffic.json) carry it; bench.py reports
 * a counter figure only when it was taken on the kernels it is running. */
const char* kfx_kernel_source_id(const char* family);
int kfx_device_count(void);
int kfx_set_device(int device);          /* hipSetDevice for the calling thread (one rank per GPU: include/kfx_slab.h) */

#ifdef __cplusplus
}
#endif
#endif /* KFX_H */
