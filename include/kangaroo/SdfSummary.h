// SdfSummary.h -- the brick summary of a BoundedVolume<SDF_t> for C++ hosts (kfx_sdf_summary, include/kfx.h).  Addition
// beside the reference API: SdfFuse / SdfReset overloads that keep the summary current and a RaycastSdf overload that steps
// through uniformly free or never-observed regions without reading the volume.  Exact numerics: images bit-identical to
// roo::RaycastSdf; fast numerics: within the fast-mode tolerance.  Include it explicitly.
//
//   roo::SdfSummary summary(vol);                       // vol: the whole volume (views of it may be passed below)
//   roo::SdfReset(vol, NaN, summary);
//   roo::SdfFuse(work_vol, summary, depth, normals, T_cw, K, trunc, max_w, mincostheta);
//   roo::RaycastSdf(d, n, i, work_vol, summary, T_wc, K, near, far, trunc, true);
#pragma once

#include <kangaroo/BoundedVolume.h>
#include <kangaroo/Image.h>
#include <kangaroo/ImageIntrinsics.h>
#include <kangaroo/Mat.h>
#include <kangaroo/Sdf.h>
#include <kangaroo/launch_utils.h>

namespace roo
{

class SdfSummary
{
public:
    template<typename Management>
    explicit SdfSummary(const BoundedVolume<SDF_t, TargetDevice, Management>& vol) : handle_(0)
    {
        GpuCheckStatus(kfx_sdf_summary_create(&handle_, vol.abi()));
    }
    ~SdfSummary() { kfx_sdf_summary_destroy(handle_); }
    SdfSummary(const SdfSummary&) = delete;
    SdfSummary& operator=(const SdfSummary&) = delete;
    // after the volume was written by anything but the overloads below
    void Invalidate() { GpuCheckStatus(kfx_sdf_summary_invalidate(handle_, 0)); }
    // ... or recompute it from what the volume holds (one pass over the volume): exact ranges and states for every brick
    void Rebuild() { GpuCheckStatus(kfx_sdf_summary_rebuild(handle_, 0)); }
    kfx_sdf_summary* get() const { return handle_; }

private:
    kfx_sdf_summary* handle_;
};

inline void SdfReset(BoundedVolume<SDF_t> vol, float trunc_dist, SdfSummary& summary)
{
    GpuCheckStatus(kfx_sdf_reset_tracked(vol.abi(), summary.get(), trunc_dist, 0));
}

inline void SdfFuse(BoundedVolume<SDF_t> vol, SdfSummary& summary, Image<float> depth, Image<float4> norm, Mat<float,3,4> T_cw, ImageIntrinsics K,
                    float trunc_dist, float maxw, float mincostheta)
{
    GpuCheckStatus(kfx_sdf_fuse_tracked(vol.abi(), summary.get(), depth.abi(), norm.abi(), T_cw.m, &K.fu, trunc_dist, maxw, mincostheta, 0, 0));
}

inline void RaycastSdf(Image<float> depth, Image<float4> norm, Image<float> img, const BoundedVolume<SDF_t> vol, SdfSummary& summary,
                       const Mat<float,3,4> T_wc, ImageIntrinsics K, float near, float far, float trunc_dist, bool subpix = true)
{
    GpuCheckStatus(kfx_raycast_sdf_tracked(depth.abi(), norm.abi(), img.abi(), vol.abi(), summary.get(), T_wc.m, &K.fu, near, far, trunc_dist,
                                           subpix ? 1 : 0, 0));
}

// RaycastSdfLevels (cu_raycast.h: several renderings of the model in one launch) with the summary consulted in every march
inline void RaycastSdfLevels(const Image<float>* depth, const Image<float4>* norm, const Image<float>* img, unsigned n,
                             const BoundedVolume<SDF_t> vol, SdfSummary& summary, const Mat<float,3,4> T_wc, const ImageIntrinsics* K,
                             float near, float far, float trunc_dist, bool subpix = true, const Image<float4>* vbo = 0)
{
    const kfx_image *d[8], *nn[8], *im[8], *vb[8];
    float k[32];
    if (n > 8) GpuCheckStatus(KFX_E_RANGE);
    for (unsigned l = 0; l < n && l < 8; ++l) {
        d[l] = depth[l].abi(); nn[l] = norm[l].abi(); im[l] = img[l].abi();
        vb[l] = vbo ? vbo[l].abi() : 0;
        k[4 * l] = K[l].fu; k[4 * l + 1] = K[l].fv; k[4 * l + 2] = K[l].u0; k[4 * l + 3] = K[l].v0;
    }
    GpuCheckStatus(kfx_raycast_sdf_levels_tracked((int)n, d, nn, im, vbo ? vb : 0, vol.abi(), summary.get(), T_wc.m, k, near, far, trunc_dist,
                                                  subpix ? 1 : 0, 0));
}

}
