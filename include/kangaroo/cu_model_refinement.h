// cu_model_refinement.h -- roo::PoseRefinementProjectiveIcpPointPlane with the reference's signature
// (include/kangaroo/cu_model_refinement.h:58-64; kernel src/cu_model_refinement.cu:541-608).  SURVEY 8(f) f-2.
#pragma once

#include <kangaroo/Image.h>
#include <kangaroo/Mat.h>
#include <kangaroo/launch_utils.h>
#include <kangaroo/platform.h>

namespace roo
{

static_assert(sizeof(LeastSquaresSystem<float,6>) == sizeof(kfx_lss6), "LeastSquaresSystem<float,6> must match kfx_lss6");

// dPl: live vertex map; dPr / dNr: model vertex map / normals (RaycastSdf + DepthToVbo); KT_lr = K * T_lr projects
// model points into the live image, T_rl maps live points into the model frame; c: Tukey cut-off (metres);
// dWorkspace: >= gridDim.x * gridDim.y * sizeof(LeastSquaresSystem<float,6>) device bytes; dDebug: per-pixel status.
// Blocks until the summed system is on the host (as the reference's thrust::reduce does).
KANGAROO_EXPORT inline
LeastSquaresSystem<float,6> PoseRefinementProjectiveIcpPointPlane(
    const Image<float4> dPl,
    const Image<float4> dPr, const Image<float4> dNr,
    const Mat<float,3,4> KT_lr, const Mat<float,3,4> T_rl, float c,
    Image<unsigned char> dWorkspace, Image<float4> dDebug
)
{
    LeastSquaresSystem<float,6> lss;
    GpuCheckStatus(kfx_icp_point_plane(dPl.abi(), dPr.abi(), dNr.abi(), KT_lr.m, T_rl.m, c, dWorkspace.abi(), dDebug.abi(),
                                       reinterpret_cast<kfx_lss6*>(&lss), 0));
    return lss;
}

}
