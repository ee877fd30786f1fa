// cu_normals.h -- roo::NormalsFromVbo with the reference's signature (include/kangaroo/cu_normals.h:9-10).
#pragma once

#include <kangaroo/Image.h>
#include <kangaroo/launch_utils.h>
#include <kangaroo/platform.h>

namespace roo
{

KANGAROO_EXPORT inline
void NormalsFromVbo(Image<float4> dN, const Image<float4> dV)
{
    GpuNoteStatus(kfx_normals_from_vbo(dN.abi(), dV.abi(), 0));
}

}
