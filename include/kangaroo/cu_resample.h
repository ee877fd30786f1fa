// cu_resample.h -- roo::BoxHalfIgnoreInvalid<To,UpType,Ti> (reference include/kangaroo/cu_resample.h,
// src/cu_resample.cu:89-120): 2x2 mean over the valid (finite) samples, invalid if there are none.
#pragma once

#include <kangaroo/Image.h>
#include <kangaroo/launch_utils.h>
#include <kangaroo/platform.h>

namespace roo
{

template<typename To, typename UpType, typename Ti>
KANGAROO_EXPORT
void BoxHalfIgnoreInvalid( Image<To> out, const Image<Ti> in);

template<> inline void BoxHalfIgnoreInvalid<float,float,float>( Image<float> out, const Image<float> in)
{
    GpuNoteStatus(kfx_box_half_ignore_invalid_f32(out.abi(), in.abi(), 0));
}

}
