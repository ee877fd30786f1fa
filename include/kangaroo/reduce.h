// reduce.h -- pyramid builders (reference include/kangaroo/reduce.h:48-59): BoxReduceIgnoreInvalid fills
// levels 1.. of a pyramid from level 0 with the NaN-aware 2x2 mean.
#pragma once

#include <kangaroo/Pyramid.h>
#include <kangaroo/cu_resample.h>

namespace roo
{

template<typename T, unsigned Levels, typename UpType>
inline void BoxReduceIgnoreInvalid(Pyramid<T,Levels> pyramid)
{
    const int w = pyramid.imgs[0].w;
    const int h = pyramid.imgs[0].h;
    for(unsigned int l=1; l<Levels && (w>>l > 0) && (h>>l > 0); ++l) {
        BoxHalfIgnoreInvalid<T,UpType,T>(pyramid.imgs[l], pyramid.imgs[l-1]);
    }
}

}
