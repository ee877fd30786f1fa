// reduce.h -- pyramid builders (reference include/kangaroo/reduce.h:48-59): BoxReduceIgnoreInvalid fills
// levels 1.. of a pyramid from level 0 with the NaN-aware 2x2 mean.
#pragma once

#include <kangaroo/Pyramid.h>
#include <kangaroo/cu_resample.h>

namespace roo
{

// Level l is the NaN-aware 2 x 2 mean of level l - 1; levels whose source would be narrower or lower than two pixels keep
// their contents (the level's size is (w >> l) x (h >> l), and a size of zero ends the chain).
template<typename T, unsigned Levels, typename UpType>
inline void BoxReduceIgnoreInvalid(Pyramid<T,Levels> pyramid)
{
    Image<T>* finer = &pyramid.imgs[0];
    size_t lw = finer->w >> 1, lh = finer->h >> 1;
    for (unsigned level = 1; level < Levels && lw != 0 && lh != 0; ++level, lw >>= 1, lh >>= 1) {
        Image<T>* coarser = &pyramid.imgs[level];
        BoxHalfIgnoreInvalid<T,UpType,T>(*coarser, *finer);
        finer = coarser;
    }
}

}
