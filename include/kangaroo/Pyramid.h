// Pyramid.h -- power-of-two image pyramid, roo::Pyramid<T, Levels, Target, Management>
// (reference include/kangaroo/Pyramid.h:9-137): level l is (w >> l) x (h >> l); each level is an
// ordinary roo::Image that cleans up after itself.
#pragma once

#include <kangaroo/Image.h>

namespace roo
{

template<typename T, unsigned Levels, typename Target = TargetDevice, typename Management = DontManage>
struct Pyramid
{
    KANGAROO_HD ~Pyramid() {}

    Pyramid() {}
    Pyramid(unsigned w, unsigned h)
    {
        Management::AllocateCheck();
        for (unsigned l = 0; l < Levels && (w >> l) > 0 && (h >> l) > 0; ++l) {
            Image<T, Target, Management> level(w >> l, h >> l);
            imgs[l].Swap(level);
        }
    }
    template<typename TargetFrom, typename ManagementFrom>
    KANGAROO_HD Pyramid(const Pyramid<T, Levels, TargetFrom, ManagementFrom>& p)
    {
        AssignmentCheck<Management, Target, TargetFrom>();
        for (unsigned l = 0; l < Levels; ++l) {
            imgs[l].pitch = p.imgs[l].pitch; imgs[l].ptr = p.imgs[l].ptr;
            imgs[l].w = p.imgs[l].w; imgs[l].h = p.imgs[l].h;
        }
    }

    template<typename TargetFrom, typename ManagementFrom>
    void CopyFrom(const Pyramid<T, Levels, TargetFrom, ManagementFrom>& p)
    {
        for (unsigned l = 0; l < Levels; ++l) imgs[l].CopyFrom(p.imgs[l]);
    }
    KANGAROO_HD void Swap(Pyramid<T, Levels, Target, Management>& p)
    {
        for (unsigned l = 0; l < Levels; ++l) imgs[l].Swap(p.imgs[l]);
    }

    KANGAROO_HD Image<T, Target, Management>& operator[](size_t l) { return imgs[l]; }
    KANGAROO_HD const Image<T, Target, Management>& operator[](size_t l) const { return imgs[l]; }
    KANGAROO_HD Image<T, Target, Management>& operator()(size_t l) { return imgs[l]; }
    KANGAROO_HD const Image<T, Target, Management>& operator()(size_t l) const { return imgs[l]; }

    template<unsigned SubLevels> KANGAROO_HD Pyramid<T, SubLevels, Target, DontManage> SubPyramid(unsigned start)
    {
        Pyramid<T, SubLevels, Target, DontManage> sub;
        for (unsigned l = 0; l < SubLevels && start + l < Levels; ++l) {
            sub.imgs[l].pitch = imgs[start + l].pitch; sub.imgs[l].ptr = imgs[start + l].ptr;
            sub.imgs[l].w = imgs[start + l].w; sub.imgs[l].h = imgs[start + l].h;
        }
        return sub;
    }

    Image<T, Target, Management> imgs[Levels];
};

}
