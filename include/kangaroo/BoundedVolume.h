// BoundedVolume.h -- a Volume plus the world-space box it spans, roo::BoundedVolume<T,...>
// (reference include/kangaroo/BoundedVolume.h:10-170), 72 bytes = kfx_volume.
// The grid is vertex-centred: voxel i sits at min + size*i/(dim-1), the first and last voxels lie on
// the box faces (VoxelPositionInUnits :115-125); SubBoundingVolume (:137-165) is how the application
// restricts fuse / raycast to the part of the volume in view.
#pragma once

#include <kangaroo/BoundingBox.h>
#include <kangaroo/Volume.h>

namespace roo
{

template<typename T, typename Target = TargetDevice, typename Management = DontManage>
class BoundedVolume : public Volume<T, Target, Management>
{
    typedef Volume<T, Target, Management> Base;

public:
    template<typename TargetFrom, typename ManagementFrom>
    KANGAROO_HD BoundedVolume(const BoundedVolume<T, TargetFrom, ManagementFrom>& vol) : Base(vol), bbox(vol.bbox) {}
    template<typename TargetFrom, typename ManagementFrom>
    KANGAROO_HD BoundedVolume(const Volume<T, TargetFrom, ManagementFrom>& vol, const BoundingBox& box) : Base(vol), bbox(box) {}
    KANGAROO_HD BoundedVolume() {}
    BoundedVolume(unsigned int w, unsigned int h, unsigned int d)
        : Base(w, h, d), bbox(make_float3(-1, -1, -1), make_float3(1, 1, 1)) {}
    BoundedVolume(unsigned int w, unsigned int h, unsigned int d, const BoundingBox& box) : Base(w, h, d), bbox(box) {}
    BoundedVolume(unsigned int w, unsigned int h, unsigned int d, float3 min_bounds, float3 max_bounds)
        : Base(w, h, d), bbox(min_bounds, max_bounds) {}

    // the C-ABI view (same 72 bytes)
    const kfx_volume* abi() const
    {
        static_assert(sizeof(BoundedVolume<T, Target, Management>) == sizeof(kfx_volume), "roo::BoundedVolume must mirror kfx_volume");
        return reinterpret_cast<const kfx_volume*>(this);
    }

    KANGAROO_HD float3 SizeUnits() const { return bbox.Size(); }
    KANGAROO_HD float3 VoxelSizeUnits() const
    {
        return div_cw(bbox.Size(), make_float3(Base::w - 1, Base::h - 1, Base::d - 1));
    }
    // a volume with fewer than 8 voxels along an axis is not worth launching on
    KANGAROO_HD bool IsValid() const { return Base::w >= 8 && Base::h >= 8 && Base::d >= 8; }

    KANGAROO_HD float GetUnitsTrilinearClamped(float3 pos_w) const
    {
        return Base::GetFractionalTrilinearClamped(div_cw(sub(pos_w, bbox.Min()), bbox.Size()));
    }
    KANGAROO_HD float3 GetUnitsBackwardDiffDxDyDz(float3 pos_w) const
    {
        const float3 deriv = Base::GetFractionalBackwardDiffDxDyDz(div_cw(sub(pos_w, bbox.Min()), bbox.Size()));
        return div_cw(deriv, VoxelSizeUnits());
    }
    KANGAROO_HD float3 GetUnitsOutwardNormal(float3 pos_w) const
    {
        const float3 deriv = GetUnitsBackwardDiffDxDyDz(pos_w);
        return div_by(deriv, length(deriv));
    }

    KANGAROO_HD float3 VoxelPositionInUnits(int x, int y, int z) const
    {
        const float3 s = bbox.Size();
        return make_float3(bbox.Min().x + s.x * x / (float)(Base::w - 1), bbox.Min().y + s.y * y / (float)(Base::h - 1),
                           bbox.Min().z + s.z * z / (float)(Base::d - 1));
    }
    KANGAROO_HD float3 VoxelPositionInUnits(int3 p) const { return VoxelPositionInUnits(p.x, p.y, p.z); }

    // view of the voxels covering `region`, with the box recomputed from the voxel positions
    KANGAROO_HD BoundedVolume<T, Target, DontManage> SubBoundingVolume(const BoundingBox& region)
    {
        const float3 lo = div_cw(sub(region.Min(), bbox.Min()), bbox.Size());
        const float3 hi = div_cw(sub(region.Max(), bbox.Min()), bbox.Size());
        const int3 min_v = make_int3(fmaxf((Base::w - 1) * lo.x, 0), fmaxf((Base::h - 1) * lo.y, 0), fmaxf((Base::d - 1) * lo.z, 0));
        const int3 max_v = make_int3(fminf(ceilf((Base::w - 1) * hi.x), Base::w - 1), fminf(ceilf((Base::h - 1) * hi.y), Base::h - 1),
                                     fminf(ceilf((Base::d - 1) * hi.z), Base::d - 1));
        const int3 size_v = make_int3(std::max(max_v.x - min_v.x + 1, 0), std::max(max_v.y - min_v.y + 1, 0), std::max(max_v.z - min_v.z + 1, 0));
        const BoundingBox nbox(VoxelPositionInUnits(min_v), VoxelPositionInUnits(max_v));
        return BoundedVolume<T, Target, DontManage>(Base::SubVolume(min_v, size_v), nbox);
    }

    BoundingBox bbox;
};

}
