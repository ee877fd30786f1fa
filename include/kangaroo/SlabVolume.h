// SlabVolume.h -- a BoundedVolume<SDF_t> spread over the ranks of a node in Z-slabs, for C++ hosts (one rank per GPU, or
// one rank per host thread with the in-process transport).  Addition beside the reference API (the reference is
// single-GPU): thin wrappers over include/kfx_slab.h in the style of the roo:: operators -- same container types, same
// error convention.  Include it explicitly; <kangaroo/kangaroo.h> stays the reference's surface.
//
//   kfx_comm comm;  kfx_comm_create_rccl(&comm, rank, world, "/tmp/kfx.id", 60);        // or kfx_comm_create_threads
//   roo::SlabVolume slab(512, 512, 512, bbox, &comm);                                    // allocates this rank's planes
//   roo::SdfReset(slab.local, NaN);
//   per frame:  slab.Fuse(depth, normals, T_cw, K, trunc, max_w, mincostheta);           // + halo exchange
//               slab.Raycast(d, n, i, T_wc, K, near, far, trunc);                        // identical images on every rank
#pragma once

#include <kfx_slab.h>

#include <kangaroo/BoundedVolume.h>
#include <kangaroo/Image.h>
#include <kangaroo/ImageIntrinsics.h>
#include <kangaroo/Mat.h>
#include <kangaroo/Sdf.h>
#include <kangaroo/cu_raycast.h>
#include <kangaroo/launch_utils.h>

namespace roo
{

class SlabVolume
{
public:
    enum HaloMode { HaloExchange, HaloRecompute };   // ghost planes: from the neighbours (RCCL send / recv) or integrated redundantly
    // nearest hit of per-slab marches; the march state handed from slab to slab (neighbour exchanges, one synchronisation per
    // frame); the same march with an all-reduce and a host check per round (the cross-check of the hand-over)
    enum RaycastMode { Composite, Exact, ExactAllReduce };
    // Composite: image strips sent to their owners and back (all-to-all + all-gather over the xGMI mesh), or two all-reduces
    enum MergeMode { MergeDirect, MergeAllReduce };

    kfx_slab_layout layout;
    BoundedVolume<SDF_t, TargetDevice, Manage> local; // planes [layout.s0, layout.s1) of the whole volume
    BoundingBox full_bbox;
    kfx_comm* comm;
    HaloMode halo;
    RaycastMode raycast;
    MergeMode merge;
    int tiles;        // Exact: 0 = whole-image stages (kfx_slab_raycast_exact); >= 1 = the hand-over pipelined over this many image row-tiles
    int last_rounds;

    SlabVolume(size_t w, size_t h, size_t d, const BoundingBox& bbox, kfx_comm* comm_, HaloMode halo_ = HaloExchange,
               RaycastMode raycast_ = Composite, int ghost = 2)
        : layout(MakeLayout(d, bbox, comm_, ghost)),
          local(w, h, layout.s1 - layout.s0, BoundingBox(make_float3(bbox.Min().x, bbox.Min().y, layout.local_zmin),
                                                           make_float3(bbox.Max().x, bbox.Max().y, layout.local_zmax))),
          full_bbox(bbox), comm(comm_), halo(halo_), raycast(raycast_), merge(MergeDirect), tiles(0), last_rounds(0), key_(0), payload_(0), state_(0), scratch_(0),
          strips_(0), tiled_(0), cap_(0), strips_cap_(0), tiled_cap_(0)
    {
    }
    ~SlabVolume()
    {
        kfx_free(key_); kfx_free(payload_); kfx_free(state_); kfx_free(scratch_); kfx_free(strips_); kfx_free(tiled_);
    }
    SlabVolume(const SlabVolume&) = delete;
    SlabVolume& operator=(const SlabVolume&) = delete;

    // SdfFuse of this rank's planes, evaluated with the whole volume's voxel positions and over the whole volume's
    // extents (bit-identical to the same planes of a single-GPU volume), then the ghost planes are brought up to date
    void Fuse(Image<float> depth, Image<float4> norm, Mat<float,3,4> T_cw, ImageIntrinsics K, float trunc_dist, float maxw, float mincostheta)
    {
        const bool own_only = halo == HaloExchange && layout.world > 1;
        const size_t first = own_only ? layout.z0 : layout.s0, count = own_only ? layout.z1 - layout.z0 : layout.s1 - layout.s0;
        kfx_volume v = *local.abi();
        v.ptr = (unsigned char*)v.ptr + (first - layout.s0) * v.img_pitch;
        v.d = count;
        const kfx_slab s = {layout.full_d, first, layout.full_zmin, layout.full_zmax};
        GpuCheckStatus(kfx_sdf_fuse_slab(&v, &s, depth.abi(), norm.abi(), T_cw.m, &K.fu, trunc_dist, maxw, mincostheta, KFX_FUSE_SLAB_EXTENT, 0));
        if (own_only) GpuCheckStatus(kfx_slab_exchange_halos(local.abi(), &layout, comm, 0));
    }

    // Input distribution (the alternative to every rank preprocessing the frame itself): rank `root` holds the filtered depth
    // and the normal map, afterwards every rank does
    void BroadcastInputs(Image<float> depth, Image<float4> norm, int root = 0)
    {
        Reserve(depth.w * depth.h);
        GpuCheckStatus(kfx_slab_broadcast_inputs(depth.abi(), norm.abi(), payload_, root, comm, 0));   // payload_: 20 B per pixel
    }

    // RaycastSdf of the whole model; every rank returns with the same images
    void Raycast(Image<float> depth, Image<float4> norm, Image<float> img, Mat<float,3,4> T_wc, ImageIntrinsics K, float near, float far,
                 float trunc_dist, bool subpix = true)
    {
        Reserve(depth.w * depth.h);
        if (raycast == Exact && tiles > 0) {
            const size_t need = kfx_slab_exact_tiled_scratch_bytes(depth.w, depth.h, tiles, comm->world);
            if (need > tiled_cap_) {
                size_t pitch;
                kfx_free(tiled_);
                tiled_ = 0; tiled_cap_ = 0;
                GpuCheckStatus(kfx_alloc_pitched(&tiled_, &pitch, need, 1));
                tiled_cap_ = need;
            }
            GpuCheckStatus(kfx_slab_raycast_exact_tiled(depth.abi(), norm.abi(), img.abi(), tiled_, local.abi(), &layout, T_wc.m, &K.fu, near, far, trunc_dist,
                                                        subpix ? 1 : 0, tiles, comm, 0, 0, &last_rounds));
        } else if (raycast == Exact) {
            GpuCheckStatus(kfx_slab_raycast_exact(depth.abi(), norm.abi(), img.abi(), (float*)state_, scratch_, local.abi(), &layout, T_wc.m,
                                                  &K.fu, near, far, trunc_dist, subpix ? 1 : 0, comm, 0, &last_rounds));
        } else if (raycast == ExactAllReduce) {
            GpuCheckStatus(kfx_slab_raycast_exact_allreduce(depth.abi(), norm.abi(), img.abi(), (float*)state_, scratch_, local.abi(), &layout, T_wc.m,
                                                            &K.fu, near, far, trunc_dist, subpix ? 1 : 0, comm, 0, &last_rounds));
        } else {
            RaycastSdf(depth, norm, img, local, T_wc, K, near, far, trunc_dist, subpix);
            if (merge == MergeDirect && comm->all_to_all && comm->all_gather) {
                const size_t need = kfx_slab_composite_direct_scratch_bytes(depth.w, depth.h, comm->world);
                if (need > strips_cap_) {
                    size_t pitch;
                    kfx_free(strips_);
                    strips_ = 0; strips_cap_ = 0;
                    GpuCheckStatus(kfx_alloc_pitched(&strips_, &pitch, need, 1));
                    strips_cap_ = need;
                }
                GpuCheckStatus(kfx_slab_composite_direct(depth.abi(), norm.abi(), img.abi(), strips_, comm, 0));
            } else {
                GpuCheckStatus(kfx_slab_composite(depth.abi(), norm.abi(), img.abi(), (long long*)key_, (float*)payload_, comm, 0));
            }
        }
    }

private:
    static kfx_slab_layout MakeLayout(size_t d, const BoundingBox& bbox, kfx_comm* c, int ghost)
    {
        kfx_slab_layout L;
        GpuCheckStatus(kfx_slab_layout_init(&L, d, bbox.Min().z, bbox.Max().z, c->rank, c->world, ghost));
        return L;
    }
    void* key_; void* payload_; void* state_; void* scratch_; void* strips_; void* tiled_;
    size_t cap_, strips_cap_, tiled_cap_;
    void Reserve(size_t n)
    {
        if (n <= cap_) return;
        kfx_free(key_); kfx_free(payload_); kfx_free(state_); kfx_free(scratch_);
        size_t pitch;
        GpuCheckStatus(kfx_alloc_pitched(&key_, &pitch, n * sizeof(long long), 1));
        GpuCheckStatus(kfx_alloc_pitched(&payload_, &pitch, 5 * n * sizeof(float), 1));
        GpuCheckStatus(kfx_alloc_pitched(&state_, &pitch, KFX_RAY_STATE_PLANES * n * sizeof(float), 1));
        GpuCheckStatus(kfx_alloc_pitched(&scratch_, &pitch, kfx_slab_exact_scratch_bytes(n, 1), 1));
        cap_ = n;
    }
};

}
