/*
 * kfx_slab.h -- the multi-GPU half of the C ABI: a BoundedVolume<SDF_t> partitioned into Z-slabs, one rank per GPU
 * (BASELINE.json north_star: "Z-slab-partitioned across the 8 GPUs of one node with RCCL halo exchange over
 * xGMI for rays that cross slabs").  The reference is single-GPU: there is no roo:: counterpart, so these entry points are
 * additions beside the drop-in boundary of kfx.h, callable from C / C++ hosts without Python.
 *
 * Rank r of `world` stores planes [s0, s1) = [z0 - ghost, z1 + ghost) clipped to the volume, of which it OWNS [z0, z1)
 * (ghost >= 1: the trilinear z+1 corner and the gradient stencil's z-1 / z+1 cells, Volume.h:240-289).  The local storage
 * is an ordinary BoundedVolume whose box is VoxelPositionInUnits of its first / last stored plane -- the view
 * BoundedVolume::SubBoundingVolume produces (BoundedVolume.h:156-164).
 *
 *   per frame and rank:  kfx_sdf_fuse_slab (kfx.h) on the owned planes            -- no communication, bit-identical
 *                        kfx_slab_exchange_halos                                  -- ghost planes from the two neighbours
 *                        kfx_raycast_sdf on the local view + kfx_slab_composite[_direct] -- nearest hit of all slabs (all-to-all + all-gather of image strips, or 2 all-reduces)
 *                     or kfx_slab_raycast_exact                                   -- march state handed from slab to slab,
 *                                                                                    bit-identical to the single-volume march
 *
 * Collectives go through a kfx_comm, a small table of transport functions.  Two transports ship:
 *   kfx_comm_create_rccl     (libkfx_rccl.so, links librccl): one PROCESS per GPU, RCCL all-reduce / all-gather / grouped send-recv over
 *                            xGMI.  Rendezvous of the ncclUniqueId through a file.
 *   kfx_comm_create_threads  (libkfx.so): the ranks are host THREADS of one process sharing one device -- the emulation
 *                            used to exercise the slab logic where only one GPU exists (tests, apps --transport threads).
 * All buffers are device pointers; operations are enqueued on `stream` (the threads transport uses the null stream).
 */
#ifndef KFX_SLAB_H
#define KFX_SLAB_H

#include "kfx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- transport ---------------------------------------------------------------------------------------------- */
#define KFX_COMM_MIN_I64 0 /* count x int64, minimum   (composite keys) */
#define KFX_COMM_SUM_F32 1 /* count x float, sum       (composite payload: one rank contributes per pixel, so exact) */
#define KFX_COMM_SUM_I32 2 /* count x int32, sum       (march state bits: one rank contributes per pixel) */

typedef struct kfx_comm {
    int rank, world;
    void* impl;
    /* in-place all-reduce of a dense device buffer */
    int (*all_reduce)(struct kfx_comm* c, void* buf, size_t count, int op, kfx_stream stream);
    /* neighbour exchange along the rank order: send_lo / recv_lo talk to rank - 1, send_hi / recv_hi to rank + 1; a byte
     * count of 0 (or a missing neighbour) skips that direction.  One batched group of point-to-point operations. */
    int (*exchange)(struct kfx_comm* c, const void* send_lo, void* recv_lo, size_t bytes_lo, const void* send_hi, void* recv_hi,
                    size_t bytes_hi, kfx_stream stream);
    int (*barrier)(struct kfx_comm* c);
    void (*destroy)(struct kfx_comm* c);
    /* `bytes` of a dense device buffer from rank `root` to every rank (the filtered depth / normal maps of a frame when only
     * one rank runs the preprocessing, SURVEY.md 8(e) "input distribution") */
    int (*broadcast)(struct kfx_comm* c, void* buf, size_t bytes, int root, kfx_stream stream);
    /* personalised exchange over the full mesh: chunk r of `send` (bytes_per_rank each) goes to rank r, chunk r of `recv` comes
     * from rank r (one grouped send / recv per peer: every xGMI link at once); and its counterpart, every rank's `send`
     * (bytes_per_rank) to chunk `rank` of every rank's `recv`.  The direct-send composite's two phases. */
    int (*all_to_all)(struct kfx_comm* c, const void* send, void* recv, size_t bytes_per_rank, kfx_stream stream);
    int (*all_gather)(struct kfx_comm* c, const void* send, void* recv, size_t bytes_per_rank, kfx_stream stream);
} kfx_comm;

/* In-process transport: fills comms[0 .. world) for `world` host threads of this process that share the current device;
 * every collective must be called by all of them (each with its own comms[r]).  Destroy through comms[0] after the threads
 * have joined. */
int kfx_comm_create_threads(kfx_comm* comms, int world);

/* RCCL transport (libkfx_rccl.so).  rank 0 removes any existing `rendezvous_file` and writes the ncclUniqueId there (exclusive
 * temporary + rename, mode 0600, no symlinks followed) behind a launch nonce folded from KFX_RUN_ID / TORCHELASTIC_RUN_ID /
 * MASTER_PORT / SLURM_JOB_ID and from the ranks' PARENT process (pid + start time: the launcher; KFX_RDV_PARENT=0 leaves it out
 * for ranks started by hand from different shells -- give those a KFX_RUN_ID that is unique per launch); the other ranks wait (at
 * most timeout_s seconds) for a regular file of this user that carries their nonce and does not precede their own process by
 * more than the launch skew (10 s; KFX_RDV_SLACK_S), so a file left by a crashed run is never taken for this run's.  The caller
 * has selected its device (hipSetDevice) beforehand.  Returns 0, KFX_E_*, or 1000 + ncclResult_t. */
int kfx_comm_create_rccl(kfx_comm* comm, int rank, int world, const char* rendezvous_file, int timeout_s);

/* ---- slab layout --------------------------------------------------------------------------------------------- */
typedef struct kfx_slab_layout {
    size_t full_d;             /* planes of the whole volume */
    float  full_zmin, full_zmax;
    int    rank, world, ghost;
    size_t z0, z1;             /* owned planes [z0, z1): contiguous, sizes differ by at most one plane between ranks */
    size_t s0, s1;             /* stored planes [s0, s1) = owned +- ghost, clipped */
    float  local_zmin, local_zmax; /* z of planes s0 and s1 - 1 by VoxelPositionInUnits of the whole volume: the local box */
} kfx_slab_layout;
int kfx_slab_layout_init(kfx_slab_layout* L, size_t full_d, float full_zmin, float full_zmax, int rank, int world, int ghost);

/* Input distribution, the alternative to every rank running the (cheap) preprocessing itself: rank `root` holds the frame's
 * filtered depth and normal maps, afterwards every rank does.  Rows travel without their padding when the images are
 * pitched (staged through `scratch`, (4 + 16) * w * h bytes of device memory; may be null for dense images). */
int kfx_slab_broadcast_inputs(const kfx_image* depth, const kfx_image* norm, void* scratch, int root, kfx_comm* comm, kfx_stream stream);

/* Refresh the ghost planes of `local` (planes [s0, s1) of the layout, fp32 SDF_t or fp16 cells: any cell size, whole
 * img_pitch-sized planes travel) from the neighbours' owned planes.  Requires every rank to own at least `ghost` planes. */
int kfx_slab_exchange_halos(const kfx_volume* local, const kfx_slab_layout* L, kfx_comm* comm, kfx_stream stream);

/* Nearest-hit composite of per-slab raycasts: on entry depth / norm / img hold this rank's RaycastSdf of its local view,
 * on return every rank holds the merged images.  key: w*h int64, payload: KFX_COMPOSITE_PAYLOAD*w*h float, dense device scratch of the caller. */
int kfx_slab_composite(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, long long* key, float* payload,
                       kfx_comm* comm, kfx_stream stream);
/* The same merge by direct sends (kfx.h, kfx_composite_strips_*): every rank owns one strip of the image; one all-to-all brings the
 * world copies of a strip to its owner, the owner keeps the nearest hit per pixel, one all-gather returns the merged strips.  Same
 * winner per pixel, same images (a -0 component of a winning normal stays -0 here and becomes +0 in the payload's sum).  scratch:
 * kfx_slab_composite_direct_scratch_bytes(w, h, world) bytes of device memory. */
size_t kfx_slab_composite_direct_scratch_bytes(size_t w, size_t h, int world);
int kfx_slab_composite_direct(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, void* scratch, kfx_comm* comm, kfx_stream stream);

/* The exact march.  A ray's march state (lambda, last_sdf, delta, status) travels with the ray from slab to slab: world + 1
 * stages of kfx_raycast_sdf_slab -- each rank advances the rays whose current sample lies in the planes it owns -- with an
 * exchange of the march planes between NEIGHBOUR ranks after each stage (point-to-point: one xGMI link per direction), then
 * one all-reduce of the finalising ranks' results and one 4-byte read-back.  No host synchronisation between the stages.
 * Every rank returns with the images of kfx_raycast_sdf on the whole volume, bit for bit.  state: KFX_RAY_STATE_PLANES * w*h
 * floats; scratch: kfx_slab_exact_scratch_bytes(w, h) bytes (device).  *rounds_out (optional) receives the number of stages.
 * kfx_slab_raycast_exact_allreduce is the cross-check: the same kernels with the state merged over ALL ranks after every
 * round (one SUM all-reduce + one host-side termination test per round, <= world + 2 rounds); same images. */
size_t kfx_slab_exact_scratch_bytes(size_t w, size_t h);
int kfx_slab_raycast_exact(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, float* state, void* scratch,
                           const kfx_volume* local, const kfx_slab_layout* L, const float T_wc[12], const float K[4],
                           float near, float far, float trunc_dist, int subpix, kfx_comm* comm, kfx_stream stream, int* rounds_out);
int kfx_slab_raycast_exact_allreduce(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, float* state, void* scratch,
                                     const kfx_volume* local, const kfx_slab_layout* L, const float T_wc[12], const float K[4],
                                     float near, float far, float trunc_dist, int subpix, kfx_comm* comm, kfx_stream stream, int* rounds_out);

#ifdef __cplusplus
}
#endif
#endif /* KFX_SLAB_H */
