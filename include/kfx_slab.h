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
    /* neighbour exchange whose four legs have their own sizes (0 / a missing neighbour skips a leg): what this rank sends down
     * must be what rank - 1 receives from above, and so on.  The tile-pipelined hand-over's steps. */
    int (*exchange_v)(struct kfx_comm* c, const void* send_lo, size_t bytes_send_lo, void* recv_lo, size_t bytes_recv_lo,
                      const void* send_hi, size_t bytes_send_hi, void* recv_hi, size_t bytes_recv_hi, kfx_stream stream);
    /* KFX_COMM_HOST_BLOCKING: a collective returns only when every rank has entered it (the in-process and the callback transports);
     * 0: collectives are enqueued on the stream and the call returns at once (RCCL).  kfx_slab_frame's pipelined frames let a host
     * that blocks trail the final exchange of a frame by the frames in flight instead of stalling in it. */
    int flags;
    /* a second communicator over the same ranks with its own order of operations (every rank calls; RCCL: ncclCommSplit): what the
     * frame object's side stream uses, so that its collectives and the main stream's need no common order across ranks */
    int (*dup)(struct kfx_comm* c, struct kfx_comm* out);
} kfx_comm;
#define KFX_COMM_HOST_BLOCKING 1
/* A transport of the caller's own fills the table itself: zero-initialise the struct first (entries after `destroy` are optional and
 * are tested against NULL; the table has grown at its end between versions of this header). */

/* In-process transport: fills comms[0 .. world) for `world` host threads of this process that share the current device;
 * every collective must be called by all of them (each with its own comms[r]).  Destroy through comms[0] after the threads
 * have joined. */
int kfx_comm_create_threads(kfx_comm* comms, int world);
/* ... whose neighbour exchanges are matched PAIRWISE and in order per directed link, the way RCCL matches ncclSend / ncclRecv, instead
 * of being a barrier of all ranks: a rank with nothing to pass on does not take part in a step, ranks drift apart by whole frames,
 * and a leg whose two sides disagree (one skips it, or names another size) BLOCKS -- as it would on RCCL, where it hangs -- until
 * timeout_ms have passed (<= 0: 10 s), then fails with KFX_E_TIMEOUT.  The collectives proper stay barriers of the group. */
int kfx_comm_create_threads_p2p(kfx_comm* comms, int world, int timeout_ms);

/* Loop-back transport (libkfx.so) for measuring ONE rank of a `world`-rank job on one GPU: every collective moves the bytes a
 * real one would deliver to this rank, but from this rank's own buffers (all_reduce: nothing arrives, all_to_all / all_gather:
 * device-local copies, exchange: the send buffers come back).  The frame's kernels, launches and copies are those of a real
 * rank; the images are NOT a rendering (nobody marched the other slabs).  Host-overhead and kernel-time floors only. */
int kfx_comm_create_loopback(kfx_comm* comm, int rank, int world);

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

/* The hand-over pipelined over image row-tiles ("<= 7 hops pipelined over image tiles").  A ray moves
 * through the slabs monotonically, so the march state of a tile travels as a token: upwards 0 -> 1 -> ... -> world - 1 (rays
 * with rising z) and downwards at the same time; rank r marches tile t when the upward token reaches it (step r + t) and when
 * the downward one does (step world - 1 - r + t), and passes the tile's march state (three planes: lambda, last_sdf, and delta or -status: 12 B per pixel) on to ONE neighbour -- a
 * message of 1 / tiles of the image per link and step, world + tiles - 1 steps, against world stages of whole images.  Every rank
 * initialises every ray itself (the entry slab's owner starts it); a rank adopts a neighbour's copy of a ray when that copy is
 * NEWER than its own (a final or hit status beats "marching", a larger lambda beats a smaller) and still under way.  One last
 * stage with a whole-image neighbour exchange lets a hit whose sub-step interpolation fell back across a slab boundary get its
 * normal from the rank that owns the gradient's base plane; then one all-reduce of the finalising ranks' results.  Same samples,
 * same images as kfx_raycast_sdf on the whole volume, bit for bit, for any number of tiles.
 * The finalised results reach every rank by direct sends, as the composite's strips do (all-to-all of the ranks' contributions to
 * a strip to its owner, integer sum, all-gather; KFX_SLAB_FINALISE=allreduce or a transport without those keeps one all-reduce).
 * scratch: kfx_slab_exact_tiled_scratch_bytes(w, h, tiles, world) bytes of device memory.  h_open: NULL -- the call synchronises the
 * stream at its end and fails with KFX_E_RANGE if a ray is left without a final status; else a host-visible (pinned) word that
 * receives that count asynchronously: the call returns without synchronising and the CALLER checks the word once the stream
 * has passed (kfx_slab_frame does).  Needs comm->exchange_v. */
size_t kfx_slab_exact_tiled_scratch_bytes(size_t w, size_t h, int tiles, int world);
/* Ghost planes per side (kfx_slab_layout_init's `ghost`) with which the hand-over needs no last stage: a hit's sub-step interpolation
 * puts it at most one march step -- max(trunc_dist, voxel size) times the longest ray direction of the image -- behind the sample that
 * found it, so with that many planes (+ the gradient stencil's) stored beyond its own the rank that finds a hit always evaluates the
 * normal itself.  kfx_slab_raycast_exact_tiled and kfx_slab_frame drop the stage (one whole-image neighbour exchange and one march
 * launch per frame) whenever the layout's ghost is at least this; KFX_SLAB_NORMALS_STAGE=1 keeps it.  Presumes what SdfFuse
 * guarantees -- no cell beyond the truncation distance the raycast is called with (cu_sdffusion.cu:49); a volume that breaks it
 * leaves rays without a final status, which the call counts and reports (KFX_E_RANGE), never a wrong image.  size_x, vol_w: the
 * volume's extent and cells along x (the march's minimum step is the voxel's x size, cu_raycast.cu:52). */
int kfx_slab_exact_ghost(size_t full_d, float full_zmin, float full_zmax, float size_x, size_t vol_w, float trunc_dist, const float K[4], int w, int h);
/* keep != 0: the last stage stays whatever the ghost width (process-wide; every rank alike).  Returns the previous setting. */
int kfx_slab_set_normals_stage(int keep);
int kfx_slab_raycast_exact_tiled(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, void* scratch,
                                 const kfx_volume* local, const kfx_slab_layout* L, const float T_wc[12], const float K[4],
                                 float near, float far, float trunc_dist, int subpix, int tiles, kfx_comm* comm, kfx_stream stream,
                                 int* h_open, int* steps_out);

/* ---- a slab rank's frame as ONE call ---------------------------------------------------------------------------
 * kfx_frame_step's counterpart for N > 1 (include/kfx.h: a frame is a fraction of a millisecond of GPU work, and a host that
 * issues it operator by operator leaves gaps of the same order): BilateralFilter -> DepthToVbo + NormalsFromVbo (or rank 0's,
 * broadcast) -> kfx_sdf_fuse_slab on this rank's planes (+ ghost-plane exchange) -> the slab raycast and its collectives, all
 * enqueued by one call on the caller's stream, with device events around the parts in a ring.
 *   raycast  EXACT (default): kfx_slab_raycast_exact_tiled -- bit-identical to the single volume (the path BASELINE's north
 *            star names: march state handed from slab to slab over the neighbour links);
 *            COMPOSITE: kfx_raycast_sdf on the local view + nearest-hit merge (direct sends or two all-reduces); the march
 *            restarts at each slab entry, so silhouette rays can end differently (512^3 / 8 slabs, S_room: 85 of 307 200
 *            pixels change between hit and miss) -- a throughput variant outside the image tolerance of the single-GPU path.
 *   overlap  COMPOSITE (halo RECOMPUTE, inputs REPLICATE only): the merge of frame k runs on the frame's own side stream under
 *            frame k + 1's preprocessing and SdfFuse; the images are valid after kfx_slab_frame_wait.
 *            EXACT: frames pipelined across the ranks.  The token chain of the hand-over takes world + tiles - 1 steps from the first
 *            rank to the last, but a rank's own part of it is `tiles` visits; what ties the ranks together once per frame is the final
 *            exchange of the finalised pixels (an all-to-all and an all-gather over all ranks).  With overlap that exchange leaves the
 *            caller's stream: frame k's march -- token steps, the normals' stage, this rank's contributions -- is enqueued on the
 *            caller's stream through the frame's communicator, its final exchange on the frame's side stream through a SECOND
 *            communicator over the same ranks (kfx_comm::dup: its own order of operations, so it may run beside any main-stream
 *            collective), into image / buffer set k % pipe_depth, while the caller's stream already carries frame k + 1's
 *            preprocessing, SdfFuse (stream order keeps it behind the rank's own march k) and march.  Known-pose streams: rank r works
 *            on frame k + 1 while the token of frame k is still on its way to rank world - 1; the frame rate is bound by a rank's own
 *            work instead of the chain.  (A tracked loop needs the images of frame k for the pose of frame k + 1: no overlap there.)
 *            Same bits as without overlap.  Frame k's images: kfx_slab_frame_images, valid after kfx_slab_frame_wait / _sync.
 *            Transports whose collectives block the host (KFX_COMM_HOST_BLOCKING) enqueue frame k's final exchange while frame
 *            k + pipe_depth - 1 is stepped, so the host runs ahead like the streams do. */
#define KFX_SLAB_HALO_RECOMPUTE    0
#define KFX_SLAB_HALO_EXCHANGE     1
#define KFX_SLAB_RAYCAST_EXACT     0
#define KFX_SLAB_RAYCAST_COMPOSITE 1
#define KFX_SLAB_MERGE_DIRECT      0
#define KFX_SLAB_MERGE_ALLREDUCE   1
#define KFX_SLAB_INPUTS_REPLICATE  0
#define KFX_SLAB_INPUTS_BROADCAST  1
#define KFX_SLAB_PIPE_MAX          4
typedef struct kfx_slab_frame kfx_slab_frame;
typedef struct kfx_slab_frame_config {
    kfx_volume local;                        /* this rank's stored planes [layout.s0, layout.s1), box = local_zmin .. local_zmax */
    kfx_slab_layout layout;
    kfx_image raw, filtered, vbo, normals;   /* the whole frame's images (every rank holds them) */
    kfx_image ray_depth, ray_norm, ray_img;  /* the merged rendering */
    float K[4];
    float bilateral_gs, bilateral_gr, bilateral_minval;
    unsigned bilateral_size;
    float near, far, trunc_dist, max_w, mincostheta;
    int halo, raycast, merge, inputs;        /* KFX_SLAB_* */
    int overlap;                             /* 1: composite merge under the next frame / exact raycast: pipelined frames (see above) */
    int tiles;                               /* exact raycast: image row-tiles of the hand-over (>= 1; 0: the library's default, 4) */
    int unchecked;                           /* 1: do not fail when the exact march leaves rays open (loop-back measurements) */
    int timing_slots;                        /* 0: no events */
    /* raycast EXACT with overlap = 1 (pipelined frames, above): pipe_depth = the number of image / buffer sets, 2 .. KFX_SLAB_PIPE_MAX;
     * set 0 is {ray_depth, ray_norm, ray_img}, sets 1 .. pipe_depth - 1 are pipe_images[3 (s - 1) ...] = {depth, norm, img} of the
     * caller (device memory, the sizes of set 0).  0: no pipelining */
    int pipe_depth;
    kfx_image pipe_images[3 * (KFX_SLAB_PIPE_MAX - 1)];
} kfx_slab_frame_config;
#define KFX_SLAB_FRAME_TIMING_FIELDS 6      /* ms: preprocess (+ broadcast), SdfFuse (+ ghost planes), RaycastSdf kernels or the whole exact march,
                                               composite merge (NaN: exact), frame (first to last event), period (to the next frame's first event) */
int kfx_slab_frame_create(kfx_slab_frame** out, const kfx_slab_frame_config* cfg, kfx_comm* comm);
int kfx_slab_frame_destroy(kfx_slab_frame* f);
/* change the policies between frames (bench.py times the variants on one object); a value < 0 keeps the current one */
int kfx_slab_frame_configure(kfx_slab_frame* f, int halo, int raycast, int merge, int inputs, int overlap, int tiles);
int kfx_slab_frame_reset(kfx_slab_frame* f, kfx_stream stream);   /* SdfReset(local, NaN) */
int kfx_slab_frame_step(kfx_slab_frame* f, const kfx_image* raw, const float T_wc[12], const float* T_cw, unsigned parts, kfx_stream stream);
/* make `stream` wait for an overlapped merge / every pipelined frame's final exchange (enqueueing the ones a host-blocking transport
 * still trails: a collective point, every rank calls) and report a failed exact march of an earlier frame (KFX_E_RANGE) */
int kfx_slab_frame_wait(kfx_slab_frame* f, kfx_stream stream);
/* where the rendering of `frame` (< 0: the last one stepped) lives: the configuration's ray_* images, or -- pipelined exact raycast --
 * set frame % pipe_depth; KFX_E_RANGE once a later frame has taken the set */
int kfx_slab_frame_images(const kfx_slab_frame* f, long long frame, kfx_image* depth, kfx_image* norm, kfx_image* img);
/* pipelined exact raycast: make `stream` wait for the final exchange of ONE frame without enqueueing anybody else's -- possible once
 * pipe_depth - 1 later frames have been stepped (a host-blocking transport enqueues it then; RCCL at once); KFX_E_RANGE before that
 * and once the set has been taken by a later frame.  Not a collective point. */
int kfx_slab_frame_wait_frame(kfx_slab_frame* f, long long frame, kfx_stream stream);
/* synchronise `stream` and whatever the frame object still has in flight (an overlapped merge on its side stream); reports a failed
 * exact march of any frame so far */
int kfx_slab_frame_sync(kfx_slab_frame* f, kfx_stream stream);
long long kfx_slab_frame_count(const kfx_slab_frame* f);
/* which of the five events the following steps record (bit 0 before the preprocessing, 1 before SdfFuse, 2 after it, 3 after the
 * march, 4 after the composite merge; 0: none, 31: all, 6: the two around SdfFuse -- an event is a marker between two launches and
 * costs the stream ~3 us).  frame = first to last recorded event, period = first recorded event to the same event of the next frame. */
#define KFX_SLAB_FRAME_EVENTS_ALL  31u
#define KFX_SLAB_FRAME_EVENTS_FUSE  6u
int kfx_slab_frame_set_timing(kfx_slab_frame* f, unsigned mask);
int kfx_slab_frame_timings(kfx_slab_frame* f, long long first_frame, int n_frames, float* ms);
int kfx_slab_frame_last_steps(const kfx_slab_frame* f);           /* steps of the last exact march: world + tiles - 1 token steps + the normals' stage */

#ifdef __cplusplus
}
#endif
#endif /* KFX_SLAB_H */
