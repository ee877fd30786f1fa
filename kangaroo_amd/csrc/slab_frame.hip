// slab_frame.hip -- kfx_slab_frame (include/kfx_slab.h): one frame of ONE RANK of the Z-slab partition enqueued by ONE call,
// the N > 1 counterpart of kfx_frame (frame.hip).  Host code only: every launch and every collective goes through the entry
// points the operator-by-operator drivers use (kfx_bilateral_f32, kfx_depth_to_vbo_normals_f32, kfx_sdf_fuse_slab,
// kfx_slab_exchange_halos, kfx_slab_broadcast_inputs, kfx_raycast_sdf + kfx_slab_composite[_direct], kfx_slab_raycast_exact_tiled),
// so a step writes what those calls write; the point is the host side.  At 8 ranks a frame's kernels take ~0.1 ms: an interpreter
// that issues a dozen calls and collectives per frame costs more than that.
// No reference counterpart (the reference is single-GPU: applications/kinectfusion/main.cpp:200-356 is the sequence).
#include <cmath>
#include <cstdlib>
#include <new>

#include "kfx_device.h"
#include "../../include/kfx_slab.h"
#include "slab_internal.h"

namespace {
constexpr int EV = 5;          // events per frame: before preprocess, before SdfFuse, after SdfFuse, after the march, after the merge
constexpr int OPEN_SLOTS = 8;  // exact march: frames whose "rays left open" word may still be on its way to the host
constexpr int PIPE_MAX = KFX_SLAB_PIPE_MAX;
// a frame of the pipelined exact raycast whose final exchange has not been enqueued yet (host-blocking transports trail by pipe - 1 frames)
struct PendingFinal { long long frame; int set, os, slot, tiles; unsigned timing; };
}

struct kfx_slab_frame {
    kfx_slab_frame_config cfg;
    kfx_comm* comm;
    long long frames;
    // device scratch (owned)
    void* exact;     size_t exact_bytes;
    void* strips;    size_t strips_bytes;
    void* keys;      size_t keys_bytes;      // int64 keys + payload of the all-reduce merge
    void* bcast;     size_t bcast_bytes;
    // exact march: the count of rays without a final status travels to pinned host words; checked once the event has passed
    int* h_open;
    hipEvent_t open_done[OPEN_SLOTS];
    long long open_frame[OPEN_SLOTS];
    int failed;                              // a past frame's march left rays open (reported by the next wait / step)
    int last_steps;
    // overlapped composite merge
    hipStream_t side;
    hipEvent_t marched, merged;
    int merge_pending;
    // Pipelined exact raycast (raycast EXACT + overlap, world > 1): frame k's march -- token steps, normals' stage, this rank's
    // contributions -- runs on the caller's stream through `comm`; its final exchange (all-to-all + strip sums + all-gather + the
    // images) runs on `side` through `side_comm`, a second communicator over the same ranks, into buffer / image set k % pipe, while
    // the caller's stream already carries frame k + 1: a rank's frame rate is bound by what IT has to do, not by the slowest token's
    // way through all ranks.  Transports whose collectives block the host enqueue a frame's final exchange pipe - 1
    // frames later, so that the host is not held in it either.
    kfx_comm side_comm;
    int have_side_comm;
    int pipe;                                // buffer / image sets in use (2 .. PIPE_MAX; 0: not pipelined)
    void* fin_mem[PIPE_MAX];
    kfx::ExactFinalBufs fin[PIPE_MAX];
    size_t fin_bytes;
    hipEvent_t marched_ev[PIPE_MAX], fin_done[PIPE_MAX];
    int fin_inflight[PIPE_MAX];              // fin_done[set] has been recorded and the caller's stream has not waited for it since
    PendingFinal pending[PIPE_MAX];
    int n_pending;
    long long rendered;                      // the latest frame whose final exchange has been enqueued (-1: none)
    // the packed texel image of the frame (owned; fuse.hip): written by the fused vbo / normals launch, staged by the SdfFuse of the same step
    kfx_image texels;
    int* agree;                              // a device word: the ranks' common verdict on a configure
    // timing ring
    int slots;
    unsigned timing;                         // which of the five events the next steps record (bit k: event k)
    hipEvent_t* ev;
    long long* ev_frame;
    unsigned char* ev_mask;
};

using namespace kfx;

static int hip_status(hipError_t e, const char* what)
{
    if (e == hipSuccess) return 0;
    (void)hipGetLastError();
    return set_error((int)e, what);
}

static int valid_image(const kfx_image& im, size_t elem) { return im.ptr && im.w > 0 && im.h > 0 && im.pitch >= im.w * elem; }

static int check_policies(const kfx_slab_frame_config& c, int world)
{
    if (c.halo != KFX_SLAB_HALO_RECOMPUTE && c.halo != KFX_SLAB_HALO_EXCHANGE) return set_error(KFX_E_RANGE, "kfx_slab_frame: halo");
    if (c.raycast != KFX_SLAB_RAYCAST_EXACT && c.raycast != KFX_SLAB_RAYCAST_COMPOSITE) return set_error(KFX_E_RANGE, "kfx_slab_frame: raycast");
    if (c.merge != KFX_SLAB_MERGE_DIRECT && c.merge != KFX_SLAB_MERGE_ALLREDUCE) return set_error(KFX_E_RANGE, "kfx_slab_frame: merge");
    if (c.inputs != KFX_SLAB_INPUTS_REPLICATE && c.inputs != KFX_SLAB_INPUTS_BROADCAST) return set_error(KFX_E_RANGE, "kfx_slab_frame: inputs");
    if (c.tiles < 0 || c.tiles > 64) return set_error(KFX_E_RANGE, "kfx_slab_frame: tiles in [0, 64]");
    // an overlapped merge issues its collectives from the side stream while the main stream may issue the ghost-plane exchange or
    // the input broadcast of the next frame: two sequences of collectives whose relative order could differ between ranks
    // (the exact raycast's overlap puts the final exchange on a second communicator -- its own order of operations -- so it may run
    //  beside any main-stream collective; it needs that communicator and at least two image sets)
    if (c.overlap && world > 1 && c.raycast == KFX_SLAB_RAYCAST_COMPOSITE && (c.halo != KFX_SLAB_HALO_RECOMPUTE || c.inputs != KFX_SLAB_INPUTS_REPLICATE))
        return set_error(KFX_E_RANGE, "kfx_slab_frame: the composite's overlap needs recomputed ghost planes and replicated inputs");
    if (c.overlap && world > 1 && c.raycast == KFX_SLAB_RAYCAST_EXACT) {
        if (c.pipe_depth < 2 || c.pipe_depth > PIPE_MAX) return set_error(KFX_E_RANGE, "kfx_slab_frame: the exact raycast's overlap needs pipe_depth in [2, KFX_SLAB_PIPE_MAX]");
        for (int k = 0; k < 3 * (c.pipe_depth - 1); ++k) {
            const kfx_image& im = c.pipe_images[k];
            const size_t elem = (k % 3 == 1) ? 16 : 4;
            if (!im.ptr || im.w < c.ray_img.w || im.h < c.ray_img.h || im.pitch < c.ray_img.w * elem || (((uintptr_t)im.ptr | im.pitch) & (elem - 1)))
                return set_error(KFX_E_SHAPE, "kfx_slab_frame: pipe_images (sets 1 .. pipe_depth - 1 of {ray_depth, ray_norm, ray_img})");
        }
    }
    return 0;
}

// (the new buffer first: a failed allocation leaves the old one -- and with it the old policies -- usable; round-5 advice)
static int grow(void** p, size_t* have, size_t need)
{
    if (need <= *have) return 0;
    void* q = nullptr;
    if (int e = hip_status(hipMalloc(&q, need), "kfx_slab_frame: hipMalloc")) return e;
    if (*p) (void)hipFree(*p);
    *p = q;
    *have = need;
    return 0;
}

// device scratch for the policies of `c` (create / configure: never inside a step)
static int ensure_scratch(kfx_slab_frame* f, const kfx_slab_frame_config& c)
{
    const size_t w = c.ray_img.w, h = c.ray_img.h;
    const int world = f->comm->world;
    if (c.raycast == KFX_SLAB_RAYCAST_EXACT) {
        if (int e = grow(&f->exact, &f->exact_bytes, kfx_slab_exact_tiled_scratch_bytes(w, h, c.tiles ? c.tiles : 4, world))) return e;
    } else if (world > 1) {
        if (c.merge == KFX_SLAB_MERGE_DIRECT) {
            if (int e = grow(&f->strips, &f->strips_bytes, kfx_slab_composite_direct_scratch_bytes(w, h, world))) return e;
        } else {
            if (int e = grow(&f->keys, &f->keys_bytes, w * h * (sizeof(long long) + KFX_COMPOSITE_PAYLOAD * sizeof(float)) + 256)) return e;
        }
    }
    if (c.inputs == KFX_SLAB_INPUTS_BROADCAST && world > 1)
        if (int e = grow(&f->bcast, &f->bcast_bytes, c.filtered.w * c.filtered.h * 20 + 256)) return e;
    if (c.raycast == KFX_SLAB_RAYCAST_EXACT && c.overlap && world > 1) {
        // the sets of the final exchange's buffers, the events between the two streams, the second communicator
        const size_t need = kfx::exact_final_bytes(w, h, world);
        for (int k = 0; k < c.pipe_depth; ++k) {
            if (need > f->fin_bytes || !f->fin_mem[k]) {
                void* q = nullptr;
                if (int e = hip_status(hipMalloc(&q, need), "kfx_slab_frame: hipMalloc")) return e;
                if (f->fin_mem[k]) (void)hipFree(f->fin_mem[k]);
                f->fin_mem[k] = q;
            }
            kfx::exact_final_carve(f->fin[k], f->fin_mem[k], w, h, world);
            if (!f->marched_ev[k]) if (int e = hip_status(hipEventCreateWithFlags(&f->marched_ev[k], hipEventDisableTiming), "kfx_slab_frame: hipEventCreate")) return e;
            if (!f->fin_done[k]) if (int e = hip_status(hipEventCreateWithFlags(&f->fin_done[k], hipEventDisableTiming), "kfx_slab_frame: hipEventCreate")) return e;
        }
        if (need > f->fin_bytes) f->fin_bytes = need;
        if (!f->have_side_comm) {
            if (!f->comm->dup) return set_error(KFX_E_RANGE, "kfx_slab_frame: the exact raycast's overlap needs a transport that can duplicate its communicator (kfx_comm::dup)");
            if (int e = f->comm->dup(f->comm, &f->side_comm)) return set_error(e, "kfx_slab_frame: kfx_comm::dup");
            f->have_side_comm = 1;
        }
    }
    return 0;
}

// A configure succeeds on every rank or on none: the ranks add up their failures (a rank that could not allocate must not run the
// old policies against peers that run the new ones: they would enter different collectives and wait for ever).  Every rank calls.
static int ranks_agree(kfx_slab_frame* f, int local_status)
{
    if (f->comm->world == 1 || !f->agree) return local_status;
    int flag = local_status ? 1 : 0, sum = 0;
    int e = hip_status(hipMemcpy(f->agree, &flag, sizeof(int), hipMemcpyHostToDevice), "kfx_slab_frame: hipMemcpy");
    const int c = f->comm->all_reduce(f->comm, f->agree, 1, KFX_COMM_SUM_I32, nullptr);
    if (c && !e) e = c;
    if (!e) e = hip_status(hipStreamSynchronize(nullptr), "kfx_slab_frame");
    if (!e) e = hip_status(hipMemcpy(&sum, f->agree, sizeof(int), hipMemcpyDeviceToHost), "kfx_slab_frame: hipMemcpy");
    if (local_status) return local_status;
    if (e) return e;
    return sum ? set_error(KFX_E_RANGE, "kfx_slab_frame: another rank could not adopt the configuration (kept the previous one on every rank)") : 0;
}

extern "C" int kfx_slab_frame_create(kfx_slab_frame** out, const kfx_slab_frame_config* cfg, kfx_comm* comm)
{
    if (!out || !cfg || !comm) return set_error(KFX_E_NULL, "kfx_slab_frame_create: null argument");
    *out = nullptr;
    if (!cfg->local.ptr) return set_error(KFX_E_NULL, "kfx_slab_frame_create: null volume");
    if (!valid_image(cfg->raw, 4) || !valid_image(cfg->filtered, 4) || !valid_image(cfg->vbo, 16) || !valid_image(cfg->normals, 16) ||
        !valid_image(cfg->ray_depth, 4) || !valid_image(cfg->ray_norm, 16) || !valid_image(cfg->ray_img, 4))
        return set_error(KFX_E_SHAPE, "kfx_slab_frame_create: image views");
    if (cfg->filtered.w != cfg->raw.w || cfg->filtered.h != cfg->raw.h || cfg->vbo.w != cfg->raw.w || cfg->vbo.h != cfg->raw.h ||
        cfg->normals.w != cfg->raw.w || cfg->normals.h != cfg->raw.h)
        return set_error(KFX_E_SHAPE, "kfx_slab_frame_create: the preprocess images differ in size");
    if (cfg->ray_depth.w < cfg->ray_img.w || cfg->ray_depth.h < cfg->ray_img.h || cfg->ray_norm.w < cfg->ray_img.w || cfg->ray_norm.h < cfg->ray_img.h)
        return set_error(KFX_E_SHAPE, "kfx_slab_frame_create: rendering images smaller than ray_img");
    const kfx_slab_layout& L = cfg->layout;
    if (L.rank != comm->rank || L.world != comm->world || cfg->local.d != L.s1 - L.s0)
        return set_error(KFX_E_SHAPE, "kfx_slab_frame_create: volume / communicator do not match the layout");
    if (cfg->timing_slots < 0 || cfg->timing_slots > (1 << 20)) return set_error(KFX_E_RANGE, "kfx_slab_frame_create: timing_slots");
    if (int e = check_policies(*cfg, comm->world)) return e;
    kfx_slab_frame* f = new (std::nothrow) kfx_slab_frame();
    if (!f) return set_error(KFX_E_RANGE, "kfx_slab_frame_create: out of memory");
    f->cfg = *cfg;
    f->comm = comm;
    f->slots = cfg->timing_slots;
    f->timing = 31u;
    for (int i = 0; i < OPEN_SLOTS; ++i) f->open_frame[i] = -1;
    f->rendered = -1;
    f->pipe = (cfg->overlap && cfg->raycast == KFX_SLAB_RAYCAST_EXACT && comm->world > 1) ? cfg->pipe_depth : 0;
    {   // the packed texel image of the frame's SdfFuse (no memory: the fuse packs per call instead)
        const size_t tpitch = (cfg->filtered.w * 16 + 255) / 256 * 256;
        void* buf = nullptr;
        if (tpitch < (1u << 24) && hipMalloc(&buf, kfx::texel_image_bytes(cfg->filtered.w, cfg->filtered.h)) == hipSuccess) f->texels = kfx_image{tpitch, buf, cfg->filtered.w, cfg->filtered.h};
        else (void)hipGetLastError();
    }
    int e = hip_status(hipMalloc((void**)&f->agree, 64), "kfx_slab_frame_create: hipMalloc");
    if (!e) e = ensure_scratch(f, f->cfg);
    if (!e) e = hip_status(hipHostMalloc((void**)&f->h_open, OPEN_SLOTS * sizeof(int), hipHostMallocDefault), "kfx_slab_frame_create: hipHostMalloc");
    if (!e) for (int i = 0; i < OPEN_SLOTS; ++i) f->h_open[i] = 0;
    for (int i = 0; i < OPEN_SLOTS && !e; ++i) e = hip_status(hipEventCreateWithFlags(&f->open_done[i], hipEventDisableTiming), "kfx_slab_frame_create: hipEventCreate");
    if (!e) e = hip_status(hipStreamCreateWithFlags(&f->side, hipStreamNonBlocking), "kfx_slab_frame_create: hipStreamCreate");
    if (!e) e = hip_status(hipEventCreateWithFlags(&f->marched, hipEventDisableTiming), "kfx_slab_frame_create: hipEventCreate");
    if (!e) e = hip_status(hipEventCreateWithFlags(&f->merged, hipEventDisableTiming), "kfx_slab_frame_create: hipEventCreate");
    if (!e && f->slots) {
        f->ev = new (std::nothrow) hipEvent_t[(size_t)f->slots * EV]();
        f->ev_frame = new (std::nothrow) long long[f->slots];
        f->ev_mask = new (std::nothrow) unsigned char[f->slots];
        if (!f->ev || !f->ev_frame || !f->ev_mask) e = set_error(KFX_E_RANGE, "kfx_slab_frame_create: out of memory");
        for (int i = 0; !e && i < f->slots; ++i) { f->ev_frame[i] = -1; f->ev_mask[i] = 0; }
        for (int i = 0; !e && i < f->slots * EV; ++i) e = hip_status(hipEventCreate(&f->ev[i]), "kfx_slab_frame_create: hipEventCreate");
    }
    if (e) {
        kfx_slab_frame_destroy(f);
        return e;
    }
    *out = f;
    return 0;
}

static int flush_pending(kfx_slab_frame* f);

extern "C" int kfx_slab_frame_destroy(kfx_slab_frame* f)
{
    if (!f) return 0;
    (void)flush_pending(f);   // (collectives: every rank destroys its frame at the same point)
    (void)hipDeviceSynchronize();
    if (f->ev) {
        for (int i = 0; i < f->slots * EV; ++i) if (f->ev[i]) (void)hipEventDestroy(f->ev[i]);
        delete[] f->ev;
    }
    delete[] f->ev_frame;
    delete[] f->ev_mask;
    for (int i = 0; i < OPEN_SLOTS; ++i) if (f->open_done[i]) (void)hipEventDestroy(f->open_done[i]);
    if (f->marched) (void)hipEventDestroy(f->marched);
    if (f->merged) (void)hipEventDestroy(f->merged);
    for (int k = 0; k < PIPE_MAX; ++k) {
        if (f->marched_ev[k]) (void)hipEventDestroy(f->marched_ev[k]);
        if (f->fin_done[k]) (void)hipEventDestroy(f->fin_done[k]);
        if (f->fin_mem[k]) (void)hipFree(f->fin_mem[k]);
    }
    if (f->have_side_comm && f->side_comm.destroy) f->side_comm.destroy(&f->side_comm);
    if (f->texels.ptr) (void)hipFree(f->texels.ptr);
    if (f->agree) (void)hipFree(f->agree);
    if (f->side) (void)hipStreamDestroy(f->side);
    if (f->h_open) (void)hipHostFree(f->h_open);
    if (f->exact) (void)hipFree(f->exact);
    if (f->strips) (void)hipFree(f->strips);
    if (f->keys) (void)hipFree(f->keys);
    if (f->bcast) (void)hipFree(f->bcast);
    (void)hipGetLastError();
    delete f;
    return 0;
}

extern "C" long long kfx_slab_frame_count(const kfx_slab_frame* f) { return f ? f->frames : 0; }
extern "C" int kfx_slab_frame_last_steps(const kfx_slab_frame* f) { return f ? f->last_steps : 0; }

// An event is a marker between two launches and costs the stream ~3 us (all five: a tenth of a 0.13 ms frame): a loop that is
// itself being timed records the two around SdfFuse (mask 6: its window, and the frame period from one to the next).
extern "C" int kfx_slab_frame_set_timing(kfx_slab_frame* f, unsigned mask)
{
    if (!f) return set_error(KFX_E_NULL, "kfx_slab_frame_set_timing: null frame");
    if (mask > 31u) return set_error(KFX_E_RANGE, "kfx_slab_frame_set_timing: mask");
    f->timing = mask;
    return 0;
}

// the exact marches whose "rays left open" word has arrived; all of them when `all` (after a synchronisation)
static int check_open(kfx_slab_frame* f, bool all)
{
    for (int i = 0; i < OPEN_SLOTS; ++i) {
        if (f->open_frame[i] < 0) continue;
        if (!all && hipEventQuery(f->open_done[i]) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (all) (void)hipEventSynchronize(f->open_done[i]);
        if (((volatile int*)f->h_open)[i] != 0 && !f->cfg.unchecked) f->failed = 1;
        f->open_frame[i] = -1;
    }
    if (f->failed) {
        f->failed = 0;
        return set_error(KFX_E_RANGE, "kfx_slab_frame: an exact march left rays without a final status");
    }
    return 0;
}

// where frame `frame`'s rendering lives: set frame % pipe of the pipelined exact raycast (set 0 = the configuration's ray_* images)
static void images_of(const kfx_slab_frame* f, int set, const kfx_image** d, const kfx_image** n, const kfx_image** i)
{
    const kfx_slab_frame_config& c = f->cfg;
    if (set <= 0) { *d = &c.ray_depth; *n = &c.ray_norm; *i = &c.ray_img; return; }
    *d = &c.pipe_images[(set - 1) * 3]; *n = &c.pipe_images[(set - 1) * 3 + 1]; *i = &c.pipe_images[(set - 1) * 3 + 2];
}

// the final exchange of the oldest pending frame, enqueued on the side stream through the side communicator (collectives: every rank
// gets here at the same point of its own sequence of calls)
static int finalise_oldest(kfx_slab_frame* f)
{
    if (f->n_pending <= 0) return 0;
    const PendingFinal P = f->pending[0];
    for (int k = 1; k < f->n_pending; ++k) f->pending[k - 1] = f->pending[k];
    f->n_pending -= 1;
    int status = 0;
    const auto note = [&](int e) { if (e && !status) status = e; };
    note(hip_status(hipStreamWaitEvent(f->side, f->marched_ev[P.set], 0), "kfx_slab_frame: hipStreamWaitEvent"));
    const kfx_image *d, *n, *i;
    images_of(f, P.set, &d, &n, &i);
    note(kfx::exact_tiled_finalise(d, n, i, f->exact, &f->fin[P.set], P.tiles, &f->side_comm, (kfx_stream)f->side, f->h_open + P.os));
    if (hipEventRecord(f->open_done[P.os], f->side) == hipSuccess) f->open_frame[P.os] = P.frame;
    else (void)hipGetLastError();
    if (P.slot >= 0 && (P.timing & 16u) && f->ev_frame[P.slot] == P.frame) {
        if (hipEventRecord(f->ev[(size_t)P.slot * EV + 4], f->side) == hipSuccess) f->ev_mask[P.slot] |= 16u;
        else (void)hipGetLastError();
    }
    note(hip_status(hipEventRecord(f->fin_done[P.set], f->side), "kfx_slab_frame: hipEventRecord"));
    f->fin_inflight[P.set] = 1;
    f->rendered = P.frame;
    return status;
}

static int flush_pending(kfx_slab_frame* f)
{
    int status = 0;
    while (f->n_pending > 0) {
        const int e = finalise_oldest(f);
        if (e && !status) status = e;
    }
    return status;
}

extern "C" int kfx_slab_frame_wait(kfx_slab_frame* f, kfx_stream stream)
{
    if (!f) return set_error(KFX_E_NULL, "kfx_slab_frame_wait: null frame");
    int e = 0;
    if (f->merge_pending) {
        e = hip_status(hipStreamWaitEvent((hipStream_t)stream, f->merged, 0), "kfx_slab_frame_wait: hipStreamWaitEvent");
        f->merge_pending = 0;
    }
    // pipelined exact raycast: every frame stepped so far gets its final exchange enqueued, and `stream` waits for all of them
    const int p = flush_pending(f);
    if (p && !e) e = p;
    for (int k = 0; k < PIPE_MAX; ++k)
        if (f->fin_inflight[k]) {
            const int w = hip_status(hipStreamWaitEvent((hipStream_t)stream, f->fin_done[k], 0), "kfx_slab_frame_wait: hipStreamWaitEvent");
            if (w && !e) e = w;
        }
    const int o = check_open(f, false);
    return e ? e : o;
}

extern "C" int kfx_slab_frame_wait_frame(kfx_slab_frame* f, long long frame, kfx_stream stream)
{
    if (!f) return set_error(KFX_E_NULL, "kfx_slab_frame_wait_frame: null frame");
    if (f->pipe <= 0) return kfx_slab_frame_wait(f, stream);
    if (frame < 0 || frame >= f->frames || f->frames - frame > f->pipe) return set_error(KFX_E_RANGE, "kfx_slab_frame_wait_frame: that frame's set has been taken by a later frame (or it was never stepped)");
    if (frame > f->rendered) return set_error(KFX_E_RANGE, "kfx_slab_frame_wait_frame: that frame's final exchange has not been enqueued yet (kfx_slab_frame_wait enqueues all)");
    return hip_status(hipStreamWaitEvent((hipStream_t)stream, f->fin_done[(int)(frame % f->pipe)], 0), "kfx_slab_frame_wait_frame: hipStreamWaitEvent");
}

extern "C" int kfx_slab_frame_images(const kfx_slab_frame* f, long long frame, kfx_image* depth, kfx_image* norm, kfx_image* img)
{
    if (!f || !depth || !norm || !img) return set_error(KFX_E_NULL, "kfx_slab_frame_images: null argument");
    if (frame < 0) frame = f->frames - 1;
    if (frame < 0 || frame >= f->frames || (f->pipe > 0 && f->frames - frame > f->pipe))
        return set_error(KFX_E_RANGE, "kfx_slab_frame_images: that frame's rendering has been overwritten (or was never stepped)");
    const kfx_image *d, *n, *i;
    images_of(f, f->pipe > 0 && f->cfg.raycast == KFX_SLAB_RAYCAST_EXACT ? (int)(frame % f->pipe) : 0, &d, &n, &i);
    *depth = *d; *norm = *n; *img = *i;
    return 0;
}

extern "C" int kfx_slab_frame_configure(kfx_slab_frame* f, int halo, int raycast, int merge, int inputs, int overlap, int tiles)
{
    if (!f) return set_error(KFX_E_NULL, "kfx_slab_frame_configure: null frame");
    kfx_slab_frame_config c = f->cfg;
    if (halo >= 0) c.halo = halo;
    if (raycast >= 0) c.raycast = raycast;
    if (merge >= 0) c.merge = merge;
    if (inputs >= 0) c.inputs = inputs;
    if (overlap >= 0) c.overlap = overlap ? 1 : 0;
    if (tiles >= 0) c.tiles = tiles;
    if (int e = check_policies(c, f->comm->world)) return e;   // (argument errors: every rank makes them alike)
    // the scratch may be replaced: nothing of this frame object may still be in flight (pending final exchanges are enqueued first:
    // collectives, entered by every rank alike)
    int st = flush_pending(f);
    const int y = hip_status(hipDeviceSynchronize(), "kfx_slab_frame_configure");
    if (y && !st) st = y;
    f->merge_pending = 0;
    for (int k = 0; k < PIPE_MAX; ++k) f->fin_inflight[k] = 0;
    const int o = check_open(f, true);
    // A failed allocation (or a communicator that could not be duplicated) is rank-local; the ranks agree on the outcome before
    // anybody adopts the new policies, and a rank that fails keeps its old buffers (grow allocates before it frees): round-5 advice
    if (!st) st = ensure_scratch(f, c);
    st = ranks_agree(f, st);
    if (st) return st;
    f->cfg = c;
    f->pipe = (c.overlap && c.raycast == KFX_SLAB_RAYCAST_EXACT && f->comm->world > 1) ? c.pipe_depth : 0;
    return o;
}

extern "C" int kfx_slab_frame_reset(kfx_slab_frame* f, kfx_stream stream)
{
    if (!f) return set_error(KFX_E_NULL, "kfx_slab_frame_reset: null frame");
    return kfx_sdf_reset(&f->cfg.local, __builtin_nanf(""), stream);   // "never observed" = (NaN, 0) (main.cpp:229)
}

extern "C" int kfx_slab_frame_step(kfx_slab_frame* f, const kfx_image* raw, const float T_wc[12], const float* T_cw, unsigned parts, kfx_stream stream)
{
    if (!f || !T_wc) return set_error(KFX_E_NULL, "kfx_slab_frame_step: null argument");
    if (parts == 0) parts = KFX_FRAME_PREPROCESS | KFX_FRAME_FUSE | KFX_FRAME_RAYCAST;
    const kfx_slab_frame_config& c = f->cfg;
    const kfx_slab_layout& L = c.layout;
    kfx_comm* comm = f->comm;
    const int world = comm->world, rank = comm->rank;
    const kfx_image* src = raw ? raw : &c.raw;
    float inv[12];
    if (!T_cw) {   // SE3inv: [R^T | -R^T t], evaluated in double and rounded once (as kfx_frame_step)
        for (int i = 0; i < 3; ++i) {
            double t = 0.0;
            for (int j = 0; j < 3; ++j) {
                inv[i * 4 + j] = T_wc[j * 4 + i];
                t += (double)T_wc[j * 4 + i] * (double)T_wc[j * 4 + 3];
            }
            inv[i * 4 + 3] = (float)-t;
        }
        T_cw = inv;
    }
    if ((parts & KFX_FRAME_RAYCAST) &&
        ((c.raycast == KFX_SLAB_RAYCAST_EXACT && !f->exact) ||
         (c.raycast == KFX_SLAB_RAYCAST_COMPOSITE && world > 1 && !(c.merge == KFX_SLAB_MERGE_DIRECT ? f->strips : f->keys))))
        return set_error(KFX_E_NULL, "kfx_slab_frame_step: no scratch for the configured raycast (a failed kfx_slab_frame_configure)");
    if ((parts & KFX_FRAME_PREPROCESS) && c.inputs == KFX_SLAB_INPUTS_BROADCAST && world > 1 && !f->bcast)
        return set_error(KFX_E_NULL, "kfx_slab_frame_step: no scratch for the input broadcast (a failed kfx_slab_frame_configure)");
    const hipStream_t s = (hipStream_t)stream;
    // From here on nothing returns early: a local failure must not keep this rank out of a collective its peers enter
    int status = 0;
    const auto note = [&](int e) { if (e && !status) status = e; };
    hipEvent_t* ev = nullptr;
    int slot = -1;
    if (f->slots) {
        slot = (int)(f->frames % f->slots);
        f->ev_frame[slot] = f->timing ? f->frames : -1;
        f->ev_mask[slot] = 0;
        if (f->timing) ev = f->ev + (size_t)slot * EV;
    }
    const auto record = [&](int k, hipStream_t on) {
        if (!ev || !(f->timing & (1u << k))) return;
        if (hipEventRecord(ev[k], on) == hipSuccess) f->ev_mask[slot] |= (unsigned char)(1u << k);
        else { (void)hipGetLastError(); note(set_error(KFX_E_RANGE, "kfx_slab_frame_step: hipEventRecord")); }
    };
    // the previous frame's overlapped merge still reads the rendering this frame will overwrite, and its strips
    if (f->merge_pending && (parts & KFX_FRAME_RAYCAST)) {
        note(hip_status(hipStreamWaitEvent(s, f->merged, 0), "kfx_slab_frame_step: hipStreamWaitEvent"));
        f->merge_pending = 0;
    }
    note(check_open(f, false));

    record(0, s);
    // the packed texels of the SdfFuse kernels' LDS-DMA staging travel from this step's own preprocess to this step's SdfFuse only
    const kfx_image* tex = ((parts & KFX_FRAME_PREPROCESS) && (parts & KFX_FRAME_FUSE) && f->texels.ptr &&
                            !(c.inputs == KFX_SLAB_INPUTS_BROADCAST && world > 1 && rank != 0)) ? &f->texels : nullptr;
    if (parts & KFX_FRAME_PREPROCESS) {
        const bool bcast = c.inputs == KFX_SLAB_INPUTS_BROADCAST && world > 1;
        if (!bcast || rank == 0) {
            note(kfx_bilateral_f32(&c.filtered, src, c.bilateral_gs, c.bilateral_gr, c.bilateral_size, c.bilateral_minval, 1, stream));
            note(kfx::depth_to_vbo_normals_texels(&c.vbo, &c.normals, &c.filtered, c.K, 1.0f, tex, stream));
        }
        if (bcast) note(kfx_slab_broadcast_inputs(&c.filtered, &c.normals, f->bcast, 0, comm, stream));
    }
    record(1, s);
    if (parts & KFX_FRAME_FUSE) {
        // this rank's planes with the WHOLE volume's voxel positions and extents (bit-identical to the same planes of a single
        // volume); ghost planes integrated here too (recompute) or fetched from the neighbours afterwards (exchange)
        const bool own_only = c.halo == KFX_SLAB_HALO_EXCHANGE && world > 1;
        const size_t first = own_only ? L.z0 : L.s0, count = own_only ? L.z1 - L.z0 : L.s1 - L.s0;
        kfx_volume v = c.local;
        v.ptr = (unsigned char*)v.ptr + (first - L.s0) * v.img_pitch;
        v.d = count;
        const kfx_slab sl = {L.full_d, first, L.full_zmin, L.full_zmax};
        note(kfx::sdf_fuse_slab_texels(&v, &sl, &c.filtered, &c.normals, tex, T_cw, c.K, c.trunc_dist, c.max_w, c.mincostheta, KFX_FUSE_SLAB_EXTENT, stream));
        if (own_only) note(kfx_slab_exchange_halos(&c.local, &L, comm, stream));
    }
    record(2, s);
    if (parts & KFX_FRAME_RAYCAST) {
        if (c.raycast == KFX_SLAB_RAYCAST_EXACT) {
            const int os = (int)(f->frames % OPEN_SLOTS);
            if (f->open_frame[os] >= 0) {   // (eight frames old: long done)
                (void)hipEventSynchronize(f->open_done[os]);
                if (((volatile int*)f->h_open)[os] != 0 && !c.unchecked) note(set_error(KFX_E_RANGE, "kfx_slab_frame: an exact march left rays without a final status"));
                f->open_frame[os] = -1;
            }
            int steps = 0;
            const int tiles = c.tiles ? c.tiles : 4;
            if (f->pipe > 0) {
                // Pipelined: the march on the caller's stream, its final exchange on the side stream through the side communicator
                // into set k % pipe; the caller's stream goes on with the next frame.  The set's buffers and images are free once
                // the final exchange of frame k - pipe has run.
                const int set = (int)(f->frames % f->pipe);
                if (f->fin_inflight[set]) {
                    note(hip_status(hipStreamWaitEvent(s, f->fin_done[set], 0), "kfx_slab_frame_step: hipStreamWaitEvent"));
                    f->fin_inflight[set] = 0;
                }
                note(kfx::exact_tiled_march(f->exact, &f->fin[set], &c.local, &L, T_wc, c.K, c.near, c.far, c.trunc_dist, 1, tiles, (int)c.ray_img.w,
                                            (int)c.ray_img.h, comm, stream, &steps));
                f->last_steps = steps;
                record(3, s);
                note(hip_status(hipEventRecord(f->marched_ev[set], s), "kfx_slab_frame_step: hipEventRecord"));
                f->pending[f->n_pending++] = PendingFinal{f->frames, set, os, slot, tiles, ev ? f->timing : 0u};
                // a host that blocks in collectives trails the final exchange by pipe - 1 frames (it meets its peers there: the last
                // rank of the token chain is that far behind the first); RCCL enqueues and goes on
                const int lag = (f->side_comm.flags & KFX_COMM_HOST_BLOCKING) ? f->pipe - 1 : 0;
                while (f->n_pending > lag) note(finalise_oldest(f));
            } else {
                note(kfx_slab_raycast_exact_tiled(&c.ray_depth, &c.ray_norm, &c.ray_img, f->exact, &c.local, &L, T_wc, c.K, c.near, c.far, c.trunc_dist, 1,
                                                  tiles, comm, stream, f->h_open + os, &steps));
                f->last_steps = steps;
                if (hipEventRecord(f->open_done[os], s) == hipSuccess) f->open_frame[os] = f->frames;
                else (void)hipGetLastError();
                record(3, s);
            }
        } else {
            note(kfx_raycast_sdf(&c.ray_depth, &c.ray_norm, &c.ray_img, &c.local, T_wc, c.K, c.near, c.far, c.trunc_dist, 1, stream));
            record(3, s);
            if (world > 1) {
                hipStream_t ms = s;
                if (c.overlap) {
                    note(hip_status(hipEventRecord(f->marched, s), "kfx_slab_frame_step: hipEventRecord"));
                    note(hip_status(hipStreamWaitEvent(f->side, f->marched, 0), "kfx_slab_frame_step: hipStreamWaitEvent"));
                    ms = f->side;
                }
                if (c.merge == KFX_SLAB_MERGE_DIRECT) {
                    note(kfx_slab_composite_direct(&c.ray_depth, &c.ray_norm, &c.ray_img, f->strips, comm, (kfx_stream)ms));
                } else {
                    long long* key = static_cast<long long*>(f->keys);
                    float* payload = reinterpret_cast<float*>(key + c.ray_img.w * c.ray_img.h + 2);   // (16-byte aligned: the key count is even + 2)
                    payload = reinterpret_cast<float*>(((uintptr_t)payload + 15) & ~(uintptr_t)15);
                    note(kfx_slab_composite(&c.ray_depth, &c.ray_norm, &c.ray_img, key, payload, comm, (kfx_stream)ms));
                }
                record(4, ms);
                if (c.overlap) {
                    note(hip_status(hipEventRecord(f->merged, f->side), "kfx_slab_frame_step: hipEventRecord"));
                    f->merge_pending = 1;
                }
            }
        }
    }
    f->frames += 1;
    return status;
}

extern "C" int kfx_slab_frame_timings(kfx_slab_frame* f, long long first_frame, int n_frames, float* ms)
{
    if (!f || !ms) return set_error(KFX_E_NULL, "kfx_slab_frame_timings: null argument");
    if (!f->slots) return set_error(KFX_E_RANGE, "kfx_slab_frame_timings: the frame was created without timing slots");
    if (n_frames <= 0) return 0;
    const long long last = first_frame + n_frames - 1;
    if (first_frame < 0 || last >= f->frames || f->frames - first_frame > f->slots) return set_error(KFX_E_RANGE, "kfx_slab_frame_timings: frames not in the ring");
    const float nan = __builtin_nanf("");
    for (int i = 0; i < n_frames; ++i) {
        const long long fr = first_frame + i;
        const int slot = (int)(fr % f->slots);
        float* o = ms + (size_t)i * KFX_SLAB_FRAME_TIMING_FIELDS;
        for (int k = 0; k < KFX_SLAB_FRAME_TIMING_FIELDS; ++k) o[k] = nan;
        if (f->ev_frame[slot] != fr) {
            if (f->ev_frame[slot] > fr) return set_error(KFX_E_RANGE, "kfx_slab_frame_timings: frame overwritten");
            continue;   // stepped with the events switched off
        }
        const unsigned m = f->ev_mask[slot];
        hipEvent_t* e = f->ev + (size_t)slot * EV;
        int lastk = -1;
        for (int k = 0; k < EV; ++k) if (m & (1u << k)) lastk = k;
        if (lastk < 0) continue;
        hipError_t he = hipEventSynchronize(e[lastk]);
        if (he == hipSuccess && (m & 16u)) he = hipEventSynchronize(e[4]);   // (the merge's event may live on the side stream)
        const auto span = [&](int a, int b, float* out) {
            if (he == hipSuccess && (m & (1u << a)) && (m & (1u << b))) he = hipEventElapsedTime(out, e[a], e[b]);
        };
        int firstk = 0;
        while (!(m & (1u << firstk))) ++firstk;
        span(0, 1, &o[0]);
        span(1, 2, &o[1]);
        span(2, 3, &o[2]);
        span(3, 4, &o[3]);
        if (firstk != lastk) span(firstk, lastk, &o[4]);
        // period: this frame's first recorded event to the same event of the next frame
        if (he == hipSuccess && fr + 1 < f->frames) {
            const int ns = (int)((fr + 1) % f->slots);
            if (f->ev_frame[ns] == fr + 1 && (f->ev_mask[ns] & (1u << firstk))) {
                he = hipEventSynchronize(f->ev[(size_t)ns * EV + firstk]);
                if (he == hipSuccess) he = hipEventElapsedTime(&o[5], e[firstk], f->ev[(size_t)ns * EV + firstk]);
            }
        }
        if (he != hipSuccess) { (void)hipGetLastError(); return set_error((int)he, "kfx_slab_frame_timings: hipEventElapsedTime"); }
    }
    return 0;
}

// synchronise everything this frame object has in flight and report a failed exact march
extern "C" int kfx_slab_frame_sync(kfx_slab_frame* f, kfx_stream stream)
{
    if (!f) return set_error(KFX_E_NULL, "kfx_slab_frame_sync: null frame");
    int e = flush_pending(f);
    const int y = hip_status(hipStreamSynchronize((hipStream_t)stream), "kfx_slab_frame_sync");
    if (y && !e) e = y;
    if (!e) e = hip_status(hipStreamSynchronize(f->side), "kfx_slab_frame_sync");
    f->merge_pending = 0;
    for (int k = 0; k < PIPE_MAX; ++k) f->fin_inflight[k] = 0;
    const int o = check_open(f, true);
    return e ? e : o;
}
