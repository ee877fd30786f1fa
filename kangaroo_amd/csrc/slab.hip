// slab.hip -- Z-slab partition of the TSDF volume across ranks (include/kfx_slab.h): layout, halo exchange, nearest-hit
// composite, exact march hand-over, and the in-process "threads" transport.  The collectives themselves belong to the
// transport behind kfx_comm (RCCL: comm_rccl.cpp); this file only orders kernels and collectives on the caller's stream.
// No reference counterpart (the reference is single-GPU); the host-side protocol is the one of
// kangaroo_amd/pipeline.py::SlabPipeline, so that C / C++ applications can use slabs without Python.
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <new>

#include "kfx_device.h"
#include "../../include/kfx_slab.h"
#include "slab_internal.h"

namespace kfx {

// ---- exact march: merging the per-rank states between rounds -------------------------------------------------------
// state planes (kfx.h, KFX_RAY_STATE_PLANES): 0 lambda, 1 last_sdf, 2 delta, 3 status, 4 touched, 5-7 normal, 8 shade.
// In a round exactly one rank touches a pixel, so SUM over ranks of (touched ? bits : 0) is that rank's state.
__global__ __launch_bounds__(256) void k_state_contrib(const int* __restrict__ state, int* __restrict__ contrib, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool touched = state[4 * n + i] != 0;
#pragma unroll
    for (int p = 0; p < 5; ++p) contrib[p * n + i] = touched ? state[p * n + i] : 0;
}

// after the all-reduce: adopt the touching rank's state; *flag |= 1 while any ray is still marching (status 0) or a hit
// awaits its normal (status 3)
__global__ __launch_bounds__(256) void k_state_merge(int* __restrict__ state, const int* __restrict__ contrib, size_t n, int merge,
                                                     int* __restrict__ flag)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    bool open = false;
    if (i < n) {
        if (merge && contrib[4 * n + i] != 0) {
#pragma unroll
            for (int p = 0; p < 5; ++p) state[p * n + i] = contrib[p * n + i];
        }
        const float status = __int_as_float(state[3 * n + i]);
        open = status == 0.0f || status == 3.0f;
    }
    if (__ballot(open) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// ---- exact march, hand-over protocol (kfx_slab_raycast_exact) --------------------------------------------------------
// A ray's march state travels with the ray: after a rank has marched its segment the state goes to the two neighbour ranks
// only (planes 0-4 as they are: nothing is compacted, the receiver picks the rays that came its way), and a rank adopts from
// a neighbour's planes the rays that neighbour advanced in this stage and that are still under way (status 0: marching, 3:
// hit found, normal pending).  No rank ever acts on a stale copy: k_raycast_sdf_slab advances a ray only while the trilinear
// base plane of its current sample is one the rank owns, a ray moves through the planes monotonically, so the plane of a
// stale position belongs to a rank the ray has left -- and that rank holds the newer state.
// fin[i] = this rank finalised pixel i (set its status to 1 hit / 2 miss), the one rank whose result counts at the end.
__global__ __launch_bounds__(256) void k_handover_merge(int* __restrict__ state, const int* __restrict__ from_lo, const int* __restrict__ from_hi,
                                                        int* __restrict__ fin, size_t n, int claim_untouched_misses)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool touched = state[4 * n + i] != 0;
    const float status = __int_as_float(state[3 * n + i]);
    if (touched && (status == 1.0f || status == 2.0f)) fin[i] = 1;
    // rays that never enter the box are misses from the first stage on, on every rank alike: rank 0 answers for them
    if (claim_untouched_misses && !touched && status == 2.0f) fin[i] = 1;
    const int* src = nullptr;
    if (from_lo && from_lo[4 * n + i] != 0) {
        const float st = __int_as_float(from_lo[3 * n + i]);
        if (st == 0.0f || st == 3.0f) src = from_lo;
    }
    if (!src && from_hi && from_hi[4 * n + i] != 0) {
        const float st = __int_as_float(from_hi[3 * n + i]);
        if (st == 0.0f || st == 3.0f) src = from_hi;
    }
    if (src) {
#pragma unroll
        for (int p = 0; p < 4; ++p) state[p * n + i] = src[p * n + i];
    }
}

// contribution of this rank to the final images: lambda, status, normal, shade of the pixels it finalised, zero elsewhere
// (integer bit patterns: NaN and -0 survive the SUM over ranks, where exactly one rank is non-zero)
__global__ __launch_bounds__(256) void k_handover_contrib(const int* __restrict__ state, const int* __restrict__ fin, int* __restrict__ contrib, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool mine = fin[i] != 0;
    contrib[0 * n + i] = mine ? state[0 * n + i] : 0;
    contrib[1 * n + i] = mine ? state[3 * n + i] : 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) contrib[(2 + p) * n + i] = mine ? state[(5 + p) * n + i] : 0;
}

// the summed contributions back into the state; *open += pixels without a final status (none, unless the protocol is broken)
__global__ __launch_bounds__(256) void k_handover_finish(int* __restrict__ state, const int* __restrict__ contrib, size_t n, int* __restrict__ open)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    bool bad = false;
    if (i < n) {
        state[0 * n + i] = contrib[0 * n + i];
        state[3 * n + i] = contrib[1 * n + i];
#pragma unroll
        for (int p = 0; p < 4; ++p) state[(5 + p) * n + i] = contrib[(2 + p) * n + i];
        const float status = __int_as_float(contrib[1 * n + i]);
        bad = !(status == 1.0f || status == 2.0f);
    }
    const unsigned long long m = __ballot(bad);
    if (m != 0ull && (threadIdx.x & 63) == 0) atomicAdd(open, __popcll(m));
}

// ---- tile-pipelined hand-over (kfx_slab_raycast_exact_tiled) ---------------------------------------------------------
// State per row-tile t (kfx.h, kfx_raycast_sdf_slab_tiles): NP march planes at M + (t * NP + k) * P -- the token of a tile --, results
// at Rz + (t * 4 + k) * P.  NP = 3, packed {lambda, last_sdf, delta of a marching ray or -status}: 12 B per pixel ("12 B
// per ray of march state"), whenever the truncation distance is positive; else the four planes {lambda, last_sdf, delta, status}.
// "Newer wins": a ray's copies on different ranks are snapshots of ONE march; status 1 / 2 (final) is later than 3 (hit, normal
// pending) is later than 0 (marching), and of two marching snapshots the one with the larger lambda is later (every step adds a
// positive delta).  A rank never advances a stale copy -- it advances a ray only while the base plane of its current sample is
// one it owns, and a kernel run leaves no marching ray of the tile inside the rank's own planes -- so adopting the neighbour's
// newer, still-open snapshot is all the merging there is; k_raycast_sdf_slab does it at the start of a visit (raycast.hip).

// Where word c of pixel i of the six result planes {lambda, status, n.x, n.y, n.z, shade} lives: dense [6][n] (S = 0), or by image
// strips (S = pixels per strip, kfx_composite_strip_pixels): strip j = pixels [j S, (j + 1) S) of the row-major image holds its six
// planes side by side at (j * 6 + c) * S -- the layout in which strip j travels to its owner rank j and back.
__device__ __forceinline__ size_t result_index(int c, size_t i, size_t n, unsigned S)
{
    if (S == 0) return (size_t)c * n + i;
    const size_t j = i / S;
    return (j * 6 + (size_t)c) * S + (i - j * S);
}

// this rank's contribution to the final images: lambda, status, normal, shade of the pixels it finalised, zero elsewhere (integer
// bit patterns: NaN and -0 survive the sum over ranks, of which exactly one is non-zero per pixel)
__global__ __launch_bounds__(256) void k_tiles_contrib(const int* __restrict__ M, const int* __restrict__ Rz, const int* __restrict__ fin,
                                                       int* __restrict__ contrib, int w, int h, int R, size_t P, unsigned S, int packed)
{
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= w || v >= h) return;
    const size_t n = (size_t)w * h, i = (size_t)v * w + u;
    const int t = v / R;
    const size_t q = (size_t)(v - t * R) * w + u;
    const int* st = M + (size_t)t * (packed ? 3 : 4) * P + q;
    const int* rs = Rz + (size_t)t * 4 * P + q;
    const bool mine = fin[i] != 0;
    contrib[result_index(0, i, n, S)] = mine ? st[0] : 0;
    int status = 0;   // the status plane's bits: a finalised ray's 1.0f / 2.0f
    if (mine) {
        if (packed) { const float c = __int_as_float(st[2 * P]); status = __float_as_int(c < 0.f ? -c : 0.f); }
        else status = st[3 * P];
    }
    contrib[result_index(1, i, n, S)] = status;
#pragma unroll
    for (int k = 0; k < 4; ++k) contrib[result_index(2 + k, i, n, S)] = mine ? rs[k * P] : 0;
}

// the owner of a strip adds up the ranks' copies of it: in + r * 6 * S = rank r's [6][S]; out [6][S]
__global__ __launch_bounds__(256) void k_strip_sum(const int* __restrict__ in, int* __restrict__ out, size_t words, int world)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= words) return;
    int acc = 0;
    for (int r = 0; r < world; ++r) acc += in[(size_t)r * words + i];
    out[i] = acc;
}

struct OutImages { unsigned char *dptr, *nptr, *iptr; size_t dpitch, npitch, ipitch; int w, h; };

// the summed contributions -> depth / normal / shade images (cu_raycast.cu:92-102); *open += pixels without a final status
__global__ __launch_bounds__(256) void k_tiles_finish(const OutImages o, const int* __restrict__ contrib, int* __restrict__ open, unsigned S)
{
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    bool bad = false;
    if (u < o.w && v < o.h) {
        const size_t n = (size_t)o.w * o.h, i = (size_t)v * o.w + u;
        const float depth = __int_as_float(contrib[result_index(0, i, n, S)]), status = __int_as_float(contrib[result_index(1, i, n, S)]);
        bad = !(status == 1.0f || status == 2.0f);
        const bool hit = status == 1.0f && depth > 0.0f;
        *(reinterpret_cast<float*>(o.dptr + (size_t)v * o.dpitch) + u) = hit ? depth : __builtin_nanf("");
        *(reinterpret_cast<float*>(o.iptr + (size_t)v * o.ipitch) + u) = hit ? __int_as_float(contrib[result_index(5, i, n, S)]) : 0.0f;
        *(reinterpret_cast<float4*>(o.nptr + (size_t)v * o.npitch) + u) =
            hit ? make_float4(__int_as_float(contrib[result_index(2, i, n, S)]), __int_as_float(contrib[result_index(3, i, n, S)]),
                              __int_as_float(contrib[result_index(4, i, n, S)]), 1.0f)
                : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const unsigned long long m = __ballot(bad);
    if (m != 0ull && (threadIdx.x & 63) == 0) atomicAdd(open, __popcll(m));
}

// ---- threads transport: reduction over the ranks' buffers by one kernel -------------------------------------------
constexpr int MAX_THREAD_RANKS = 16;
struct PtrList { void* p[MAX_THREAD_RANKS]; };

template <typename T, int OP>
__global__ __launch_bounds__(256) void k_group_reduce(PtrList bufs, int world, size_t count)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    T acc = reinterpret_cast<const T*>(bufs.p[0])[i];
    for (int r = 1; r < world; ++r) {
        const T v = reinterpret_cast<const T*>(bufs.p[r])[i];
        if constexpr (OP == 0) acc = v < acc ? v : acc;
        else acc = acc + v;   // rank order: every rank ends up with the same bits
    }
    for (int r = 0; r < world; ++r) reinterpret_cast<T*>(bufs.p[r])[i] = acc;
}

struct ThreadGroup {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long generation = 0;
    void* buf[MAX_THREAD_RANKS];
    const void* send_lo[MAX_THREAD_RANKS];
    const void* send_hi[MAX_THREAD_RANKS];
    size_t bytes_lo[MAX_THREAD_RANKS], bytes_hi[MAX_THREAD_RANKS];   // what a rank SENDS down / up
    // per-rank error of a collective, double-buffered by the collective's sequence parity: a rank that has left collective k
    // and already entered k + 1 posts into the other set while a slower peer may still be reading k's in group_status(); the
    // set of k is written again in k + 2, which every rank enters only after it returned from k + 1 -- whose barriers all
    // ranks passed after they had returned from k (round-3 advice).  seq[r] is rank r's own count of collectives.
    std::atomic<int> err[2][MAX_THREAD_RANKS];
    unsigned seq[MAX_THREAD_RANKS] = {};
    // Point-to-point mode (kfx_comm_create_threads_p2p): the neighbour exchanges are matched PAIRWISE, in order, per directed link --
    // the way RCCL matches ncclSend / ncclRecv -- instead of being a barrier of all ranks: a rank with nothing to pass on does not
    // take part, ranks drift apart by whole frames (what kfx_slab_frame's pipelined frames are for), and a leg whose two sides
    // disagree -- one side skips it, or names another size -- does not fail at once: it BLOCKS, as it would on RCCL, until
    // `timeout_ms` have passed, and then reports KFX_E_TIMEOUT.  link[0][r]: r -> r + 1, link[1][r]: r -> r - 1; one message in
    // flight per link (the sender returns from its call once the receiver has taken the message).
    struct Link { const void* src = nullptr; size_t bytes = 0; bool full = false; };
    bool p2p = false;
    int timeout_ms = 0;
    Link link[2][MAX_THREAD_RANKS];
    ThreadGroup* dup_out = nullptr;   // threads_dup: the new group travels from rank 0 to the others through the old one
    ThreadGroup() { for (auto& set : err) for (auto& e : set) e.store(0); }

    void wait_all()
    {
        std::unique_lock<std::mutex> lk(m);
        const unsigned long g = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != g; });
        }
    }
};

static int hip_status(hipError_t e, const char* what)
{
    if (e == hipSuccess) return 0;
    (void)hipGetLastError();
    return set_error((int)e, what);
}

// Every rank passes both barriers of a collective whatever happened locally: a rank that returned early on its own error
// would leave its peers waiting on the condition variable for ever.  Local errors are posted in err[parity][rank]; after the
// last barrier every rank returns the first error any rank posted for THIS collective, so all ranks return the same status.
static int group_status(const ThreadGroup* g, int par)
{
    for (int r = 0; r < g->world; ++r)
        if (const int e = g->err[par][r].load()) return e;
    return 0;
}

static int threads_all_reduce(kfx_comm* c, void* buf, size_t count, int op, kfx_stream stream)
{
    ThreadGroup* g = static_cast<ThreadGroup*>(c->impl);
    int st = 0;
    if (!buf && count) st = set_error(KFX_E_NULL, "kfx_comm(threads) all_reduce: null buffer");
    if (!st) st = hip_status(hipStreamSynchronize((hipStream_t)stream), "kfx_comm(threads) all_reduce"); // this rank's producers are done
    const int par = (int)(g->seq[c->rank]++ & 1u);
    g->err[par][c->rank].store(st);
    g->buf[c->rank] = buf;
    g->wait_all();
    if (c->rank == 0 && count && group_status(g, par) == 0) {
        PtrList pl;
        for (int r = 0; r < g->world; ++r) pl.p[r] = g->buf[r];
        const dim3 grid((unsigned)((count + 255) / 256));
        hipStream_t s = (hipStream_t)stream;
        if (op == KFX_COMM_MIN_I64) hipLaunchKernelGGL((k_group_reduce<long long, 0>), grid, dim3(256), 0, s, pl, g->world, count);
        else if (op == KFX_COMM_SUM_F32) hipLaunchKernelGGL((k_group_reduce<float, 1>), grid, dim3(256), 0, s, pl, g->world, count);
        else if (op == KFX_COMM_SUM_I32) hipLaunchKernelGGL((k_group_reduce<int, 1>), grid, dim3(256), 0, s, pl, g->world, count);
        else st = set_error(KFX_E_RANGE, "kfx_comm all_reduce: unknown op");
        if (!st) st = check_launch("kfx_comm(threads) all_reduce");
        if (!st) st = hip_status(hipStreamSynchronize(s), "kfx_comm(threads) all_reduce");
        if (st) g->err[par][0].store(st);
    }
    g->wait_all();
    return group_status(g, par);
}

static int threads_exchange_v(kfx_comm* c, const void* send_lo, size_t bytes_send_lo, void* recv_lo, size_t bytes_recv_lo, const void* send_hi,
                              size_t bytes_send_hi, void* recv_hi, size_t bytes_recv_hi, kfx_stream stream)
{
    ThreadGroup* g = static_cast<ThreadGroup*>(c->impl);
    const int r = c->rank;
    int st = hip_status(hipStreamSynchronize((hipStream_t)stream), "kfx_comm(threads) exchange");
    const int par = (int)(g->seq[r]++ & 1u);
    g->err[par][r].store(st);
    g->send_lo[r] = send_lo; g->bytes_lo[r] = r > 0 ? bytes_send_lo : 0;
    g->send_hi[r] = send_hi; g->bytes_hi[r] = r + 1 < g->world ? bytes_send_hi : 0;
    g->wait_all();
    hipStream_t s = (hipStream_t)stream;
    if (!st && r > 0 && bytes_recv_lo && g->err[par][r - 1].load() == 0) { // what rank - 1 sends upwards
        if (g->bytes_hi[r - 1] != bytes_recv_lo) st = set_error(KFX_E_SHAPE, "kfx_comm exchange: neighbours disagree on the byte count");
        else st = hip_status(hipMemcpyAsync(recv_lo, g->send_hi[r - 1], bytes_recv_lo, hipMemcpyDeviceToDevice, s), "kfx_comm(threads) exchange");
    } else if (!st && r > 0 && g->err[par][r - 1].load() == 0 && g->bytes_hi[r - 1] != 0) {
        st = set_error(KFX_E_SHAPE, "kfx_comm exchange: the lower neighbour sends what this rank does not receive");
    }
    if (!st && r + 1 < g->world && bytes_recv_hi && g->err[par][r + 1].load() == 0) { // what rank + 1 sends downwards
        if (g->bytes_lo[r + 1] != bytes_recv_hi) st = set_error(KFX_E_SHAPE, "kfx_comm exchange: neighbours disagree on the byte count");
        else st = hip_status(hipMemcpyAsync(recv_hi, g->send_lo[r + 1], bytes_recv_hi, hipMemcpyDeviceToDevice, s), "kfx_comm(threads) exchange");
    } else if (!st && r + 1 < g->world && g->err[par][r + 1].load() == 0 && g->bytes_lo[r + 1] != 0) {
        st = set_error(KFX_E_SHAPE, "kfx_comm exchange: the upper neighbour sends what this rank does not receive");
    }
    if (!st) st = hip_status(hipStreamSynchronize(s), "kfx_comm(threads) exchange");
    if (st) g->err[par][r].store(st);   // (atomic: a neighbour may be reading the slot; it sees 0 or an error, and every rank
                                        //  reads the final value after the barrier below)
    g->wait_all(); // nobody reuses a send buffer before its reader is done
    return group_status(g, par);
}

// Point-to-point mode: post the sends, take what the neighbours posted, wait until the own messages have been taken.  No barrier,
// no other rank involved; a leg without a partner -- or whose partner names another size -- blocks until the timeout.
static int threads_exchange_v_p2p(kfx_comm* c, const void* send_lo, size_t bytes_send_lo, void* recv_lo, size_t bytes_recv_lo, const void* send_hi,
                                  size_t bytes_send_hi, void* recv_hi, size_t bytes_recv_hi, kfx_stream stream)
{
    ThreadGroup* g = static_cast<ThreadGroup*>(c->impl);
    const int r = c->rank;
    const bool has_lo = r > 0, has_hi = r + 1 < g->world;
    const bool sl = has_lo && bytes_send_lo, rl = has_lo && bytes_recv_lo, sh = has_hi && bytes_send_hi, rh = has_hi && bytes_recv_hi;
    if (!sl && !rl && !sh && !rh) return 0;   // (as the RCCL transport: nothing to do, nobody to meet)
    if ((sl && !send_lo) || (rl && !recv_lo) || (sh && !send_hi) || (rh && !recv_hi)) return set_error(KFX_E_NULL, "kfx_comm(threads, p2p) exchange: null buffer");
    hipStream_t s = (hipStream_t)stream;
    int st = hip_status(hipStreamSynchronize(s), "kfx_comm(threads, p2p) exchange");   // this rank's producers are done
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(g->timeout_ms > 0 ? g->timeout_ms : 10000);
    std::unique_lock<std::mutex> lk(g->m);
    ThreadGroup::Link* up = &g->link[0][r];      // r -> r + 1
    ThreadGroup::Link* down = &g->link[1][r];    // r -> r - 1
    const auto post = [&](ThreadGroup::Link* l, const void* src, size_t bytes) {
        if (!g->cv.wait_until(lk, deadline, [&] { return !l->full; })) return false;   // (the previous message of this link was never taken)
        l->src = src; l->bytes = bytes; l->full = true;
        return true;
    };
    bool timed_out = false;
    if (!st && sh) timed_out = !post(up, send_hi, bytes_send_hi) || timed_out;
    if (!st && sl) timed_out = !post(down, send_lo, bytes_send_lo) || timed_out;
    g->cv.notify_all();
    // a message is taken only by a receive of ITS size: any other pairing waits (for ever on RCCL; here until the deadline)
    const auto take = [&](ThreadGroup::Link* l, void* dst, size_t bytes) {
        if (!g->cv.wait_until(lk, deadline, [&] { return l->full && l->bytes == bytes; })) return false;
        const void* src = l->src;
        lk.unlock();
        int e = hip_status(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s), "kfx_comm(threads, p2p) exchange");
        if (!e) e = hip_status(hipStreamSynchronize(s), "kfx_comm(threads, p2p) exchange");
        lk.lock();
        if (e && !st) st = e;
        l->full = false;   // the sender may go on (its buffer has been read)
        g->cv.notify_all();
        return true;
    };
    if (!st && !timed_out && rl) timed_out = !take(&g->link[0][r - 1], recv_lo, bytes_recv_lo) || timed_out;   // what rank - 1 sends upwards
    if (!st && !timed_out && rh) timed_out = !take(&g->link[1][r + 1], recv_hi, bytes_recv_hi) || timed_out;   // what rank + 1 sends downwards
    // nobody reuses a send buffer before its reader is done
    if (!st && !timed_out && sh) timed_out = !g->cv.wait_until(lk, deadline, [&] { return !up->full || up->src != send_hi; }) || timed_out;
    if (!st && !timed_out && sl) timed_out = !g->cv.wait_until(lk, deadline, [&] { return !down->full || down->src != send_lo; }) || timed_out;
    if (timed_out) {
        // withdraw what was posted and never taken, so that a later exchange of this link starts clean
        if (sh && up->full && up->src == send_hi) up->full = false;
        if (sl && down->full && down->src == send_lo) down->full = false;
        g->cv.notify_all();
        return set_error(KFX_E_TIMEOUT, "kfx_comm(threads, p2p) exchange: a leg found no partner of its size in time (on RCCL this rank would hang: the two sides of a leg must derive the same byte count)");
    }
    return st;
}

static int threads_exchange_v_any(kfx_comm* c, const void* send_lo, size_t bsl, void* recv_lo, size_t brl, const void* send_hi, size_t bsh, void* recv_hi,
                                  size_t brh, kfx_stream stream)
{
    if (static_cast<ThreadGroup*>(c->impl)->p2p) return threads_exchange_v_p2p(c, send_lo, bsl, recv_lo, brl, send_hi, bsh, recv_hi, brh, stream);
    return threads_exchange_v(c, send_lo, bsl, recv_lo, brl, send_hi, bsh, recv_hi, brh, stream);
}

static int threads_exchange(kfx_comm* c, const void* send_lo, void* recv_lo, size_t bytes_lo, const void* send_hi, void* recv_hi,
                            size_t bytes_hi, kfx_stream stream)
{
    return threads_exchange_v_any(c, send_lo, bytes_lo, recv_lo, bytes_lo, send_hi, bytes_hi, recv_hi, bytes_hi, stream);
}

static int threads_broadcast(kfx_comm* c, void* buf, size_t bytes, int root, kfx_stream stream)
{
    ThreadGroup* g = static_cast<ThreadGroup*>(c->impl);
    int st = 0;
    if (root < 0 || root >= g->world) st = set_error(KFX_E_RANGE, "kfx_comm broadcast: root");
    if (!st && !buf && bytes) st = set_error(KFX_E_NULL, "kfx_comm(threads) broadcast: null buffer");
    if (!st) st = hip_status(hipStreamSynchronize((hipStream_t)stream), "kfx_comm(threads) broadcast"); // the root's producers are done
    const int par = (int)(g->seq[c->rank]++ & 1u);
    g->err[par][c->rank].store(st);
    g->buf[c->rank] = buf;
    g->wait_all();
    if (!st && c->rank != root && bytes && g->err[par][root].load() == 0) {
        st = hip_status(hipMemcpyAsync(buf, g->buf[root], bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream), "kfx_comm(threads) broadcast");
        if (!st) st = hip_status(hipStreamSynchronize((hipStream_t)stream), "kfx_comm(threads) broadcast");
        if (st) g->err[par][c->rank].store(st);
    }
    g->wait_all(); // the root keeps its buffer untouched until every reader is done
    return group_status(g, par);
}

// all_to_all (gather = false): recv chunk r <- chunk `rank` of rank r's send; all_gather (gather = true): recv chunk r <- rank r's send
static int threads_mesh(kfx_comm* c, const void* send, void* recv, size_t bytes, bool gather, kfx_stream stream)
{
    ThreadGroup* g = static_cast<ThreadGroup*>(c->impl);
    const int r = c->rank;
    int st = 0;
    if ((!send || !recv) && bytes) st = set_error(KFX_E_NULL, "kfx_comm(threads) all_to_all / all_gather: null buffer");
    if (!st) st = hip_status(hipStreamSynchronize((hipStream_t)stream), "kfx_comm(threads) all_to_all / all_gather"); // this rank's producers are done
    const int par = (int)(g->seq[r]++ & 1u);
    g->err[par][r].store(st);
    g->send_lo[r] = send; g->bytes_lo[r] = bytes;   // (the exchange's slots: one collective at a time uses the group)
    g->wait_all();
    hipStream_t s = (hipStream_t)stream;
    for (int k = 0; k < g->world && !st && bytes; ++k) {
        const int from = (r + k) % g->world;
        if (g->err[par][from].load() != 0) continue;
        if (g->bytes_lo[from] != bytes) { st = set_error(KFX_E_SHAPE, "kfx_comm all_to_all / all_gather: ranks disagree on the byte count"); break; }
        const unsigned char* src = static_cast<const unsigned char*>(g->send_lo[from]) + (gather ? 0 : (size_t)r * bytes);
        st = hip_status(hipMemcpyAsync(static_cast<unsigned char*>(recv) + (size_t)from * bytes, src, bytes, hipMemcpyDeviceToDevice, s),
                        "kfx_comm(threads) all_to_all / all_gather");
    }
    if (!st) st = hip_status(hipStreamSynchronize(s), "kfx_comm(threads) all_to_all / all_gather");
    if (st) g->err[par][r].store(st);
    g->wait_all(); // nobody reuses a send buffer before its readers are done
    return group_status(g, par);
}

static int threads_all_to_all(kfx_comm* c, const void* send, void* recv, size_t bytes, kfx_stream stream) { return threads_mesh(c, send, recv, bytes, false, stream); }
static int threads_all_gather(kfx_comm* c, const void* send, void* recv, size_t bytes, kfx_stream stream) { return threads_mesh(c, send, recv, bytes, true, stream); }

static int threads_barrier(kfx_comm* c)
{
    static_cast<ThreadGroup*>(c->impl)->wait_all();
    return 0;
}

static void threads_destroy(kfx_comm* c)
{
    if (c && c->impl && c->rank == 0) delete static_cast<ThreadGroup*>(c->impl);
    if (c) c->impl = nullptr;
}

static int threads_dup(kfx_comm* c, kfx_comm* out);

static void threads_fill(kfx_comm* c, ThreadGroup* g, int rank)
{
    c->rank = rank;
    c->world = g->world;
    c->impl = g;
    c->all_reduce = threads_all_reduce;
    c->exchange = threads_exchange;
    c->barrier = threads_barrier;
    c->destroy = threads_destroy;
    c->broadcast = threads_broadcast;
    c->all_to_all = threads_all_to_all;
    c->all_gather = threads_all_gather;
    c->exchange_v = threads_exchange_v_any;
    c->flags = KFX_COMM_HOST_BLOCKING;   // a collective returns when every rank has entered it
    c->dup = threads_dup;
}

// A second group over the same threads (every rank calls; rank 0 destroys the copy like the original: through its own table)
static int threads_dup(kfx_comm* c, kfx_comm* out)
{
    if (!c || !out || !c->impl) return set_error(KFX_E_NULL, "kfx_comm(threads) dup: null argument");
    ThreadGroup* g = static_cast<ThreadGroup*>(c->impl);
    if (c->rank == 0) {
        ThreadGroup* n = new (std::nothrow) ThreadGroup;
        if (n) { n->world = g->world; n->p2p = g->p2p; n->timeout_ms = g->timeout_ms; }
        g->dup_out = n;
    }
    g->wait_all();
    ThreadGroup* n = g->dup_out;
    g->wait_all();   // (everybody has read it before a later dup overwrites it)
    if (!n) return set_error(KFX_E_RANGE, "kfx_comm(threads) dup: out of memory");
    threads_fill(out, n, c->rank);
    return 0;
}

} // namespace kfx

using namespace kfx;

static int create_threads(kfx_comm* comms, int world, bool p2p, int timeout_ms)
{
    if (!comms) return set_error(KFX_E_NULL, "kfx_comm_create_threads: null comms");
    if (world < 1 || world > MAX_THREAD_RANKS) return set_error(KFX_E_RANGE, "kfx_comm_create_threads: world in [1, 16]");
    ThreadGroup* g = new (std::nothrow) ThreadGroup;
    if (!g) return set_error(KFX_E_RANGE, "kfx_comm_create_threads: out of memory");
    g->world = world;
    g->p2p = p2p;
    g->timeout_ms = timeout_ms;
    for (int r = 0; r < world; ++r) threads_fill(&comms[r], g, r);
    return 0;
}

extern "C" int kfx_comm_create_threads(kfx_comm* comms, int world) { return create_threads(comms, world, false, 0); }
extern "C" int kfx_comm_create_threads_p2p(kfx_comm* comms, int world, int timeout_ms) { return create_threads(comms, world, true, timeout_ms); }

// ---- loop-back transport (kfx_slab.h): one rank of a `world`-rank job measured by itself ------------------------------
namespace {
int loop_copy(void* dst, const void* src, size_t bytes, kfx_stream stream)
{
    if (!bytes || !dst || !src || dst == src) return 0;
    return hip_status(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream), "kfx_comm(loopback)");
}
int loop_all_reduce(kfx_comm*, void*, size_t, int, kfx_stream) { return 0; }   // nothing arrives: the buffer is the "sum"
int loop_exchange_v(kfx_comm* c, const void* send_lo, size_t bsl, void* recv_lo, size_t brl, const void* send_hi, size_t bsh, void* recv_hi,
                    size_t brh, kfx_stream stream)
{
    // what would arrive from below is as large as what this rank sends up, and vice versa: the own buffers come back
    if (c->rank > 0) if (int e = loop_copy(recv_lo, send_hi ? send_hi : send_lo, brl < (send_hi ? bsh : bsl) ? brl : (send_hi ? bsh : bsl), stream)) return e;
    if (c->rank + 1 < c->world) if (int e = loop_copy(recv_hi, send_lo ? send_lo : send_hi, brh < (send_lo ? bsl : bsh) ? brh : (send_lo ? bsl : bsh), stream)) return e;
    return 0;
}
int loop_exchange(kfx_comm* c, const void* send_lo, void* recv_lo, size_t bytes_lo, const void* send_hi, void* recv_hi, size_t bytes_hi, kfx_stream stream)
{
    return loop_exchange_v(c, send_lo, bytes_lo, recv_lo, bytes_lo, send_hi, bytes_hi, recv_hi, bytes_hi, stream);
}
int loop_barrier(kfx_comm*) { return hip_status(hipDeviceSynchronize(), "kfx_comm(loopback) barrier"); }
void loop_destroy(kfx_comm* c) { if (c) c->impl = nullptr; }
int loop_broadcast(kfx_comm*, void*, size_t, int, kfx_stream) { return 0; }
int loop_all_to_all(kfx_comm* c, const void* send, void* recv, size_t bytes, kfx_stream stream) { return loop_copy(recv, send, bytes * (size_t)c->world, stream); }
int loop_all_gather(kfx_comm* c, const void* send, void* recv, size_t bytes, kfx_stream stream)
{
    for (int r = 0; r < c->world; ++r)
        if (int e = loop_copy(static_cast<unsigned char*>(recv) + (size_t)r * bytes, send, bytes, stream)) return e;
    return 0;
}
} // namespace

extern "C" int kfx_comm_create_loopback(kfx_comm* comm, int rank, int world)
{
    if (!comm) return set_error(KFX_E_NULL, "kfx_comm_create_loopback: null comm");
    if (world < 1 || rank < 0 || rank >= world) return set_error(KFX_E_RANGE, "kfx_comm_create_loopback: 0 <= rank < world");
    comm->rank = rank; comm->world = world; comm->impl = nullptr;
    comm->all_reduce = loop_all_reduce; comm->exchange = loop_exchange; comm->barrier = loop_barrier; comm->destroy = loop_destroy;
    comm->broadcast = loop_broadcast; comm->all_to_all = loop_all_to_all; comm->all_gather = loop_all_gather; comm->exchange_v = loop_exchange_v;
    comm->flags = 0;
    comm->dup = [](kfx_comm* c, kfx_comm* out) -> int { return kfx_comm_create_loopback(out, c->rank, c->world); };
    return 0;
}

extern "C" int kfx_slab_layout_init(kfx_slab_layout* L, size_t full_d, float full_zmin, float full_zmax, int rank, int world, int ghost)
{
    if (!L) return set_error(KFX_E_NULL, "kfx_slab_layout_init: null layout");
    if (full_d < 2 || world < 1 || rank < 0 || rank >= world || ghost < 1 || (size_t)world > full_d)
        return set_error(KFX_E_RANGE, "kfx_slab_layout_init: need full_d >= 2, 0 <= rank < world <= full_d, ghost >= 1");
    const size_t base = full_d / (size_t)world, rem = full_d % (size_t)world;
    if (world > 1 && base < (size_t)ghost) return set_error(KFX_E_RANGE, "kfx_slab_layout_init: every rank must own at least `ghost` planes");
    L->full_d = full_d;
    L->full_zmin = full_zmin;
    L->full_zmax = full_zmax;
    L->rank = rank;
    L->world = world;
    L->ghost = ghost;
    const size_t r = (size_t)rank;
    L->z0 = r * base + (r < rem ? r : rem);
    L->z1 = L->z0 + base + (r < rem ? 1 : 0);
    L->s0 = L->z0 >= (size_t)ghost ? L->z0 - (size_t)ghost : 0;
    L->s1 = L->z1 + (size_t)ghost <= full_d ? L->z1 + (size_t)ghost : full_d;
    // BoundedVolume::VoxelPositionInUnits (BoundedVolume.h:115-125): min + size * i / (float)(d - 1)
    const float size_z = full_zmax - full_zmin;
    L->local_zmin = full_zmin + size_z * (float)L->s0 / (float)(full_d - 1);
    L->local_zmax = full_zmin + size_z * (float)(L->s1 - 1) / (float)(full_d - 1);
    return 0;
}

extern "C" int kfx_slab_broadcast_inputs(const kfx_image* depth, const kfx_image* norm, void* scratch, int root, kfx_comm* comm, kfx_stream stream)
{
    if (!depth || !norm || !comm || !depth->ptr || !norm->ptr) return set_error(KFX_E_NULL, "kfx_slab_broadcast_inputs: null argument");
    if (norm->w != depth->w || norm->h != depth->h) return set_error(KFX_E_SHAPE, "kfx_slab_broadcast_inputs: depth and normals differ in size");
    if (depth->pitch < depth->w * 4 || norm->pitch < norm->w * 16) return set_error(KFX_E_SHAPE, "kfx_slab_broadcast_inputs: image pitch");
    if (comm->world == 1) return 0;
    if (!comm->broadcast) return set_error(KFX_E_NULL, "kfx_slab_broadcast_inputs: the transport has no broadcast");
    const size_t w = depth->w, h = depth->h;
    if (w == 0 || h == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool dense = depth->pitch == w * 4 && norm->pitch == w * 16;
    if (dense) {
        if (int e = comm->broadcast(comm, depth->ptr, w * h * 4, root, stream)) return e;
        return comm->broadcast(comm, norm->ptr, w * h * 16, root, stream);
    }
    if (!scratch) return set_error(KFX_E_NULL, "kfx_slab_broadcast_inputs: pitched images need the scratch buffer");
    unsigned char* sd = static_cast<unsigned char*>(scratch);
    unsigned char* sn = sd + w * h * 4;
    if (comm->rank == root) {
        if (int e = hip_status(hipMemcpy2DAsync(sd, w * 4, depth->ptr, depth->pitch, w * 4, h, hipMemcpyDeviceToDevice, s), "kfx_slab_broadcast_inputs")) return e;
        if (int e = hip_status(hipMemcpy2DAsync(sn, w * 16, norm->ptr, norm->pitch, w * 16, h, hipMemcpyDeviceToDevice, s), "kfx_slab_broadcast_inputs")) return e;
    }
    if (int e = comm->broadcast(comm, sd, w * h * 20, root, stream)) return e;
    if (comm->rank != root) {
        if (int e = hip_status(hipMemcpy2DAsync(depth->ptr, depth->pitch, sd, w * 4, w * 4, h, hipMemcpyDeviceToDevice, s), "kfx_slab_broadcast_inputs")) return e;
        if (int e = hip_status(hipMemcpy2DAsync(norm->ptr, norm->pitch, sn, w * 16, w * 16, h, hipMemcpyDeviceToDevice, s), "kfx_slab_broadcast_inputs")) return e;
    }
    return 0;
}

extern "C" int kfx_slab_exchange_halos(const kfx_volume* local, const kfx_slab_layout* L, kfx_comm* comm, kfx_stream stream)
{
    if (!local || !local->ptr || !L || !comm) return set_error(KFX_E_NULL, "kfx_slab_exchange_halos: null argument");
    if (local->d != L->s1 - L->s0 || comm->rank != L->rank || comm->world != L->world)
        return set_error(KFX_E_SHAPE, "kfx_slab_exchange_halos: volume / communicator do not match the layout");
    if (L->world == 1) return 0;
    unsigned char* base = static_cast<unsigned char*>(local->ptr);
    const size_t plane = local->img_pitch;
    const size_t lo_ghost = L->z0 - L->s0, hi_ghost = L->s1 - L->z1;   // planes [s0, z0) come from rank - 1, [z1, s1) from rank + 1
    const size_t own0 = L->z0 - L->s0, own1 = L->z1 - L->s0;           // local indices of the owned range
    // the lower neighbour's upper ghost is `ghost` planes unless it is clipped by the volume's end -- it never is for
    // rank - 1 < world - 1; symmetric for the upper neighbour: both directions carry lo_ghost / hi_ghost planes
    const void* send_lo = L->rank > 0 ? base + own0 * plane : nullptr;                 // my first owned planes -> rank - 1's upper ghost
    void* recv_lo = L->rank > 0 ? base : nullptr;
    const void* send_hi = L->rank + 1 < L->world ? base + (own1 - hi_ghost) * plane : nullptr; // my last owned planes -> rank + 1's lower ghost
    void* recv_hi = L->rank + 1 < L->world ? base + own1 * plane : nullptr;
    return comm->exchange(comm, send_lo, recv_lo, L->rank > 0 ? lo_ghost * plane : 0, send_hi, recv_hi,
                          L->rank + 1 < L->world ? hi_ghost * plane : 0, stream);
}

extern "C" int kfx_slab_composite(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, long long* key, float* payload,
                                  kfx_comm* comm, kfx_stream stream)
{
    if (!comm || !depth) return set_error(KFX_E_NULL, "kfx_slab_composite: null argument");
    if (comm->world == 1) return 0;
    const size_t n = depth->w * depth->h;
    if (int e = kfx_composite_pack(depth, norm, img, key, comm->rank, stream)) return e;
    if (int e = comm->all_reduce(comm, key, n, KFX_COMM_MIN_I64, stream)) return e;
    if (int e = kfx_composite_select(depth, norm, img, key, payload, comm->rank, stream)) return e;
    if (int e = comm->all_reduce(comm, payload, KFX_COMPOSITE_PAYLOAD * n, KFX_COMM_SUM_F32, stream)) return e;
    return kfx_composite_unpack(depth, norm, img, key, payload, stream);
}

extern "C" size_t kfx_slab_composite_direct_scratch_bytes(size_t w, size_t h, int world)
{
    if (world < 1) return 0;   // send strips, received / gathered strips, this rank's merged strip
    return (2 * (size_t)world + 1) * KFX_COMPOSITE_STRIP_PLANES * kfx_composite_strip_pixels(w, h, world) * sizeof(float);
}

extern "C" int kfx_slab_composite_direct(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, void* scratch, kfx_comm* comm,
                                         kfx_stream stream)
{
    if (!comm || !depth) return set_error(KFX_E_NULL, "kfx_slab_composite_direct: null argument");
    if (comm->world == 1) return 0;
    // argument errors every rank makes alike come back before any collective; from here on a LOCAL failure (a launch error) must
    // not keep this rank out of the two collectives its peers enter -- they would wait for ever (round-4 advice): it is
    // remembered, the collectives are entered all the same, and the first error is returned at the end
    if (!scratch) return set_error(KFX_E_NULL, "kfx_slab_composite_direct: null scratch");
    if (!comm->all_to_all || !comm->all_gather) return set_error(KFX_E_RANGE, "kfx_slab_composite_direct: the transport has no all_to_all / all_gather");
    const int W = comm->world;
    const size_t S = kfx_composite_strip_pixels(depth->w, depth->h, W), strip = KFX_COMPOSITE_STRIP_PLANES * S;
    float* send = static_cast<float*>(scratch);
    float* recv = send + (size_t)W * strip;
    float* merged = recv + (size_t)W * strip;
    int status = 0;
    auto note = [&](int e) { if (e && !status) status = e; };
    note(kfx_composite_strips_pack(depth, norm, img, send, 0, W, stream));
    note(comm->all_to_all(comm, send, recv, strip * sizeof(float), stream));
    note(kfx_composite_strips_merge(recv, merged, S, 0, W, stream));
    note(comm->all_gather(comm, merged, recv, strip * sizeof(float), stream));
    if (status) return status;
    return kfx_composite_strips_unpack(depth, norm, img, recv, 0, W, stream);
}

extern "C" size_t kfx_slab_exact_scratch_bytes(size_t w, size_t h) { return (17 * w * h + 64) * sizeof(int); }

static int exact_args(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, float* state, void* scratch,
                      const kfx_volume* local, const kfx_slab_layout* L, kfx_comm* comm, const char* who)
{
    if (!depth || !norm || !img || !state || !scratch || !local || !L || !comm) return set_error(KFX_E_NULL, who);
    if (local->d != L->s1 - L->s0 || comm->rank != L->rank || comm->world != L->world)
        return set_error(KFX_E_SHAPE, "kfx_slab_raycast_exact: volume / communicator do not match the layout");
    return 0;
}

// The blueprint's hand-over: world + 1 march stages with a neighbour exchange of the march planes between
// them -- a ray that enters at one end needs `world` stages to reach the other, one more lets a hit on a slab boundary have
// its normal evaluated by the neighbour that owns the gradient's base plane --, then ONE all-reduce of the finalising ranks'
// results and ONE read-back (the count of rays without a final status: zero).  No host synchronisation between the stages.
extern "C" int kfx_slab_raycast_exact(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, float* state, void* scratch,
                                      const kfx_volume* local, const kfx_slab_layout* L, const float T_wc[12], const float K[4],
                                      float near, float far, float trunc_dist, int subpix, kfx_comm* comm, kfx_stream stream,
                                      int* rounds_out)
{
    if (int e = exact_args(depth, norm, img, state, scratch, local, L, comm, "kfx_slab_raycast_exact: null argument")) return e;
    const int w = (int)depth->w, h = (int)depth->h;
    const size_t n = (size_t)w * h;
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int* from_lo = static_cast<int*>(scratch);
    int* from_hi = from_lo + 5 * n;
    int* fin = from_hi + 5 * n;
    int* contrib = fin + n;
    int* open = contrib + 6 * n;
    int* istate = reinterpret_cast<int*>(state);
    const kfx_slab slab = {L->full_d, L->s0, L->full_zmin, L->full_zmax};
    const dim3 grid((unsigned)((n + 255) / 256));
    const int world = comm->world, rank = comm->rank;
    const int stages = world > 1 ? world + 1 : 1;
    // a local failure must not keep this rank out of a collective its peers enter: remember it, go on, report it at the end
    int status = 0;
    auto note = [&](int e) { if (e && !status) status = e; };
    if (world > 1) note(hip_status(hipMemsetAsync(fin, 0, (7 * n + 1) * sizeof(int), s), "kfx_slab_raycast_exact")); // fin, contrib, open
    for (int stage = 0; stage < stages; ++stage) {
        note(kfx_raycast_sdf_slab(state, stage == 0, local, &slab, (int)L->z0, (int)L->z1, w, h, T_wc, K, near, far, trunc_dist, subpix, stream));
        if (world == 1) break;
        const bool exchange = stage + 1 < stages;
        if (exchange) note(comm->exchange(comm, istate, from_lo, 5 * n * sizeof(int), istate, from_hi, 5 * n * sizeof(int), stream));
        hipLaunchKernelGGL(k_handover_merge, grid, dim3(256), 0, s, istate, (exchange && rank > 0) ? from_lo : nullptr,
                           (exchange && rank + 1 < world) ? from_hi : nullptr, fin, n, (stage == 0 && rank == 0) ? 1 : 0);
        note(check_launch("kfx_slab_raycast_exact"));
    }
    if (world > 1) {
        hipLaunchKernelGGL(k_handover_contrib, grid, dim3(256), 0, s, istate, fin, contrib, n);
        note(check_launch("kfx_slab_raycast_exact"));
        note(comm->all_reduce(comm, contrib, 6 * n, KFX_COMM_SUM_I32, stream));
        hipLaunchKernelGGL(k_handover_finish, grid, dim3(256), 0, s, istate, contrib, n, open);
        note(check_launch("kfx_slab_raycast_exact"));
        int n_open = 0;
        note(hip_status(hipMemcpyAsync(&n_open, open, sizeof(int), hipMemcpyDeviceToHost, s), "kfx_slab_raycast_exact"));
        note(hip_status(hipStreamSynchronize(s), "kfx_slab_raycast_exact"));   // the frame's one synchronisation
        if (!status && n_open) status = set_error(KFX_E_RANGE, "kfx_slab_raycast_exact: rays without a final status after world + 1 stages");
    }
    if (rounds_out) *rounds_out = stages;
    if (status) return status;
    return kfx_raycast_state_to_images(depth, norm, img, state, stream);
}

// The cross-check of the hand-over: rounds of kfx_raycast_sdf_slab with one SUM all-reduce of the touched pixels' march state
// per round (every rank always holds every ray's state) and a host-side termination test after each -- <= world + 2 rounds,
// each a host synchronisation.  Same images, bit for bit.
extern "C" int kfx_slab_raycast_exact_allreduce(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, float* state, void* scratch,
                                      const kfx_volume* local, const kfx_slab_layout* L, const float T_wc[12], const float K[4],
                                      float near, float far, float trunc_dist, int subpix, kfx_comm* comm, kfx_stream stream,
                                      int* rounds_out)
{
    if (int e = exact_args(depth, norm, img, state, scratch, local, L, comm, "kfx_slab_raycast_exact_allreduce: null argument")) return e;
    const int w = (int)depth->w, h = (int)depth->h;
    const size_t n = (size_t)w * h;
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int* contrib = static_cast<int*>(scratch);
    int* flag = contrib + 5 * n;
    int* istate = reinterpret_cast<int*>(state);
    const kfx_slab slab = {L->full_d, L->s0, L->full_zmin, L->full_zmax};
    const dim3 grid((unsigned)((n + 255) / 256));
    int rounds = 0;
    for (;;) {
        if (int e = kfx_raycast_sdf_slab(state, rounds == 0, local, &slab, (int)L->z0, (int)L->z1, w, h, T_wc, K, near, far,
                                         trunc_dist, subpix, stream))
            return e;
        ++rounds;
        if (comm->world > 1) {
            hipLaunchKernelGGL(k_state_contrib, grid, dim3(256), 0, s, istate, contrib, n);
            if (int e = check_launch("kfx_slab_raycast_exact")) return e;
            if (int e = comm->all_reduce(comm, contrib, 5 * n, KFX_COMM_SUM_I32, stream)) return e;
        }
        if (int e = hip_status(hipMemsetAsync(flag, 0, sizeof(int), s), "kfx_slab_raycast_exact")) return e;
        hipLaunchKernelGGL(k_state_merge, grid, dim3(256), 0, s, istate, contrib, n, comm->world > 1 ? 1 : 0, flag);
        if (int e = check_launch("kfx_slab_raycast_exact")) return e;
        int open = 0;
        if (int e = hip_status(hipMemcpyAsync(&open, flag, sizeof(int), hipMemcpyDeviceToHost, s), "kfx_slab_raycast_exact")) return e;
        if (int e = hip_status(hipStreamSynchronize(s), "kfx_slab_raycast_exact")) return e;
        if (!open) break; // the merged state is identical on every rank, so all ranks leave together
        if (rounds > L->world + 3) return set_error(KFX_E_RANGE, "kfx_slab_raycast_exact: march did not terminate");
    }
    if (comm->world > 1) // normals (planes 5-7) and shade (8): written by one rank per pixel
        if (int e = comm->all_reduce(comm, istate + 5 * n, 4 * n, KFX_COMM_SUM_I32, stream)) return e;
    if (rounds_out) *rounds_out = rounds;
    return kfx_raycast_state_to_images(depth, norm, img, state, stream);
}

// ---- the hand-over pipelined over image row-tiles (kfx_slab.h) -------------------------------------------------------------
namespace {
struct TiledScratch {
    size_t P, n;       // plane stride of a tile (pixels), pixels of the image
    int R, T;          // rows per tile, tiles
    int *M, *Rz, *from_lo, *from_hi, *fin;
    kfx::ExactFinalBufs own;   // the final exchange's buffers inside the scratch (set 0)
    size_t S;          // pixels per image strip of the final exchange (0 for one rank)
};
// sizes (ints) of the final exchange's four buffers: dense [6][n] for the all-reduce (and for one rank), by strips [world][6][S]
// for the direct sends
struct FinalSizes { size_t contrib, gathered, mine, open, S; };
FinalSizes final_sizes(size_t w, size_t h, int world)
{
    FinalSizes z;
    const size_t n = w * h;
    z.S = world > 1 ? kfx_composite_strip_pixels(w, h, world) : 0;
    const size_t by_strips = (size_t)(world > 1 ? world : 0) * 6 * z.S;
    z.contrib = ((by_strips > 6 * n ? by_strips : 6 * n) + 63) / 64 * 64;
    z.gathered = (by_strips + 63) / 64 * 64;
    z.mine = (6 * z.S + 63) / 64 * 64;
    z.open = 64;
    return z;
}
// the scratch buffer's parts; returns its size in ints (scratch may be null: sizes only)
size_t tiled_layout(TiledScratch& t, void* scratch, size_t w, size_t h, int tiles, int world)
{
    if (tiles > 64) tiles = 64;   // (the march keeps a 64-bit mask of the tiles it has initialised; clamped HERE so that the advertised size and the march agree)
    t.T = tiles < 1 ? 1 : (tiles > (int)h ? (int)(h ? h : 1) : tiles);
    t.R = (int)((h + (size_t)t.T - 1) / (size_t)t.T);
    t.T = (int)((h + (size_t)t.R - 1) / (size_t)(t.R ? t.R : 1));   // (tiles that are not empty)
    t.P = ((size_t)t.R * w + 63) / 64 * 64;
    t.n = w * h;
    int* base = static_cast<int*>(scratch);
    size_t o = 0;
    const auto take = [&](size_t ints) { int* q = base ? base + o : nullptr; o += ints; return q; };
    t.M = take((size_t)t.T * 4 * t.P);
    t.Rz = take((size_t)t.T * 4 * t.P);
    t.from_lo = take((size_t)t.T * 4 * t.P);
    t.from_hi = take((size_t)t.T * 4 * t.P);
    t.fin = take((t.n + 63) / 64 * 64);
    const FinalSizes z = final_sizes(w, h, world);
    t.S = z.S;
    t.own.contrib = take(z.contrib);
    t.own.gathered = take(z.gathered);
    t.own.mine = take(z.mine);
    t.own.open = take(z.open);
    return o;
}
bool finalise_by_allreduce()
{
    static const bool v = [] { const char* e = getenv("KFX_SLAB_FINALISE"); return e && e[0] == 'a'; }();
    return v;
}
} // namespace

extern "C" size_t kfx_slab_exact_tiled_scratch_bytes(size_t w, size_t h, int tiles, int world)
{
    if (w == 0 || h == 0) return 256;
    TiledScratch t;
    return tiled_layout(t, nullptr, w, h, tiles, world < 1 ? 1 : world) * sizeof(int);
}

// One more set of the final exchange's buffers (kfx_slab_frame keeps several frames' final exchanges in flight: slab_internal.h)
size_t kfx::exact_final_bytes(size_t w, size_t h, int world)
{
    const FinalSizes z = final_sizes(w, h, world < 1 ? 1 : world);
    return (z.contrib + z.gathered + z.mine + z.open) * sizeof(int);
}
void kfx::exact_final_carve(ExactFinalBufs& b, void* mem, size_t w, size_t h, int world)
{
    const FinalSizes z = final_sizes(w, h, world < 1 ? 1 : world);
    int* q = static_cast<int*>(mem);
    b.contrib = q; q += z.contrib;
    b.gathered = q; q += z.gathered;
    b.mine = q; q += z.mine;
    b.open = q;
}

static int exact_tiled_args(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, void* scratch, const kfx_volume* local,
                            const kfx_slab_layout* L, kfx_comm* comm, const float* T_wc, const float* K)
{
    if (!depth || !norm || !img || !scratch || !local || !L || !comm || !T_wc || !K || !depth->ptr || !norm->ptr || !img->ptr)
        return set_error(KFX_E_NULL, "kfx_slab_raycast_exact_tiled: null argument");
    if (local->d != L->s1 - L->s0 || comm->rank != L->rank || comm->world != L->world)
        return set_error(KFX_E_SHAPE, "kfx_slab_raycast_exact_tiled: volume / communicator do not match the layout");
    if (depth->w < img->w || depth->h < img->h || norm->w < img->w || norm->h < img->h || depth->pitch < img->w * 4 || img->pitch < img->w * 4 ||
        norm->pitch < img->w * 16)
        return set_error(KFX_E_SHAPE, "kfx_slab_raycast_exact_tiled: image sizes");
    if ((((uintptr_t)depth->ptr | depth->pitch | (uintptr_t)img->ptr | img->pitch) & 3) || (((uintptr_t)norm->ptr | norm->pitch) & 15) || ((uintptr_t)scratch & 15))
        return set_error(KFX_E_ALIGN, "kfx_slab_raycast_exact_tiled: alignment");
    if (comm->world > 1 && !comm->exchange_v) return set_error(KFX_E_RANGE, "kfx_slab_raycast_exact_tiled: the transport has no exchange_v");
    return 0;
}

// Ghost planes per side with which the exact hand-over needs no last stage for the normals.  A hit is found at a sample whose base
// plane the finder owns; its sub-step interpolation (cu_raycast.cu:70-74) puts it at most one step behind that sample, a step is at
// most max(trunc_dist, voxel.x) (the march's step is the sampled value, and a TSDF's cells do not exceed the truncation distance:
// Sdf.h, cu_sdffusion.cu:49), a ray moves |ray| <= the longest ((u - u0) / fu, (v - v0) / fv, 1) of the image per unit of lambda;
// the gradient stencil (Volume.h:262-289) takes one plane below and one above its base plane.
extern "C" int kfx_slab_exact_ghost(size_t full_d, float full_zmin, float full_zmax, float size_x, size_t vol_w, float trunc_dist, const float K[4], int w, int h)
{
    if (!K || full_d < 2 || vol_w < 2 || w < 1 || h < 1) return 2;
    const double vz = std::fabs(((double)full_zmax - (double)full_zmin) / (double)(full_d - 1)), vx = std::fabs((double)size_x / (double)(vol_w - 1));
    double rmax = 1.0;
    for (int c = 0; c < 4; ++c) {
        const double cx = ((c & 1 ? (double)(w - 1) : 0.0) - (double)K[2]) / (double)K[0], cy = ((c & 2 ? (double)(h - 1) : 0.0) - (double)K[3]) / (double)K[1];
        const double r = std::sqrt(cx * cx + cy * cy + 1.0);
        rmax = r > rmax ? r : rmax;
    }
    const double step = (double)trunc_dist > vx ? (double)trunc_dist : vx;
    const double planes = rmax * step / vz;
    if (!(trunc_dist > 0.f) || !(vz > 0.0) || !(planes < 1e6)) return 2;   // (degenerate: the ordinary width, with the last stage)
    return (int)std::ceil(planes * 1.001 + 0.01) + 2;
}

namespace {
std::atomic<int> g_normals_stage{-1};   // -1: not read yet (KFX_SLAB_NORMALS_STAGE, default 0)
bool normals_stage_forced()
{
    int v = g_normals_stage.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("KFX_SLAB_NORMALS_STAGE");
        v = (e && atoi(e) != 0) ? 1 : 0;
        g_normals_stage.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
} // namespace

// 1: the hand-over keeps its last stage whatever the layout's ghost width (measurements, cross-checks; every rank alike); 0: dropped
// where the ghost planes allow.  Returns the previous setting.
extern "C" int kfx_slab_set_normals_stage(int keep)
{
    const int prev = normals_stage_forced() ? 1 : 0;
    g_normals_stage.store(keep ? 1 : 0, std::memory_order_relaxed);
    return prev;
}

// The march: token steps, the normals' stage, and this rank's contribution to the final images left in `into` (null: the scratch's own
// set).  Everything here is ordered on `stream`; the collectives are neighbour exchanges only (comm->exchange_v).
int kfx::exact_tiled_march(void* scratch, const ExactFinalBufs* into, const kfx_volume* local, const kfx_slab_layout* L, const float T_wc[12],
                           const float K[4], float near, float far, float trunc_dist, int subpix, int tiles, int w, int h, kfx_comm* comm,
                           kfx_stream stream, int* steps_out)
{
    const int world = comm->world, rank = comm->rank;
    TiledScratch t;
    tiled_layout(t, scratch, (size_t)w, (size_t)h, tiles, world);
    const ExactFinalBufs& fb = into ? *into : t.own;
    const int T = t.T, R = t.R;
    // the tile state packed into three planes (12 B per pixel and hop) unless a marching ray's delta could be negative; every rank
    // decides alike from the same arguments
    const int packed = trunc_dist > 0.f ? 1 : 0, NP = packed ? 3 : 4;
    const size_t P = t.P, tile_bytes = (size_t)NP * P * sizeof(int);
    // With ghost planes as wide as a hit can fall back behind the sample that found it (kfx_slab_exact_ghost), the rank that finds a
    // hit always holds its gradient stencil: it finalises the hit itself and the hand-over needs no last stage for the normals
    const int self_normals = (world > 1 && L->ghost >= kfx_slab_exact_ghost(L->full_d, L->full_zmin, L->full_zmax, local->boxmax[0] - local->boxmin[0],
                                                                          (size_t)local->w, trunc_dist, K, w, h) && !normals_stage_forced()) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    const kfx_slab slab = {L->full_d, L->s0, L->full_zmin, L->full_zmax};
    // a local failure must not keep this rank out of a collective its peers enter: remember it, go on, report it at the end
    int status = 0;
    auto note = [&](int e) { if (e && !status) status = e; };
    // one launch per visit of a tile: it initialises the tile's rays when this is the rank's first visit, takes over what the
    // neighbours handed on (their newer, still open snapshots), and marches what lies in the rank's own planes
    auto march = [&](int v0, int v1, int init, const int* from_lo, const int* from_hi, int tile_major) {
        note(kfx_raycast_sdf_slab_tiles(reinterpret_cast<float*>(t.M), reinterpret_cast<float*>(t.Rz), P, R, v0, v1, init, t.fin, rank == 0 ? 1 : 0,
                                        reinterpret_cast<const float*>(from_lo), reinterpret_cast<const float*>(from_hi), (tile_major ? 1 : 0) | (packed ? 2 : 0) | (self_normals ? 4 : 0), local, &slab,
                                        (int)L->z0, (int)L->z1, w, h, T_wc, K, near, far, trunc_dist, subpix, stream));
    };
    auto rows_of = [&](int tile, int& v0, int& v1) { v0 = tile * R; v1 = v0 + R < h ? v0 + R : h; };
    int steps = 0;
    if (world == 1) {
        march(0, h, 1, nullptr, nullptr, 0);
        steps = 1;
    } else {
        // Every rank starts every ray itself, tile by tile at its first visit (the entry slab's owner is the one that can advance
        // it).  Rank r visits tile t when the upward token reaches it (step r + t) and when the downward one does (W - 1 - r + t).
        // Which tokens exist: a ray's z-direction is R_wc's third row against ((u - u0) / fu, (v - v0) / fv, 1) -- affine in the
        // pixel, so its extremes sit in the image's corners.  When every ray rises (the usual case: the camera looks along the
        // slabs' axis) nothing ever travels downwards -- no downward visits, no downward messages -- and the normals' stage sends
        // downwards only (a rising ray's hit falls BACK across a boundary); likewise for all-falling rays.  Every rank derives the
        // same two flags from the same pose (with a margin: rays that are flat within it count as both).
        bool need_up = true, need_down = true;
        {
            double zlo = 1e300, zhi = -1e300, mag = 0.0;
            for (int c = 0; c < 4; ++c) {
                const double cx = ((c & 1 ? (double)(w - 1) : 0.0) - (double)K[2]) / (double)K[0], cy = ((c & 2 ? (double)(h - 1) : 0.0) - (double)K[3]) / (double)K[1];
                const double z = (double)T_wc[8] * cx + (double)T_wc[9] * cy + (double)T_wc[10];
                const double m = std::fabs((double)T_wc[8] * cx) + std::fabs((double)T_wc[9] * cy) + std::fabs((double)T_wc[10]);
                zlo = z < zlo ? z : zlo; zhi = z > zhi ? z : zhi; mag = m > mag ? m : mag;
            }
            const double margin = 1e-4 * mag;
            if (zlo == zlo && zhi == zhi && mag < 1e300) {   // (a pose with NaN keeps both)
                need_down = !(zlo > margin);
                need_up = !(zhi < -margin);
            }
        }
        unsigned long long seen = 0ull;   // tiles this rank has initialised (tiles <= 64)
        bool got_lo = false, got_hi = false;   // what the previous step's exchange brought: tile A (from below) / tile B (from above) of THIS step
        for (int d = 0; d < world + T - 1; ++d, ++steps) {
            const int A = d - rank, B = d - (world - 1 - rank);   // the tiles the upward / downward token brings to this rank now
            const bool a_ok = need_up && A >= 0 && A < T, b_ok = need_down && B >= 0 && B < T;
            int v0, v1;
            if (a_ok) {
                rows_of(A, v0, v1);
                march(v0, v1, (seen >> A) & 1ull ? 0 : 1, got_lo ? t.from_lo : nullptr, (b_ok && B == A && got_hi) ? t.from_hi : nullptr, 0);
                seen |= 1ull << A;
            }
            if (b_ok && !(a_ok && B == A)) {
                rows_of(B, v0, v1);
                march(v0, v1, (seen >> B) & 1ull ? 0 : 1, nullptr, got_hi ? t.from_hi : nullptr, 0);
                seen |= 1ull << B;
            }
            // pass the tokens on: tile A upwards, tile B downwards; what arrives is the tile this rank marches in the next step
            const bool up = a_ok && rank + 1 < world, down = b_ok && rank > 0;
            const bool from_below = need_up && rank > 0 && A + 1 >= 0 && A + 1 < T, from_above = need_down && rank + 1 < world && B + 1 >= 0 && B + 1 < T;
            // (every rank calls the exchange in every step, with empty legs where it has nothing to pass on: a transport may
            //  synchronise its ranks inside the call, as the in-process one does in its barrier mode)
            note(comm->exchange_v(comm, down ? t.M + (size_t)B * NP * P : nullptr, down ? tile_bytes : 0, from_below ? t.from_lo : nullptr, from_below ? tile_bytes : 0,
                                  up ? t.M + (size_t)A * NP * P : nullptr, up ? tile_bytes : 0, from_above ? t.from_hi : nullptr, from_above ? tile_bytes : 0, stream));
            got_lo = from_below; got_hi = from_above;
        }
        // one more stage over the whole image: a hit whose sub-step interpolation fell back across a slab boundary has its normal
        // evaluated by the neighbour that owns the gradient's base plane -- the one BEHIND the ray: rising rays' hits go down, falling
        // rays' up
        // (not with ghost planes wide enough for every finder to finalise its own hits: self_normals.  A ray left without a final
        //  status -- cells beyond the truncation distance -- is counted by the final exchange and reported: KFX_E_RANGE)
        if (!self_normals) {
            const size_t all = (size_t)T * tile_bytes;
            const bool send_down = need_up && rank > 0, recv_from_above = need_up && rank + 1 < world;       // rising rays' pending normals
            const bool send_up = need_down && rank + 1 < world, recv_from_below = need_down && rank > 0;     // falling rays'
            note(comm->exchange_v(comm, send_down ? t.M : nullptr, send_down ? all : 0, recv_from_below ? t.from_lo : nullptr, recv_from_below ? all : 0,
                                  send_up ? t.M : nullptr, send_up ? all : 0, recv_from_above ? t.from_hi : nullptr, recv_from_above ? all : 0, stream));
            march(0, h, 0, recv_from_below ? t.from_lo : nullptr, recv_from_above ? t.from_hi : nullptr, 1);
            ++steps;
        }
    }
    // Every pixel has been given its final status by exactly one rank: this rank's contribution to the final images, in the layout of
    // the exchange that follows (exact_tiled_finalise)
    const bool direct = world > 1 && !finalise_by_allreduce() && comm->all_to_all && comm->all_gather;
    const unsigned S = direct ? (unsigned)t.S : 0u;
    const dim3 grid2(ceil_div(w, 64), ceil_div(h, 4));
    if (direct && (size_t)world * t.S > t.n)   // the last strip's padding travels too: defined (and summed as zero)
        note(hip_status(hipMemsetAsync(fb.contrib + ((size_t)(world - 1) * 6) * t.S, 0, 6 * t.S * sizeof(int), s), "kfx_slab_raycast_exact_tiled"));
    hipLaunchKernelGGL(k_tiles_contrib, grid2, dim3(256), 0, s, t.M, t.Rz, t.fin, fb.contrib, w, h, R, P, S, packed);
    note(check_launch("kfx_slab_raycast_exact_tiled"));
    if (steps_out) *steps_out = steps;   // world + tiles - 1 token steps + the normals' stage (1 for a single rank)
    return status;
}

// The final exchange.  The results reach every rank by direct sends over the mesh, as the composite's strips do (composite.hip):
// rank j owns strip j of the image; one all-to-all brings the ranks' contributions to a strip -- zeros but for the one that
// finalised the pixel -- to its owner, the owner adds them up (integers: NaN and -0 survive), one all-gather returns the strips:
// 24 bytes per pixel cross each link once per phase, every link at once, where an all-reduce of the 7.4 MB (640 x 480) takes 14
// dependent ring steps.  KFX_SLAB_FINALISE=allreduce (or a transport without all_to_all / all_gather) keeps the all-reduce; same
// images.  `comm` and `stream` may be another communicator over the same ranks and another stream than the march's
// (kfx_slab_frame's pipelined frames): nothing here touches the march's planes.  h_open: a host-visible word that receives the
// number of pixels left without a final status (zero) once `stream` has passed.
int kfx::exact_tiled_finalise(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, void* scratch, const ExactFinalBufs* from, int tiles,
                              kfx_comm* comm, kfx_stream stream, int* h_open)
{
    const int world = comm->world;
    const int w = (int)img->w, h = (int)img->h;
    TiledScratch t;
    tiled_layout(t, scratch, (size_t)w, (size_t)h, tiles, world);
    const ExactFinalBufs& fb = from ? *from : t.own;
    hipStream_t s = (hipStream_t)stream;
    int status = 0;
    auto note = [&](int e) { if (e && !status) status = e; };
    note(hip_status(hipMemsetAsync(fb.open, 0, sizeof(int), s), "kfx_slab_raycast_exact_tiled"));
    const bool direct = world > 1 && !finalise_by_allreduce() && comm->all_to_all && comm->all_gather;
    const unsigned S = direct ? (unsigned)t.S : 0u;
    const dim3 grid2(ceil_div(w, 64), ceil_div(h, 4));
    const int* final_planes = fb.contrib;
    if (direct) {
        const size_t words = 6 * t.S;
        note(comm->all_to_all(comm, fb.contrib, fb.gathered, words * sizeof(int), stream));
        hipLaunchKernelGGL(k_strip_sum, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, fb.gathered, fb.mine, words, world);
        note(check_launch("kfx_slab_raycast_exact_tiled"));
        note(comm->all_gather(comm, fb.mine, fb.gathered, words * sizeof(int), stream));
        final_planes = fb.gathered;
    } else if (world > 1) {
        note(comm->all_reduce(comm, fb.contrib, 6 * t.n, KFX_COMM_SUM_I32, stream));
    }
    const OutImages out{(unsigned char*)depth->ptr, (unsigned char*)norm->ptr, (unsigned char*)img->ptr, depth->pitch, norm->pitch, img->pitch, w, h};
    hipLaunchKernelGGL(k_tiles_finish, grid2, dim3(256), 0, s, out, final_planes, fb.open, S);
    note(check_launch("kfx_slab_raycast_exact_tiled"));
    if (h_open) note(hip_status(hipMemcpyAsync(h_open, fb.open, sizeof(int), hipMemcpyDeviceToHost, s), "kfx_slab_raycast_exact_tiled"));
    return status;
}

extern "C" int kfx_slab_raycast_exact_tiled(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, void* scratch,
                                            const kfx_volume* local, const kfx_slab_layout* L, const float T_wc[12], const float K[4],
                                            float near, float far, float trunc_dist, int subpix, int tiles, kfx_comm* comm, kfx_stream stream,
                                            int* h_open, int* steps_out)
{
    if (int e = exact_tiled_args(depth, norm, img, scratch, local, L, comm, T_wc, K)) return e;
    const int w = (int)img->w, h = (int)img->h;
    if (w == 0 || h == 0) return 0;
    // (a local failure of the march does not keep this rank out of the final exchange its peers enter)
    int status = exact_tiled_march(scratch, nullptr, local, L, T_wc, K, near, far, trunc_dist, subpix, tiles, w, h, comm, stream, steps_out);
    int n_open = 0;
    const int e = exact_tiled_finalise(depth, norm, img, scratch, nullptr, tiles, comm, stream, h_open ? h_open : nullptr);
    if (e && !status) status = e;
    if (h_open) return status;
    hipStream_t s = (hipStream_t)stream;
    TiledScratch t;
    tiled_layout(t, scratch, (size_t)w, (size_t)h, tiles, comm->world);
    const int c = hip_status(hipMemcpyAsync(&n_open, t.own.open, sizeof(int), hipMemcpyDeviceToHost, s), "kfx_slab_raycast_exact_tiled");
    if (c && !status) status = c;
    const int y = hip_status(hipStreamSynchronize(s), "kfx_slab_raycast_exact_tiled");
    if (y && !status) status = y;
    if (!status && n_open) status = set_error(KFX_E_RANGE, "kfx_slab_raycast_exact_tiled: rays without a final status after the hand-over");
    return status;
}
