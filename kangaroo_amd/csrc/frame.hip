// frame.hip -- kfx_frame (include/kfx.h): one frame of the reference application's loop
// (applications/kinectfusion/main.cpp:200-356, known poses) enqueued by ONE call.  Host code only: every launch goes through
// the library's own entry points (kfx_bilateral_f32, kfx_depth_to_vbo_normals_f32, kfx_sdf_fuse[_tracked],
// kfx_raycast_sdf[_tracked]), so a step writes exactly what the separate calls write; the point is the host side -- seven
// interpreter-level calls per 0.45 ms frame become one -- and device-event timing of the frame's parts that does not depend
// on the caller's runtime (torch's events see only torch's current stream).
#include <cmath>
#include <cstdlib>
#include <new>

#include "kfx_device.h"

struct kfx_frame {
    kfx_frame_config cfg;
    kfx_sdf_summary* summary;   // owned; created by the first set_track(1)
    int track;
    long long frames;           // steps so far
    int slots;                  // timing ring (frames); 0: no events
    hipEvent_t* ev;             // slots x 4: before preprocess, before SdfFuse, after SdfFuse, after RaycastSdf
    long long* ev_frame;        // frame recorded in each slot, -1: none
    unsigned char* ev_mask;     // which of the four events the frame in each slot recorded
    unsigned mask;              // which events the next steps record (kfx_frame_set_timing)
    // the packed texel image {nx, ny, nz, depth} of the frame (owned): written by the fused vbo / normals launch of a step, staged by
    // LDS-DMA in the SdfFuse of the SAME step (fuse.hip); a step that integrates without preprocessing lets kfx_sdf_fuse pack its own
    kfx_image texels;
};

using namespace kfx;

static int valid_image(const kfx_image& im, size_t elem)
{
    return im.ptr && im.w > 0 && im.h > 0 && im.pitch >= im.w * elem;
}

extern "C" int kfx_frame_create(kfx_frame** out, const kfx_frame_config* cfg)
{
    if (!out || !cfg) return set_error(KFX_E_NULL, "kfx_frame_create: null argument");
    *out = nullptr;
    if (!cfg->vol.ptr) return set_error(KFX_E_NULL, "kfx_frame_create: null volume");
    if (!valid_image(cfg->raw, 4) || !valid_image(cfg->filtered, 4) || !valid_image(cfg->vbo, 16) || !valid_image(cfg->normals, 16) ||
        !valid_image(cfg->ray_depth, 4) || !valid_image(cfg->ray_norm, 16) || !valid_image(cfg->ray_img, 4))
        return set_error(KFX_E_SHAPE, "kfx_frame_create: image views");
    if (cfg->filtered.w != cfg->raw.w || cfg->filtered.h != cfg->raw.h || cfg->vbo.w != cfg->raw.w || cfg->vbo.h != cfg->raw.h ||
        cfg->normals.w != cfg->raw.w || cfg->normals.h != cfg->raw.h)
        return set_error(KFX_E_SHAPE, "kfx_frame_create: the preprocess images differ in size");
    if (cfg->timing_slots < 0 || cfg->timing_slots > (1 << 20)) return set_error(KFX_E_RANGE, "kfx_frame_create: timing_slots");
    kfx_frame* f = new (std::nothrow) kfx_frame;
    if (!f) return set_error(KFX_E_RANGE, "kfx_frame_create: out of memory");
    f->cfg = *cfg;
    f->summary = nullptr;
    f->track = 0;
    f->frames = 0;
    f->slots = cfg->timing_slots;
    f->ev = nullptr;
    f->ev_frame = nullptr;
    f->ev_mask = nullptr;
    f->mask = KFX_FRAME_EVENTS_ALL;
    f->texels = kfx_image{0, nullptr, 0, 0};
    {   // (no device yet -- the argument checks of tests/test_abi_cpu.py run without one -- or no memory: steps pack per call instead)
        const size_t tpitch = (cfg->filtered.w * 16 + 255) / 256 * 256;
        void* buf = nullptr;
        if (tpitch < (1u << 24) && hipMalloc(&buf, kfx::texel_image_bytes(cfg->filtered.w, cfg->filtered.h)) == hipSuccess) f->texels = kfx_image{tpitch, buf, cfg->filtered.w, cfg->filtered.h};
        else (void)hipGetLastError();
    }
    if (f->slots) {
        f->ev = new (std::nothrow) hipEvent_t[(size_t)f->slots * 4];
        f->ev_frame = new (std::nothrow) long long[f->slots];
        f->ev_mask = new (std::nothrow) unsigned char[f->slots];
        if (!f->ev || !f->ev_frame || !f->ev_mask) {
            delete[] f->ev; delete[] f->ev_frame; delete[] f->ev_mask;
            if (f->texels.ptr) (void)hipFree(f->texels.ptr);
            delete f;
            return set_error(KFX_E_RANGE, "kfx_frame_create: out of memory");
        }
        for (int i = 0; i < f->slots; ++i) { f->ev_frame[i] = -1; f->ev_mask[i] = 0; }
        for (int i = 0; i < f->slots * 4; ++i) {
            const hipError_t e = hipEventCreate(&f->ev[i]);   // (hipEventReleaseToDevice events cost the stream the same: measured)
            if (e != hipSuccess) {
                (void)hipGetLastError();
                for (int k = 0; k < i; ++k) (void)hipEventDestroy(f->ev[k]);
                delete[] f->ev; delete[] f->ev_frame; delete[] f->ev_mask;
                if (f->texels.ptr) (void)hipFree(f->texels.ptr);
                delete f;
                return set_error((int)e, "kfx_frame_create: hipEventCreate");
            }
        }
    }
    *out = f;
    return 0;
}

extern "C" int kfx_frame_destroy(kfx_frame* f)
{
    if (!f) return 0;
    if (f->summary) kfx_sdf_summary_destroy(f->summary);   // (synchronises the device)
    if (f->ev) {
        for (int i = 0; i < f->slots * 4; ++i) (void)hipEventDestroy(f->ev[i]);
        delete[] f->ev;
    }
    delete[] f->ev_frame;
    delete[] f->ev_mask;
    if (f->texels.ptr) { (void)hipFree(f->texels.ptr); (void)hipGetLastError(); }
    delete f;
    return 0;
}

// Which of a frame's four events the following steps record.  An event costs the stream a marker between two launches (four per
// frame: 2.7 % of a 0.42 ms frame, measured); a loop that only needs the SdfFuse window and the frame period records
// KFX_FRAME_EVENTS_FUSE.
extern "C" int kfx_frame_set_timing(kfx_frame* f, unsigned mask)
{
    if (!f) return set_error(KFX_E_NULL, "kfx_frame_set_timing: null frame");
    if (mask > KFX_FRAME_EVENTS_ALL) return set_error(KFX_E_RANGE, "kfx_frame_set_timing: mask");
    f->mask = mask;
    return 0;
}

extern "C" int kfx_frame_get_track(const kfx_frame* f) { return f ? f->track : set_error(KFX_E_NULL, "kfx_frame_get_track: null frame"); }
extern "C" kfx_sdf_summary* kfx_frame_summary(kfx_frame* f) { return f ? f->summary : nullptr; }
extern "C" long long kfx_frame_count(const kfx_frame* f) { return f ? f->frames : 0; }

extern "C" int kfx_frame_set_track(kfx_frame* f, int on, kfx_stream stream)
{
    if (!f) return set_error(KFX_E_NULL, "kfx_frame_set_track: null frame");
    if (!on) {
        f->track = 0;   // the summary stays allocated and goes stale: the next set_track(1) rebuilds it
        return 0;
    }
    if (!f->summary)
        if (int e = kfx_sdf_summary_create(&f->summary, &f->cfg.vol)) return e;
    if (!f->track)
        if (int e = kfx_sdf_summary_rebuild(f->summary, stream)) return e;
    f->track = 1;
    return 0;
}

// SdfReset(vol, NaN): "never observed" = (NaN, 0) (main.cpp:229)
extern "C" int kfx_frame_reset(kfx_frame* f, kfx_stream stream)
{
    if (!f) return set_error(KFX_E_NULL, "kfx_frame_reset: null frame");
    const float nan = __builtin_nanf("");
    if (f->track) return kfx_sdf_reset_tracked(&f->cfg.vol, f->summary, nan, stream);
    return kfx_sdf_reset(&f->cfg.vol, nan, stream);
}

extern "C" int kfx_frame_step(kfx_frame* f, const kfx_image* raw, const float T_wc[12], const float* T_cw, unsigned parts, kfx_stream stream)
{
    if (!f || !T_wc) return set_error(KFX_E_NULL, "kfx_frame_step: null argument");
    if (parts == 0) parts = KFX_FRAME_PREPROCESS | KFX_FRAME_FUSE | KFX_FRAME_RAYCAST;
    const kfx_frame_config& c = f->cfg;
    const kfx_image* src = raw ? raw : &c.raw;
    float inv[12];
    if (!T_cw) {   // SE3inv: [R^T | -R^T t], evaluated in double and rounded once
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
    hipEvent_t* ev = nullptr;
    unsigned m = 0;
    if (f->slots && f->mask) {
        const int slot = (int)(f->frames % f->slots);
        ev = f->ev + (size_t)slot * 4;
        m = f->mask;
        f->ev_frame[slot] = f->frames;
        f->ev_mask[slot] = (unsigned char)m;
    } else if (f->slots) {
        f->ev_frame[(int)(f->frames % f->slots)] = -1;
    }
    const hipStream_t s = (hipStream_t)stream;
    int e = 0;
    const auto record = [&](int k) {   // (a failed record leaves an event that timings() would wait on for ever: reported, the frame stops)
        if (!(m & (1u << k))) return;
        const hipError_t he = hipEventRecord(ev[k], s);
        if (he != hipSuccess) {
            (void)hipGetLastError();
            f->ev_mask[(int)(f->frames % f->slots)] &= (unsigned char)~(1u << k);
            if (!e) e = set_error((int)he, "kfx_frame_step: hipEventRecord");
        }
    };
    record(0);
    // the packed texels travel from this step's preprocess to this step's SdfFuse only (nobody else can have touched the maps in between)
    const kfx_image* tex = ((parts & KFX_FRAME_PREPROCESS) && (parts & KFX_FRAME_FUSE) && f->texels.ptr) ? &f->texels : nullptr;
    if (!e && (parts & KFX_FRAME_PREPROCESS)) {
        e = kfx_bilateral_f32(&c.filtered, src, c.bilateral_gs, c.bilateral_gr, c.bilateral_size, c.bilateral_minval, 1, stream);
        if (!e) e = depth_to_vbo_normals_texels(&c.vbo, &c.normals, &c.filtered, c.K, 1.0f, tex, stream);
    }
    record(1);
    if (!e && (parts & KFX_FRAME_FUSE))
        e = sdf_fuse_texels(&c.vol, f->track ? f->summary : nullptr, &c.filtered, &c.normals, tex, T_cw, c.K, c.trunc_dist, c.max_w, c.mincostheta, c.fuse_flags, stream);
    record(2);
    if (!e && (parts & KFX_FRAME_RAYCAST)) {
        if (f->track) e = kfx_raycast_sdf_tracked(&c.ray_depth, &c.ray_norm, &c.ray_img, &c.vol, f->summary, T_wc, c.K, c.near, c.far, c.trunc_dist, 1, stream);
        else e = kfx_raycast_sdf(&c.ray_depth, &c.ray_norm, &c.ray_img, &c.vol, T_wc, c.K, c.near, c.far, c.trunc_dist, 1, stream);
    }
    record(3);
    f->frames += 1;
    return e;
}

extern "C" int kfx_frame_timings(kfx_frame* f, long long first_frame, int n_frames, float* ms)
{
    if (!f || !ms) return set_error(KFX_E_NULL, "kfx_frame_timings: null argument");
    if (!f->slots) return set_error(KFX_E_RANGE, "kfx_frame_timings: the frame was created without timing slots");
    if (n_frames <= 0) return 0;
    const long long last = first_frame + n_frames - 1;
    if (first_frame < 0 || last >= f->frames || f->frames - first_frame > f->slots) return set_error(KFX_E_RANGE, "kfx_frame_timings: frames not in the ring");
    const auto slot_of = [&](long long fr) { return (int)(fr % f->slots); };
    const auto first_event = [](unsigned m) { for (int k = 0; k < 4; ++k) if (m & (1u << k)) return k; return -1; };
    const auto last_event = [](unsigned m) { for (int k = 3; k >= 0; --k) if (m & (1u << k)) return k; return -1; };
    // wait for the latest event any of the answers needs (events of one stream complete in order): the end of the last frame's
    // period -- the next frame's copy of the last frame's first event -- or, failing that, the last event of the latest frame
    // asked for that recorded any
    {
        const auto recorded = [&](long long fr) { return fr >= 0 && fr < f->frames && f->ev_frame[slot_of(fr)] == fr ? (unsigned)f->ev_mask[slot_of(fr)] : 0u; };
        hipEvent_t wait_for = nullptr;
        const int b = first_event(recorded(last));
        if (b >= 0 && (recorded(last + 1) & (1u << b))) wait_for = f->ev[(size_t)slot_of(last + 1) * 4 + b];
        for (long long fr = last; !wait_for && fr >= first_frame; --fr)
            if (recorded(fr)) wait_for = f->ev[(size_t)slot_of(fr) * 4 + last_event(recorded(fr))];
        if (wait_for) {
            const hipError_t he = hipEventSynchronize(wait_for);
            if (he != hipSuccess) { (void)hipGetLastError(); return set_error((int)he, "kfx_frame_timings: hipEventSynchronize"); }
        }
    }
    const float nan = __builtin_nanf("");
    for (int i = 0; i < n_frames; ++i) {
        const long long fr = first_frame + i;
        const int slot = slot_of(fr);
        float* o = ms + (size_t)i * KFX_FRAME_TIMING_FIELDS;
        for (int k = 0; k < KFX_FRAME_TIMING_FIELDS; ++k) o[k] = nan;
        if (f->ev_frame[slot] != fr) {
            if (f->ev_frame[slot] > fr) return set_error(KFX_E_RANGE, "kfx_frame_timings: frame overwritten");
            continue;   // a frame stepped with no events: NaN
        }
        const unsigned m = f->ev_mask[slot];
        hipEvent_t* e = f->ev + (size_t)slot * 4;
        hipError_t he = hipSuccess;
        const auto span = [&](int a, int b, float* out) {
            if (he == hipSuccess && (m & (1u << a)) && (m & (1u << b))) he = hipEventElapsedTime(out, e[a], e[b]);
        };
        span(0, 1, &o[0]);
        span(1, 2, &o[1]);
        span(2, 3, &o[2]);
        span(0, 3, &o[3]);
        // period: this frame's first recorded event to the same event of the next frame
        const int b = first_event(m);
        if (he == hipSuccess && b >= 0 && fr + 1 < f->frames) {
            const int ns = slot_of(fr + 1);
            if (f->ev_frame[ns] == fr + 1 && (f->ev_mask[ns] & (1u << b))) he = hipEventElapsedTime(&o[4], e[b], f->ev[(size_t)ns * 4 + b]);
        }
        if (he != hipSuccess) { (void)hipGetLastError(); return set_error((int)he, "kfx_frame_timings: hipEventElapsedTime"); }
    }
    return 0;
}
