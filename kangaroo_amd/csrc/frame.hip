// frame.hip -- kfx_frame (include/kfx.h): one frame of the reference application's loop
// (applications/kinectfusion/main.cpp:200-356, known poses) enqueued by ONE call.  Host code only: every launch goes through
// the library's own entry points (kfx_bilateral_f32, kfx_depth_to_vbo_normals_f32, kfx_sdf_fuse[_tracked],
// kfx_raycast_sdf[_tracked]), so a step writes exactly what the separate calls write; the point is the host side -- seven
// interpreter-level calls per 0.45 ms frame become one -- and device-event timing of the frame's parts that does not depend
// on the caller's runtime (torch's events see only torch's current stream).
#include <cmath>
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
    if (f->slots) {
        f->ev = new (std::nothrow) hipEvent_t[(size_t)f->slots * 4];
        f->ev_frame = new (std::nothrow) long long[f->slots];
        if (!f->ev || !f->ev_frame) {
            delete[] f->ev; delete[] f->ev_frame; delete f;
            return set_error(KFX_E_RANGE, "kfx_frame_create: out of memory");
        }
        for (int i = 0; i < f->slots; ++i) f->ev_frame[i] = -1;
        for (int i = 0; i < f->slots * 4; ++i) {
            const hipError_t e = hipEventCreate(&f->ev[i]);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                for (int k = 0; k < i; ++k) (void)hipEventDestroy(f->ev[k]);
                delete[] f->ev; delete[] f->ev_frame; delete f;
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
    delete f;
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
    if (f->slots) {
        const int slot = (int)(f->frames % f->slots);
        ev = f->ev + (size_t)slot * 4;
        f->ev_frame[slot] = f->frames;
    }
    const hipStream_t s = (hipStream_t)stream;
    int e = 0;
    if (ev) (void)hipEventRecord(ev[0], s);
    if (parts & KFX_FRAME_PREPROCESS) {
        e = kfx_bilateral_f32(&c.filtered, src, c.bilateral_gs, c.bilateral_gr, c.bilateral_size, c.bilateral_minval, 1, stream);
        if (!e) e = kfx_depth_to_vbo_normals_f32(&c.vbo, &c.normals, &c.filtered, c.K, 1.0f, stream);
    }
    if (ev) (void)hipEventRecord(ev[1], s);
    if (!e && (parts & KFX_FRAME_FUSE)) {
        if (f->track) e = kfx_sdf_fuse_tracked(&c.vol, f->summary, &c.filtered, &c.normals, T_cw, c.K, c.trunc_dist, c.max_w, c.mincostheta, c.fuse_flags, stream);
        else e = kfx_sdf_fuse(&c.vol, &c.filtered, &c.normals, T_cw, c.K, c.trunc_dist, c.max_w, c.mincostheta, c.fuse_flags, stream);
    }
    if (ev) (void)hipEventRecord(ev[2], s);
    if (!e && (parts & KFX_FRAME_RAYCAST)) {
        if (f->track) e = kfx_raycast_sdf_tracked(&c.ray_depth, &c.ray_norm, &c.ray_img, &c.vol, f->summary, T_wc, c.K, c.near, c.far, c.trunc_dist, 1, stream);
        else e = kfx_raycast_sdf(&c.ray_depth, &c.ray_norm, &c.ray_img, &c.vol, T_wc, c.K, c.near, c.far, c.trunc_dist, 1, stream);
    }
    if (ev) (void)hipEventRecord(ev[3], s);
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
    // the period of the last frame asked for ends at the next frame's first event, if there is one
    const bool next_known = last + 1 < f->frames;
    hipEvent_t* ev_last = f->ev + (size_t)(last % f->slots) * 4;
    hipError_t he = hipEventSynchronize(next_known ? f->ev[(size_t)((last + 1) % f->slots) * 4] : ev_last[3]);
    if (he != hipSuccess) { (void)hipGetLastError(); return set_error((int)he, "kfx_frame_timings: hipEventSynchronize"); }
    for (int i = 0; i < n_frames; ++i) {
        const long long fr = first_frame + i;
        const int slot = (int)(fr % f->slots);
        if (f->ev_frame[slot] != fr) return set_error(KFX_E_RANGE, "kfx_frame_timings: frame overwritten");
        hipEvent_t* e = f->ev + (size_t)slot * 4;
        float* o = ms + (size_t)i * KFX_FRAME_TIMING_FIELDS;
        he = hipEventElapsedTime(&o[0], e[0], e[1]);
        if (he == hipSuccess) he = hipEventElapsedTime(&o[1], e[1], e[2]);
        if (he == hipSuccess) he = hipEventElapsedTime(&o[2], e[2], e[3]);
        if (he == hipSuccess) he = hipEventElapsedTime(&o[3], e[0], e[3]);
        o[4] = __builtin_nanf("");
        if (he == hipSuccess && fr + 1 < f->frames) {
            const int nslot = (int)((fr + 1) % f->slots);
            if (f->ev_frame[nslot] == fr + 1) he = hipEventElapsedTime(&o[4], e[0], f->ev[(size_t)nslot * 4]);
        }
        if (he != hipSuccess) { (void)hipGetLastError(); return set_error((int)he, "kfx_frame_timings: hipEventElapsedTime"); }
    }
    return 0;
}
