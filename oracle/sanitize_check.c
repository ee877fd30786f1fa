/* sanitize_check.c -- drives every oracle entry point on small inputs; built by `make -C oracle asan` with
 * -fsanitize=address,undefined so that out-of-bounds reads / undefined behaviour in the parity oracle itself show
 * up on the CPU (GPU sanitizers are not available on the target pool).  Exit code 0 = clean run. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kfx_oracle.h"

static kfo_image mk_image(size_t w, size_t h, size_t elem, size_t pad)
{
    kfo_image im;
    im.w = w; im.h = h; im.pitch = w * elem + pad;
    im.ptr = calloc(im.pitch * h, 1);
    return im;
}
static kfo_volume mk_volume(size_t w, size_t h, size_t d, size_t elem, const float lo[3], const float hi[3])
{
    kfo_volume v;
    v.w = w; v.h = h; v.d = d; v.pitch = w * elem; v.img_pitch = v.pitch * h;
    v.ptr = calloc(v.img_pitch * d, 1);
    for (int i = 0; i < 3; ++i) { v.boxmin[i] = lo[i]; v.boxmax[i] = hi[i]; }
    return v;
}

int main(void)
{
    const size_t w = 52, h = 37, N = 19;
    const float K[4] = {46.f, 46.f, 25.5f, 18.f};
    const float T[12] = {1, 0, 0, 0.02f, 0, 1, 0, -0.01f, 0, 0, 1, 0.f};
    float Tinv[12];
    kfo_se3_inverse(Tinv, T);
    const float lo[3] = {-1, -1, 2}, hi[3] = {1, 1, 4};
    kfo_image raw = mk_image(w, h, 4, 12), filt = mk_image(w, h, 4, 0), vbo = mk_image(w, h, 16, 16), nrm = mk_image(w, h, 16, 0);
    kfo_render_scene(&raw, 0, T, K);
    ((float*)raw.ptr)[5] = NAN;
    kfo_bilateral_f32(&filt, &raw, 1.5f, 0.1f, 3, 0.2f, 1, 2);
    kfo_depth_to_vbo_f32(&vbo, &filt, K, 1.0f);
    kfo_normals_from_vbo(&nrm, &vbo);
    kfo_volume vol = mk_volume(N, N + 2, N + 5, 8, lo, hi), cvol = mk_volume(N, N + 2, N + 5, 4, lo, hi);
    kfo_sdf_reset(&vol, NAN);
    kfo_color_reset(&cvol);
    const float tr = 0.35f;
    unsigned long long n = kfo_sdf_fuse(&vol, &filt, &nrm, Tinv, K, tr, 100.f, 0.1f, 1, 2);
    kfo_image rgb = mk_image(w, h, 3, 1);
    memset(rgb.ptr, 77, rgb.pitch * h);
    n += kfo_sdf_fuse_color(&vol, &cvol, &filt, &nrm, Tinv, K, &rgb, Tinv, K, tr, 100.f, 0.1f, 1, 2);
    kfo_image rd = mk_image(w, h, 4, 0), rn = mk_image(w, h, 16, 0), ri = mk_image(w, h, 4, 0);
    kfo_raycast_stats st;
    kfo_raycast_sdf(&rd, &rn, &ri, &vol, T, K, 0.4f, 8.f, tr, 1, 2, &st);
    kfo_raycast_sdf_color(&rd, &rn, &ri, &vol, &cvol, T, K, 0.4f, 8.f, tr, 1, 2);
    /* slab march, one slab = whole volume */
    float* state = calloc(9 * w * h, 4);
    kfo_slab slab = {vol.d, 0, lo[2], hi[2]};
    kfo_raycast_sdf_slab(state, 1, &vol, &slab, 0, (int)vol.d, (int)w, (int)h, T, K, 0.4f, 8.f, tr, 1);
    /* ICP on the raycast maps */
    kfo_image rv = mk_image(w, h, 16, 0), dbg = mk_image(w, h, 16, 0);
    kfo_depth_to_vbo_f32(&rv, &rd, K, 1.0f);
    const float KT[12] = {K[0], 0, K[2], 0, 0, K[1], K[3], 0, 0, 0, 1, 0}, I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    kfo_lss6 lss;
    unsigned g[4];
    kfo_icp_block_dims(w, h, g);
    kfo_lss6* blocks = calloc((size_t)g[2] * g[3], sizeof(kfo_lss6));
    kfo_icp_point_plane(&vbo, &rv, &rn, KT, I, 0.1f, &dbg, &lss, blocks);
    /* pyramid / pre-amble helpers */
    kfo_image half_ = mk_image(w / 2, h / 2, 4, 0), sb = mk_image(w, h, 4, 0);
    kfo_box_half_ignore_invalid_f32(&half_, &filt);
    kfo_elementwise_scale_bias_f32(&sb, &raw, 0.001f, 0.f);
    /* analytic renderers, SdfDistance, depth tools */
    kfo_image an = mk_image(w, h, 4, 0), ai = mk_image(w, h, 4, 0), dist = mk_image(w, h, 4, 0), id4 = mk_image(w, h, 4, 0);
    const float c[3] = {0, 0, 3}, nw[3] = {0, 0, -0.26f};
    kfo_raycast_box(&an, T, K, lo, hi);
    kfo_raycast_sphere(&an, &ai, T, K, c, 0.5f);
    kfo_raycast_plane(&an, &ai, T, K, nw);
    kfo_sdf_distance(&dist, &an, &vol, T, K);
    kfo_disp2depth(&raw, &sb, 570.f, 0.075f, 1.f);
    kfo_filter_bad_kinect(&sb, &raw, 0);
    kfo_colour_vbo(&id4, &vbo, &rgb, KT);
    kfo_bilateral_guided(&sb, &filt, &raw, 0, 1.5f, 0.1f, 0.1f, 2);
    /* ROI helpers */
    float bmin[3], bmax[3];
    kfo_fit_to_frustum(bmin, bmax, T, (float)w, (float)h, K, 0.4f, 4.f);
    kfo_volume sub;
    kfo_sub_bounding_volume(&sub, &vol, bmin, bmax);
    printf("sanitize_check: %llu voxel updates, %llu ray hits, %u icp observations, sub volume %zux%zux%zu\n", n,
           (unsigned long long)st.hits, lss.obs, sub.w, sub.h, sub.d);
    return (n > 0 && st.hits > 0 && lss.obs > 0) ? 0 : 1;
}
