// slab_internal.h -- what slab.hip offers slab_frame.hip beyond include/kfx_slab.h: the exact hand-over in its two halves, so that a
// frame object can leave the final exchange of frame k on its own stream and communicator while the march of frame k + 1 runs.
#pragma once
#include "../../include/kfx_slab.h"

namespace kfx {
// the buffers of one final exchange: this rank's contributions, the received / gathered strips, this rank's merged strip, the count of
// pixels left without a final status
struct ExactFinalBufs { int* contrib; int* gathered; int* mine; int* open; };
size_t exact_final_bytes(size_t w, size_t h, int world);
void exact_final_carve(ExactFinalBufs& b, void* mem, size_t w, size_t h, int world);
// kfx_slab_raycast_exact_tiled = exact_tiled_march + exact_tiled_finalise on one stream with the scratch's own buffers (into / from null)
int exact_tiled_march(void* scratch, const ExactFinalBufs* into, const kfx_volume* local, const kfx_slab_layout* L, const float T_wc[12], const float K[4],
                      float near, float far, float trunc_dist, int subpix, int tiles, int w, int h, kfx_comm* comm, kfx_stream stream, int* steps_out);
int exact_tiled_finalise(const kfx_image* depth, const kfx_image* norm, const kfx_image* img, void* scratch, const ExactFinalBufs* from, int tiles,
                         kfx_comm* comm, kfx_stream stream, int* h_open);
}
