/* kfx_debug.h -- measurement aids and arithmetic self-checks exported by libkfx_debug.so (which links against libkfx.so);
 * not part of the drop-in boundary (include/kfx.h) and not in the product library. */
#ifndef KFX_DEBUG_H
#define KFX_DEBUG_H

#include "kfx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* In-place read-modify-write sweep of a BoundedVolume<SDF_t> (val += 1 on every cell, 16 B per lane), the memory
 * ceiling SdfFuse is measured against (scripts/rmw_floor.py).  variant selects the thread
 * mapping: 0 = linear grid-stride sweep of the contiguous span, 1 = the tiled fuse kernel's brick (64 x 8 x 16 voxels
 * per workgroup, z-march), 2 / 5 = 128-voxel rows x 4 with 16 / 4 slices, 3 = whole z-columns, 4 = 64-slice bricks,
 * 10-17 = generated shapes with and without nontemporal accesses (kangaroo_amd/csrc/debug.hip). */
int kfx_debug_rmw(const kfx_volume* vol, int variant, kfx_stream stream);

/* Exhaustive check of the kernels' division-by-a-uniform-divisor shortcut (kfx_device.h: div_uniform) against the
 * hardware IEEE division: all 2^32 numerator bit patterns for divisor b.  d_out[0] += mismatching patterns,
 * d_out[1] += patterns tested (those inside the shortcut's operand range); both device-side 64-bit counters. */
int kfx_debug_div_uniform_check(float b, unsigned long long* d_out, kfx_stream stream);

/* The exact-mode SdfFuse kernel's division / square-root shortcuts (kfx_device.h: rcp_nr + div_core, sqrt_core) against
 * the hardware IEEE operations.  div: every divisor significand in ten binades of [2^-40, 2^40] x per_divisor numerators
 * (random and awkward significands in [2^-60, 2^60], zeros); sqrt: every float in [2^-80, 2^80].  d_out[0] += results that
 * differ (a zero of the other sign counts as equal), d_out[1] += cases tested. */
int kfx_debug_div_core_check(unsigned seed, int per_divisor, unsigned long long* d_out, kfx_stream stream);
int kfx_debug_sqrt_core_check(unsigned long long* d_out, kfx_stream stream);
/* wave_xor_combine (DPP / permlane swaps, kfx_device.h) against __shfl_xor for lane distances 1, 2, 16, 32, min and max:
 * d_out[0] += mismatches, d_out[1] += comparisons */
int kfx_debug_wave_xor_check(unsigned seed, unsigned long long* d_out, kfx_stream stream);

/* Copies of a kfx_sdf_summary's tables for tests: R_out receives {lo, hi, state bits, 0} per 8 x 8 x 8 brick (state 0: every
 * cell has a value in [lo, hi], 1: every cell NaN, 2: every cell is NaN or has a value in [lo, hi]; an invalidated brick has
 * the infinite range), C_out the ray-march's class tables built for relative tolerance `tol`, reference value `vref` and a fine
 * level of 2^fine_shift cells (3, 4; 5: the 32^3-cell level only): per entry two bits in two planes, rows of entries along x
 * as uint2 {plane 0, plane 1} per 32 entries -- 0 sample, 1 every cell (and the +1 cells a sample based in the entry reads)
 * holds vref, 2 every cell NaN, 3 every cell NaN or vref.  dims_out = {bricks along x, y, z; fine level: first word, words per
 * row, rows per plane of entries; the same for the 32^3 level; total words; number of 32^3-cell entries; the count of them with
 * class != 0 as published by the last finished build (-1: none yet)}.  Device buffers: R_out n float4, C_out dims_out[9] words
 * (call with C_out = NULL first to learn the size). */
int kfx_debug_summary_export(kfx_sdf_summary* s, float tol, float vref, int fine_shift, void* R_out, void* C_out, int dims_out[12],
                             kfx_stream stream);


#ifdef __cplusplus
}
#endif
#endif
