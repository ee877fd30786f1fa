#!/usr/bin/env python3
"""march_probe.py -- what a step of the ray-march costs depending on where its cells come from (kfx_debug_march_probe,
include/kfx_debug.h): the measurement behind the LDS-slab march decision (round-3 verdict item 4, EXPERIMENTS.md section 6).
Prints one JSON object: microseconds per step for the plain march's global sample, an LDS-resident box, and workgroup-staged
boxes re-staged every S steps (with and without the next box's loads in flight), for a lone wave per SIMD (one workgroup per
CU: the tail of the real kernel) and for the whole image's worth of workgroups."""
import ctypes as C
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from kangaroo_amd import _lib, roo  # noqa: E402


def main():
    D = _lib.load_debug()
    D.kfx_debug_march_probe.restype = C.c_int
    D.kfx_debug_march_probe.argtypes = [_lib.PV, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    N = 512
    vol = roo.BoundedVolume(N, N, N, (-1, -1, 2), (1, 1, 4))
    roo.SdfReset(vol, 1.0)
    steps = 160
    out = {"volume": N, "steps": steps, "tile": "32 x 8 pixels per workgroup (4 waves of 32 x 2), r = 1.35 voxels per pixel, one voxel per step"}
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for wgs, label in ((256, "one workgroup per CU (a lone wave per SIMD: the tail)"), (1200, "1200 workgroups (a 640 x 480 image)")):
        res = {}
        cyc = torch.zeros(wgs * 4, dtype=torch.int64, device="cuda")
        for mode, S, name in ((0, 4, "global_sample"), (1, 4, "lds_resident"), (2, 2, "lds_restage_S2"), (2, 4, "lds_restage_S4"),
                              (3, 2, "lds_restage_prefetch_S2"), (3, 4, "lds_restage_prefetch_S4")):
            ts = []
            for rep in range(6):
                if mode == 0:   # a cold start for the global march: sweep another buffer through the caches
                    junk = torch.empty(1 << 28, dtype=torch.uint8, device="cuda").zero_()
                    del junk
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                e = D.kfx_debug_march_probe(vol.ref(), mode, steps, S, wgs, C.c_float(1.35), C.c_void_p(cyc.data_ptr()), st)
                b.record()
                torch.cuda.synchronize()
                assert e == 0, e
                ts.append(a.elapsed_time(b))
            ts = sorted(ts[1:])
            res[name] = {"kernel_ms": round(ts[len(ts) // 2], 4), "us_per_step": round(1e3 * ts[len(ts) // 2] / steps, 4)}
        out[label] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
