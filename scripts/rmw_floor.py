#!/usr/bin/env python3
"""HBM ceiling of in-place RMW sweeps over a 512^3 SDF volume (kfx_debug_rmw variants)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kangaroo_amd import roo, _lib
L = _lib.load()
_lib.load_debug().kfx_debug_rmw.argtypes = [_lib.PV, C.c_int, C.c_void_p]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
vol = roo.BoundedVolume(N, N, N)
roo.SdfReset(vol, 0.0)
names = {0: "linear grid-stride", 1: "brick 64x8x16", 2: "rows 128x4x16", 3: "columns 128x4xZ", 4: "brick 64x8x64", 5: "rows 128x4x4", 10: "gen 512x1x16", 11: "gen 256x2x16", 12: "gen 128x4x16", 13: "gen 128x4x16 NT", 14: "gen 512x1x16 NT", 15: "gen 128x4x1", 16: "gen 128x4x64", 17: "gen 512x1x4", 18: "gen 256x2x16 NT"}
gb = 16.0 * N ** 3 / 1e9
# device-to-device copy ceiling for reference
a = torch.empty(N ** 3 * 8, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
for v in list(names) + ["copy"]:
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
    for s, e in ev:
        s.record()
        if v == "copy":
            b.copy_(a)
        else:
            assert _lib.load_debug().kfx_debug_rmw(vol.ref(), v, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in ev)[2:]
    med = ms[len(ms) // 2]
    print("%-22s med %.4f ms  min %.4f ms  -> %.0f GB/s (med) %.0f GB/s (min)" % (names.get(v, "torch copy_ (D2D)"), med, ms[0], gb / med * 1e3, gb / ms[0] * 1e3))
