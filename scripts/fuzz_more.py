#!/usr/bin/env python3
"""Runs the seeded GPU fuzz cases of tests/test_gpu_fuzz.py over a wider seed range (ad hoc soak; not part of pytest).
Usage: python scripts/fuzz_more.py [first_seed] [count]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kangaroo_amd import roo  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = []
for seed in range(first, first + count):
    for fn in (F.test_gpu_fuzz_fuse_count_raycast, F.test_gpu_fuzz_icp_colour_mesh, F.test_gpu_fuzz_half_cells_and_slabs,
               F.test_gpu_fuzz_fast_mode_tolerance):
        try:
            fn(roo, seed)
        except Exception:   # noqa: BLE001
            bad.append((fn.__name__, seed))
            print("FAIL", fn.__name__, seed)
            traceback.print_exc(limit=2)
print("fuzz_more: %d seeds x 4 cases, %d failures %s" % (count, len(bad), bad[:20]))
sys.exit(1 if bad else 0)
