#!/usr/bin/env python3
"""The ICP-tracked Python loop (bench.py's tracked_variant) by itself, for rocprofv3 --kernel-trace: prints its frame time; with
KFX_TRACE_CSV=<kernel_trace.csv> instead analyses a trace of it -- per frame: kernel time, idle time, the gap between the pose's
read-back and the first SdfFuse launch.  Usage: rocprofv3 --kernel-trace --output-format csv -d out -- python3 scripts/tracked_python_gaps.py"""
import csv
import os
import statistics
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if os.environ.get("KFX_TRACE_CSV"):
    rows = sorted(csv.DictReader(open(os.environ["KFX_TRACE_CSV"])), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kfx::", "") for r in rows]
    starts = [i for i, n in enumerate(names) if n.startswith("k_raycast_sdf_levels")]
    per = []
    for a, b in zip(starts[30:-1], starts[31:]):
        seq = rows[a:b]
        total = (int(rows[b]["Start_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e3
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seq) / 1e3
        gap = None
        for i in range(1, len(seq)):
            if names[a + i].startswith("k_sdf_fuse") and not names[a + i - 1].startswith("k_sdf_fuse"):
                gap = (int(seq[i]["Start_Timestamp"]) - int(seq[i - 1]["End_Timestamp"])) / 1e3
        per.append((total, busy, gap, names[a + 1:b]))
    print("frames %d  median frame %.1f us  kernels %.1f us  idle %.1f us  gap before SdfFuse %.1f us" % (
        len(per), statistics.median(p[0] for p in per), statistics.median(p[1] for p in per),
        statistics.median(p[0] - p[1] for p in per), statistics.median(p[2] for p in per if p[2] is not None)))
    print("a frame's launches:", per[len(per) // 2][3])
    sys.exit(0)

import torch  # noqa: E402
from kangaroo_amd import roo, scenes  # noqa: E402
import bench  # noqa: E402
roo.set_math_mode("fast")
out = bench.tracked_leg(SimpleNamespace(res=512, width=640, height=480), torch, roo, scenes, 240)
print({k: out[k] for k in ("frames_per_sec", "ms_per_step", "worst_position_error_mm")})
