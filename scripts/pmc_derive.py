#!/usr/bin/env python3
"""Derived figures from a scripts/pmc_any.sh summary, with the counters' units made explicit (MI355X_MICROARCH.md):
SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count QUAD-cycles (x 4 = cycles), summed over the device; GRBM_GUI_ACTIVE counts
cycles, summed over the 8 XCDs (/ 8 x 256 CUs = CU-cycles of the launch); SQ_LDS_IDX_ACTIVE, TA_TA_BUSY count cycles.
Usage: python scripts/pmc_derive.py summary.txt [out.json]"""
import json
import re
import sys

CUS, XCDS, SIMDS = 256, 8, 4
kern, cur = {}, None
for line in open(sys.argv[1]):
    m = re.match(r"== (.+)", line)
    if m:
        cur = kern.setdefault(m.group(1).strip(), {})
        continue
    m = re.match(r"\s+(\w+)\s+avg=\s*([0-9.]+)", line)
    if m and cur is not None:
        cur[m.group(1)] = float(m.group(2))
    m = re.match(r"time (.+?)\s+calls (\d+) avg ([0-9.]+) us", line)
    if m:
        for k in kern:
            if k[:40] in m.group(1).replace("void ", "").replace("kfx::", ""):
                kern[k]["_avg_us"] = float(m.group(3))
out = {}
for k, c in kern.items():
    if "GRBM_GUI_ACTIVE" not in c or "SQ_WAVE_CYCLES" not in c:
        continue
    cu_cycles = c["GRBM_GUI_ACTIVE"] / XCDS * CUS
    simd_cycles = cu_cycles * SIMDS
    q = lambda name: 4.0 * c.get(name, 0.0)   # noqa: E731  quad-cycles -> cycles
    d = {"avg_launch_us": c.get("_avg_us"), "waves": c.get("SQ_WAVES"), "cu_cycles": round(cu_cycles),
         "waves_resident_per_cu": round(q("SQ_WAVE_CYCLES") / cu_cycles, 2),
         "valu_busy_of_simd_cycles": round(q("SQ_ACTIVE_INST_VALU") / simd_cycles, 3),
         "wave_time_parked": round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 3),
         "wave_time_issue_stalled": round(c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 3),
         "wave_time_issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 3),
         "lds_data_path_busy_of_cu_cycles": round(c.get("SQ_LDS_IDX_ACTIVE", 0) / cu_cycles, 3),
         "ta_busy_of_cu_cycles": round(c.get("TA_TA_BUSY_sum", 0) / cu_cycles, 3),
         "valu_wave_instructions": c.get("SQ_INSTS_VALU"), "valu_instructions_per_wave": round(c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_WAVES", 1), 1), 1),
         "l2_hit_rate": round(c.get("TCC_HIT_sum", 0) / max(c.get("TCC_REQ_sum", 1), 1), 3)}
    out[k] = d
    print(k, json.dumps(d))
if len(sys.argv) > 2:
    json.dump({"units": __doc__.split("\n")[1:5], "kernels": out}, open(sys.argv[2], "w"), indent=1)
