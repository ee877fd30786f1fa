#!/usr/bin/env python3
"""Condenses a gpu_profile.sh output directory into a small text summary:
per-kernel count / avg / total from the kernel-trace, and FETCH_SIZE / WRITE_SIZE per launch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    return name.split("(")[0].replace("void ", "").replace("kfx::", "")[:72]


def kernel_stats(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    agg = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            try:
                agg[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            except (KeyError, ValueError):
                pass
    return agg


def pmc(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


print("== kernel trace (us) ==")
ks = kernel_stats(os.path.join(out, "trace"))
tot = sum(sum(v) for v in ks.values()) or 1.0
for k, v in sorted(ks.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("%-62s n=%5d avg=%10.2f med=%10.2f min=%10.2f max=%10.2f total=%12.1f (%5.1f%%)" % (
        k, len(v), sum(v) / len(v), v2[len(v2) // 2], v2[0], v2[-1], sum(v), 100 * sum(v) / tot))
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    print("== %s per launch (counter units as reported by rocprofv3; KiB on gfx9-family) ==" % name)
    for k, v in sorted(pmc(os.path.join(out, sub), name).items(), key=lambda kv: -sum(kv[1])):
        print("%-62s n=%5d avg=%14.1f min=%14.1f max=%14.1f" % (k, len(v), sum(v) / len(v), min(v), max(v)))
