#!/usr/bin/env python3
"""Prints a window of a rocprofv3 kernel trace (start / end / duration / queue / kernel), e.g. one frame of an app:
python scripts/trace_window.py <trace dir> [first fraction 0..1] [count]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:], r.get("Queue_Id", "")) for r in csv.DictReader(open(f)))
win = ev[int(len(ev) * frac):int(len(ev) * frac) + count]
t0 = win[0][0]
for s, e, n, q in win:
    print("%9.1f %9.1f  %7.1f us  q=%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
