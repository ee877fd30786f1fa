#!/bin/bash
# Runs on the GPU box (via gpurun): the ICP-tracked frame loop of the C++ application (apps/kinectfusion_headless --device-icp
# --fused-launches --fast, 512^3, 640x480) under rocprofv3's kernel trace -- per kernel: launches per frame, average duration and
# microseconds per frame -- beside the application's own frame time without the profiler (what bench.py's tracked_variant times
# through the Python loop).  Usage: [APP_ARGS='--res 512 --frames 150 --warmup 30 --fast --track'] scripts/tracked_profile.sh <tag>
# (APP_ARGS must keep --frames 150: the per-frame figures divide by it)
set -u
TAG=${1:-r05_tracked}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
make -C $ROOT/apps -s > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
FRAMES=150
ARGS=${APP_ARGS:-"--res 512 --frames $FRAMES --warmup 30 --fast --device-icp --fused-launches"}
export KFX_PROFILE_ARGS="$ARGS"
$ROOT/apps/kinectfusion_headless $ARGS > $OUT/app.log 2>&1
$ROOT/apps/kinectfusion_headless $ARGS >> $OUT/app.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- $ROOT/apps/kinectfusion_headless $ARGS > $OUT/trace_app.log 2> $OUT/trace.err
python3 - "$OUT" $FRAMES <<'PY'
import csv, glob, json, os, re, sys
from collections import defaultdict
out, frames = sys.argv[1], int(sys.argv[2])
agg = defaultdict(list)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kfx::", "")
        agg[name[:72]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = [{"kernel": k, "launches_per_frame": round(len(v) / frames, 2), "avg_us": round(sum(v) / len(v), 2), "us_per_frame": round(sum(v) / frames, 2)}
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))]
ms = [float(m) for m in re.findall(r"([0-9.]+) ms/frame", open(os.path.join(out, "app.log")).read())]
res = {"command": "apps/kinectfusion_headless " + os.environ.get("KFX_PROFILE_ARGS", ""),
       "frame_ms_without_profiler": ms, "kernels_us_per_frame_total": round(sum(r["us_per_frame"] for r in rows), 2),
       "launches_per_frame_total": round(sum(r["launches_per_frame"] for r in rows), 2), "kernels": rows}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "kernels"}))
for r in rows[:24]:
    print("%-74s %6.2f/frame avg %8.2f us  %8.2f us/frame" % (r["kernel"], r["launches_per_frame"], r["avg_us"], r["us_per_frame"]))
PY
cat $OUT/app.log | tail -4
find $OUT -name "*.csv" -size +2M -delete
