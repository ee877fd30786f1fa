#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command of this repo; prints the per-kernel table (name, calls, average / min / max us).
# Usage: scripts/kernel_stats.sh <tag> <script.py> [args...]        (run on the GPU box through gpurun)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/"$@" > $OUT/stdout.txt 2> $OUT/trace.err
python3 - <<PY > $OUT/kernel_stats.txt
import csv, glob
rows = []
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((float(r["TotalDurationNs"]), r))
for _, r in sorted(rows, key=lambda t: -t[0])[:25]:
    print("%-100s calls %6s avg %9.2f us min %9.2f max %9.2f" % (r["Name"].replace("kfx::", "")[:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
cat $OUT/kernel_stats.txt
rm -rf $OUT/trace
