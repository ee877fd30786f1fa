#!/bin/bash
# fresh_runs.sh TAG [N] [IDLE_S] [extra bench.py args...] -- the driver's command, N times, each as a fresh process after
# IDLE_S seconds of GPU idle (round-3 verdict item 1: the headline has to hold on a cold box).  Keeps every JSON line and the
# per-step dumps (KFX_BENCH_DUMP=1) under gpurun_out/TAG/ and prints a one-line summary per run.
TAG=${1:-fresh}; N=${2:-10}; IDLE=${3:-30}
if [ $# -ge 3 ]; then shift 3; else shift $#; fi   # (what follows the three positional arguments goes to bench.py)
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
for i in $(seq 1 "$N"); do
    sleep "$IDLE"
    KFX_BENCH_DUMP=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 "$@" > "$OUT/run_$i.json" 2> "$OUT/run_$i.err"
    python3 - "$OUT/run_$i.json" "$i" <<'EOF'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r, pv = d["roofline"], d.get("plain_variant") or {}
    print("run %s: %.1f frames/s  fuse %.4f ms frac %.4f  ray %.4f  total %.4f  plain_variant %s  prime %s  decision %s" % (
        sys.argv[2], d["value"], r["avg_launch_ms"], r["frac"], d["kernels_ms"].get("parts", {}).get("raycast_sdf", d["kernels_ms"].get("raycast_sdf", -1)), d["kernels_ms"]["frame_total"],
        pv.get("frames_per_sec"), d.get("prime"), (d["config"].get("summary_policy") or {}).get("decision")))
except Exception as e:
    print("run %s: failed (%r)" % (sys.argv[2], e))
EOF
done
