#!/bin/bash
# Runs on the GPU box (via gpurun): the exact slab raycast's kernels with 8 rank threads on the one GPU, under rocprofv3's kernel
# trace -- the whole-image stage protocol (kfx_slab_raycast_exact: world + 1 stages of whole images) against the hand-over
# pipelined over image row-tiles (kfx_slab_raycast_exact_tiled, 1 / 4 / 8 tiles).  Per variant: kernel time summed over the
# eight ranks and per frame, launches per rank and frame, bytes a rank sends per stage / step.
# Usage: scripts/handover_profile.sh <tag>
set -u
TAG=${1:-r05_handover}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
FRAMES=12
for V in stages tiles1 tiles4 tiles8; do
  case $V in
    stages) ARGS="--tiles 0";;
    tiles1) ARGS="--tiles 1";;
    tiles4) ARGS="--tiles 4";;
    tiles8) ARGS="--tiles 8";;
  esac
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$V -- $ROOT/apps/kinectfusion_slabs --res 512 --frames $FRAMES --ranks 8 --raycast exact --halo recompute --fast $ARGS > $OUT/$V.log 2> $OUT/$V.err
  tail -2 $OUT/$V.log
done
python3 - "$OUT" $FRAMES <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out, frames = sys.argv[1], int(sys.argv[2])
W, w, h = 8, 640, 480
res = {}
for v in ("stages", "tiles1", "tiles4", "tiles8"):
    agg = defaultdict(list)
    for f in glob.glob(os.path.join(out, v, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kfx::", "")
            agg[name[:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    march = {k: v_ for k, v_ in agg.items() if any(s in k for s in ("k_raycast_sdf_slab", "k_handover", "k_adopt_newer", "k_tiles_", "k_strip_sum", "k_group_reduce", "k_raycast_state_to_images"))}
    tiles = {"stages": 0, "tiles1": 1, "tiles4": 4, "tiles8": 8}[v]
    n_px = w * h
    if tiles == 0:
        steps, per_step = W + 1, 2 * 5 * n_px * 4          # every stage: the five march planes to BOTH neighbours
    else:
        rows = -(-h // tiles)
        steps, per_step = W + tiles - 1, 2 * 5 * (((rows * w + 63) // 64) * 64) * 4   # a step: one tile up, one tile down (where the tokens are)
    res[v] = {"kernels_us_per_frame_all_ranks": {k: round(sum(x) / frames, 2) for k, x in sorted(march.items())},
              "launches_per_frame_and_rank": {k: round(len(x) / frames / W, 2) for k, x in sorted(march.items())},
              "march_kernels_total_us_per_frame_all_ranks": round(sum(sum(x) for x in march.values()) / frames, 2),
              "march_launches_per_frame_and_rank": round(sum(len(x) for x in march.values()) / frames / W, 2),
              "steps_with_an_exchange": steps, "bytes_sent_per_rank_and_step_at_most": per_step,
              "final_stage_bytes_per_rank": 0 if tiles == 0 else 2 * 5 * tiles * (((-(-h // tiles) * w + 63) // 64) * 64) * 4,
              "final_all_reduce_bytes": 6 * n_px * 4}
    print(v, json.dumps(res[v]))
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
PY
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*.csv" -size +4M -delete
