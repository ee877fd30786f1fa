#!/bin/bash
# A/B of the tiled SdfFuse kernels' tile staging on the GPU box: the in-tree library (LDS-DMA of the packed texel image) against
# build_ab/regstage/libkfx.so (scripts/build_ab.sh regstage -DKFX_FUSE_STAGE_DMA=0: round 5's staging through registers), interleaved,
# same box, bench.py's own frame loop (SdfFuse between device events).  Usage: scripts/fuse_stage_ab.sh <tag> [rounds] [build_ab name]
# (the column "regstage" is the build_ab library, whichever it is)
TAG=${1:-r06_stage_ab}; ROUNDS=${2:-2}; AB=${3:-regstage}   # AB: the build_ab/<name> library the in-tree one is compared with
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
CONFIGS=("c2_full|" "c2_room|--scene room" "c3_room|--config c3" "c3_full|--config c3 --scene full" "c2_room_exact|--scene room --math exact" "c2_full_exact|--math exact")
for r in $(seq 1 $ROUNDS); do
  for cfg in "${CONFIGS[@]}"; do
    name=${cfg%%|*}; args=${cfg#*|}
    for lib in dma regstage; do
      if [ $lib = regstage ]; then export KFX_LIB_PATH=$ROOT/build_ab/$AB/libkfx.so; else unset KFX_LIB_PATH; fi
      python3 bench.py --steps 120 --warmup 10 --prime-seconds 1 --no-extra-legs --no-cpu-baseline $args > $OUT/${name}_${lib}_$r.json 2> $OUT/${name}_${lib}_$r.err
    done
  done
done
unset KFX_LIB_PATH
python3 - $OUT <<'PY'
import glob, json, os, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    b = os.path.basename(f)[:-5]
    if b == "summary":
        continue
    name, lib, r = b.rsplit("_", 2)
    try:
        d = json.load(open(f))
    except ValueError:
        print("no line:", f); continue
    rows[name][lib].append((d["roofline"]["avg_launch_ms"], d["value"], d["roofline"]["frac"]))
res = {}
for name, libs in rows.items():
    res[name] = {lib: {"sdf_fuse_ms": [v[0] for v in vals], "frames_per_sec": [v[1] for v in vals], "frac": [v[2] for v in vals]} for lib, vals in libs.items()}
    line = name.ljust(16)
    for lib in ("regstage", "dma"):
        if lib in libs:
            line += "  %s: %s ms (%s fps)" % (lib, "/".join("%.4f" % v[0] for v in libs[lib]), "/".join("%.0f" % v[1] for v in libs[lib]))
    print(line)
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
PY
