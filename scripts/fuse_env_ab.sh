#!/bin/bash
# A/B of one environment switch of the SdfFuse launch on the GPU box, interleaved, same library, bench.py's own frame loop (SdfFuse
# between device events).  Usage: scripts/fuse_env_ab.sh <tag> <VAR> [rounds]   (VAR=0 against VAR=1)
TAG=${1:-r06_env_ab}; VAR=${2:-KFX_FUSE_POS_DIV}; ROUNDS=${3:-2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
CONFIGS=("c2_full|" "c2_room|--scene room" "c3_room|--config c3" "c2_room_exact|--scene room --math exact")
for r in $(seq 1 $ROUNDS); do
  for cfg in "${CONFIGS[@]}"; do
    name=${cfg%%|*}; args=${cfg#*|}
    for v in 0 1; do
      export $VAR=$v
      python3 bench.py --steps 120 --warmup 10 --prime-seconds 1 --no-extra-legs --no-cpu-baseline $args > $OUT/${name}_v${v}_$r.json 2> $OUT/${name}_v${v}_$r.err
    done
  done
done
unset $VAR
python3 - $OUT $VAR <<'PY'
import glob, json, os, sys, collections
out, var = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "*_v[01]_*.json"))):
    name, v, r = os.path.basename(f)[:-5].rsplit("_", 2)
    try:
        d = json.load(open(f))
    except ValueError:
        print("no line:", f); continue
    rows[name][v].append((d["roofline"]["avg_launch_ms"], d["value"]))
res = {"switch": var}
for name, vs in rows.items():
    res[name] = {v: {"sdf_fuse_ms": [x[0] for x in vals], "frames_per_sec": [x[1] for x in vals]} for v, vals in vs.items()}
    print(name.ljust(16) + "".join("  %s=%s: %s ms (%s fps)" % (var, v[1:], "/".join("%.4f" % x[0] for x in vs[v]), "/".join("%.0f" % x[1] for x in vs[v])) for v in ("v0", "v1") if v in vs))
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
PY
