#!/bin/bash
# PMC passes aimed at RaycastSdf's memory pipeline (TA / TCP / UTCL1 / TCC), counters only.
# Usage: scripts/pmc_raycast.sh <tag> <scene> [env assignments...]
TAG=${1:-pmc_rc}; SCENE=${2:-room}; shift 2 || true
for kv in "$@"; do export "$kv"; done
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
PMCG=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"
        "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum"
        "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
        "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
        "TCC_HIT_sum TCC_MISS_sum"
        "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
        "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum")
# PMC_GROUPS_FILE: one counter group per line instead of the default set
if [ -n "$PMC_GROUPS_FILE" ]; then mapfile -t PMCG < "$ROOT/$PMC_GROUPS_FILE"; fi
for grp in "${PMCG[@]}"; do
  i=$((i+1))
  timeout -k 5 45 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $ROOT/scripts/fuse_only.py $SCENE 3 > /dev/null 2> $OUT/g$i.err || echo "group $i failed: $grp"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ","").replace("kfx::","")[:40]
        if "k_raycast" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in agg:
    print("==", k)
    for c, v in sorted(agg[k].items()):
        print("  %-48s avg=%16.1f n=%d" % (c, sum(v)/len(v), len(v)))
PY
rm -rf $OUT/g*/
