#!/bin/bash
# Counter passes (rocprofv3 --pmc, counters + kernel trace only) over any python script of this repo, one pass per group.
# Usage: scripts/pmc_any.sh <tag> <kernel-name substring> <script.py> [script args...]   (environment is inherited)
TAG=$1; KSUB=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMCG=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"
      "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
      "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
      "TCC_HIT_sum TCC_MISS_sum")
if [ -n "$PMC_GROUPS_FILE" ]; then mapfile -t PMCG < "$ROOT/$PMC_GROUPS_FILE"; fi
i=0
for grp in "${PMCG[@]}"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $ROOT/"$@" > /dev/null 2> $OUT/g$i.err || echo "group $i failed: $grp"
done
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/"$@" > /dev/null 2> $OUT/trace.err
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ","").replace("kfx::","")[:60]
        if "$KSUB" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in agg:
    print("==", k)
    for c, v in sorted(agg[k].items()):
        print("  %-40s avg=%16.1f n=%d" % (c, sum(v)/len(v), len(v)))
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KSUB" in r["Name"]:
            print("time %-70s calls %s avg %.1f us min %.1f max %.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
rm -rf $OUT/g*/ $OUT/trace
