#!/bin/bash
# PMC sweep of the fuse/raycast kernels: one rocprofv3 pass per counter group (counters only,
# with --kernel-trace), each pass under its own timeout.
# Usage: scripts/pmc_sweep.sh <tag> <scene> [env assignments for the workload...]
TAG=${1:-pmc}; SCENE=${2:-full}; shift 2 || true
for kv in "$@"; do export "$kv"; done
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $ROOT/scripts/fuse_only.py $SCENE 3 > /dev/null 2> $OUT/g$i.err || echo "group $i failed: $grp"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ","").replace("kfx::","")[:40]
        if "k_sdf_fuse" in k or "k_raycast" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in agg:
    print("==", k)
    for c, v in sorted(agg[k].items()):
        print("  %-36s avg=%16.1f n=%d" % (c, sum(v)/len(v), len(v)))
PY
rm -rf $OUT/g*/
