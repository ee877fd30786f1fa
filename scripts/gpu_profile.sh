#!/bin/bash
# Runs on the GPU box (via gpurun): bench lines + rocprofv3 kernel-trace stats + PMC passes.
# Usage: scripts/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$*"
python3 $ROOT/bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json
# per-kernel time: kernel trace + stats (no PMC in this pass)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS --no-cpu-baseline --no-extra-legs > $OUT/trace_bench.json 2> $OUT/trace.err
# HBM traffic: separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS --no-cpu-baseline --no-extra-legs --steps 10 --warmup 2 --prime-seconds 0.05 --prime-cap-seconds 0.3 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS --no-cpu-baseline --no-extra-legs --steps 10 --warmup 2 --prime-seconds 0.05 --prime-cap-seconds 0.3 > /dev/null 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | head -20
python3 $ROOT/scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep the merged directory small: drop the raw per-dispatch traces, keep stats + summaries
find $OUT -name "*kernel_trace.csv" -size +8M -delete
