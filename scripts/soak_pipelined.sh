#!/bin/bash
# long pipelined runs of the slab application against the one-slab run (checksums of every frame's depth, the last images, the volume)
cd ${GRAFT_REPO_ROOT:-/root/repo}
APP=./apps/kinectfusion_slabs
COMMON="--res 128 --frames 300 --width 320 --height 240 --raycast exact"
ref=$($APP $COMMON --ranks 1 | grep -o "checksums.*hits=[0-9]*")
echo "ref: $ref"
fail=0
for cfg in "--ranks 8 --tiles 4 --pipeline 3 --transport threads-p2p --ghost auto --halo recompute" \
           "--ranks 8 --tiles 1 --pipeline 4 --transport threads --ghost 2 --halo exchange" \
           "--ranks 4 --tiles 8 --pipeline 2 --transport threads-p2p --ghost auto --halo exchange --inputs broadcast" \
           "--ranks 3 --tiles 4 --pipeline 3 --transport threads --ghost auto --halo recompute" \
           "--ranks 5 --tiles 2 --pipeline 4 --transport threads-p2p --ghost 2 --halo recompute" \
           "--ranks 2 --tiles 4 --pipeline 2 --transport threads-p2p --ghost auto --halo recompute"; do
  for rep in 1 2; do
    got=$(timeout 300 $APP $COMMON --driver frame $cfg | grep -o "checksums.*hits=[0-9]*")
    if [ "$got" != "$ref" ]; then echo "MISMATCH [$cfg] rep $rep: $got"; fail=1; else echo "ok [$cfg] rep $rep"; fi
  done
done
echo "soak_pipe fail=$fail"
