#!/bin/bash
# Round-6 verdict item 3: the per-launch VALU instruction count of the tiled SdfFuse kernels with the LDS-DMA staging (in-tree library)
# and with round 5's staging through registers (build_ab/regstage), C3 / S_room (1280x960) and C2 / S_room: separate --pmc passes.
# Usage: scripts/fuse_stage_pmc.sh <tag>
TAG=${1:-r06_c3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export PMC_GROUPS_FILE=scripts/pmc_groups_stage.txt
for lib in dma regstage; do
  if [ $lib = regstage ]; then export KFX_LIB_PATH=$ROOT/build_ab/regstage/libkfx.so; else unset KFX_LIB_PATH; fi
  bash scripts/pmc_any.sh $TAG/${lib}_1280x960 k_sdf_fuse_tiled scripts/fuse_only.py room 20 1280 960 fast > gpurun_out/$TAG.${lib}_1280x960.txt 2>&1
  bash scripts/pmc_any.sh $TAG/${lib}_640x480 k_sdf_fuse_tiled scripts/fuse_only.py room 20 640 480 fast > gpurun_out/$TAG.${lib}_640x480.txt 2>&1
done
unset KFX_LIB_PATH
tail -n +1 gpurun_out/$TAG.*.txt
