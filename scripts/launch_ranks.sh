#!/bin/bash
# Starts N ranks of a native program on one node, one per GPU, with the environment torchrun would give them
# (RANK, WORLD_SIZE, LOCAL_RANK), e.g. the RCCL transport of the slab application:
#   scripts/launch_ranks.sh 8 apps/kinectfusion_slabs --transport rccl --res 1024 --raycast composite
# Exit code: the first non-zero rank exit code.
N=${1:?usage: launch_ranks.sh N program [args...]}; shift
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
RDV=${KFX_RENDEZVOUS:-/tmp/kfx_slabs.$$.id}
rm -f "$RDV"
pids=()
for ((r = 0; r < N; r++)); do
  RANK=$r WORLD_SIZE=$N LOCAL_RANK=$r "$@" --rendezvous "$RDV" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || { c=$?; [ $rc -eq 0 ] && rc=$c; }; done
rm -f "$RDV"
exit $rc
