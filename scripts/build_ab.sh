#!/bin/bash
# An A/B build of libkfx.so with extra compiler flags, beside the in-tree one: build_ab/<name>/libkfx.so (git-ignored; travels to the
# GPU box with gpurun).  Load it with KFX_LIB_PATH=build_ab/<name>/libkfx.so (kangaroo_amd/_lib.py).
# Usage: scripts/build_ab.sh <name> <extra flags...>      e.g. scripts/build_ab.sh regstage -DKFX_FUSE_STAGE_DMA=0
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$ROOT/build_ab/$NAME
mkdir -p $D/x/csrc
ln -sfn $ROOT/include $D/include
cp $ROOT/kangaroo_amd/csrc/*.hip $ROOT/kangaroo_amd/csrc/*.h $ROOT/kangaroo_amd/csrc/*.inc $ROOT/kangaroo_amd/csrc/*.cpp $ROOT/kangaroo_amd/csrc/Makefile $D/x/csrc/
make -C $D/x/csrc -j6 EXTRA="$*" ../libkfx.so ../libkfx_debug.so > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
cp $D/x/libkfx.so $D/x/libkfx_debug.so $D/
echo "built $D/libkfx.so with: $*"
