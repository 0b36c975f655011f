#!/bin/bash
# Build a variant of ONE kernel source into its own library for same-box A/B runs (scripts/ab.sh):
#   bash scripts/build_variant.sh <name> <source.hip> "<extra compiler flags>"
# -> mpc-sensorlessao_amd/lib/libfastmpc_<name>.so (every other object from the production build, or from the timing build
#    when the flags contain -DFW_TIMING).  Run `make -C mpc-sensorlessao_amd/csrc [timing]` first.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; FLAGS=$3
CS=$ROOT/mpc-sensorlessao_amd/csrc; LIB=$ROOT/mpc-sensorlessao_amd/lib
BASE=obj; case "$FLAGS" in *FW_TIMING*) BASE=obj_timing;; esac
mkdir -p $LIB/obj_$NAME
STEM=$(basename $SRC .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $FLAGS -c $CS/$SRC -o $LIB/obj_$NAME/$STEM.o 2>&1 | grep -E "error|spill" || true
OBJS=$(ls $LIB/$BASE/*.o | grep -v "/$STEM.o")
HOSTO=""; [ -f $LIB/$BASE/fmpc_host.o ] || HOSTO=$LIB/obj/fmpc_host.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $HOSTO $LIB/obj_$NAME/$STEM.o -o $LIB/libfastmpc_$NAME.so
echo "built $LIB/libfastmpc_$NAME.so"
