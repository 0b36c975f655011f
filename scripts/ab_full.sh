#!/bin/bash
# like ab.sh but prints the whole output of the script for each library
R=$1; shift
LIBS=()
while [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; shift
for r in $(seq 1 $R); do
  for L in "${LIBS[@]}"; do
    echo "== $L"; FMPC_LIB=$PWD/mpc-sensorlessao_amd/lib/$L python3 "$@" 2>&1 | grep -v "amdgpu.ids\|tiled\|t < 1"
  done
done
