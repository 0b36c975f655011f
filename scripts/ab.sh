#!/bin/bash
# Same-box A/B of library builds (the clock a box holds differs between boxes by several per cent):
#   bash scripts/ab.sh <rounds> <lib A> <lib B> ... -- <python script and args>
R=$1; shift
LIBS=()
while [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; shift
for r in $(seq 1 $R); do
  for L in "${LIBS[@]}"; do
    echo "== $L"; FMPC_LIB=$PWD/mpc-sensorlessao_amd/lib/$L python3 "$@" 2>&1 | grep "wave "
  done
done
