#!/bin/bash
# Same-box A/B of wave-kernel library variants on the per-problem-factor leg (explicit start):
#   bash scripts/ab_wave.sh <rounds> <batch> <n_newton> "<lib suffix>[:FLAGS]" ...     (suffix "" = production library)
R=$1; B=$2; NW=$3; shift 3
for r in $(seq 1 $R); do
  for V in "$@"; do
    L=${V%%:*}; F=""; case "$V" in *:*) F=${V#*:};; esac
    LIBF=$PWD/mpc-sensorlessao_amd/lib/libfastmpc${L:+_$L}.so
    echo -n "[$V] "; FMPC_PERF_WAVE_ONLY=1 FMPC_WAVE_FLAGS=$F FMPC_LIB=$LIBF python3 scripts/general_perf.py $B $NW 10 2>&1 | grep "^wave "
  done
done
