#!/bin/bash
# chunk-sized units (HX2P_PAIRN_HALF_C) against unit-sized ones on the under-filled 128-channel launches
export REPS=${REPS:-1000}
K=tools/kbench/conv_bench
for a in "8 128 128 0 0 512" "8 128 128 0 1 512" "8 256 128 0 2 512" "8 128 128 0 1 256" "8 256 128 0 2 256" "16 128 128 0 1 128" "16 256 128 0 2 128" "16 128 128 0 1 64"; do
  for c in 0 1; do
    echo -n "chunk $c: "; RGFM_HX2P_CHUNK=$c timeout -k 10 120 $K $a hx2p || exit 1
  done
done
