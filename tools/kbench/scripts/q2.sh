#!/bin/bash
# hx2q (four waves per SIMD) against hx2p, same box, steady clocks
K=tools/kbench/conv_bench
export REPS=${REPS:-1000}
for args in "32 64 64 0 0" "32 64 64 0 1" "32 128 64 0 2" "32 192 64 0 2" "16 64 64 0 1" "16 128 128 0 1" "16 256 128 0 2" "16 64 128 0 2" "32 128 128 2 0" "16 128 128 2 0" "32 64 64 2 0"; do
  for w in hx2p hx2q; do
    timeout -k 10 120 $K $args 512 $w || exit 1
  done
done
