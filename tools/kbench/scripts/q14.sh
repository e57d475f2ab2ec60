#!/bin/bash
# the decoder's second convs (64 -> 64 at 32x32 with the fused 1x1 skip from the 192 / 128-channel concat input) and the
# MNIST ones (32 -> 32, skip from 96 / 64): pipelined kernel against the tile-stream kernel
export REPS=${REPS:-1000}
K=tools/kbench/conv_bench
for B in 256 512; do
for a in "192 32 64 64" "128 32 64 64" "96 32 32 32" "64 32 32 32" "128 16 64 64" "96 16 64 64"; do
  set -- $a
  for k in hx2p hx2q; do
    echo; echo -n "R=$1 $k: "; RGFM_KB_R=$1 timeout -k 10 120 $K $2 $3 $4 0 2 $B $k | tr "\n" " " || exit 1
  done
done
done
