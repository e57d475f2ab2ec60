#!/bin/bash
# the MNIST net's 16x16 level (Cout = 64) at pre-phase batch sizes: pipelined kernel (four-wave cut below 256 workgroups)
# against the tile stream with the tile threshold lifted
export REPS=${REPS:-1000}
K=tools/kbench/conv_bench
for B in 128 256 512; do
for a in "16 64 64 0 1" "16 64 64 0 0" "16 128 64 0 0" "16 96 64 0 0" "16 32 64 0 0"; do
  echo -n "hx2p:        "; timeout -k 10 120 $K $a $B hx2p | tail -1 || exit 1
  for t in 1 2; do
    echo -n "hx2q tpw $t:  "; RGFM_HX2Q_MIN=1 RGFM_HX2Q_TPW=$t timeout -k 10 120 $K $a $B hx2q | tail -1 || exit 1
  done
done
done
