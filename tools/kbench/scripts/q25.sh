#!/bin/bash
# A/B of two kbench builds on the 16x16 and 8x8 layers of the pipelined kernel (slot rotation key hx >> 2 / hx >> 1)
export REPS=${REPS:-1000}
OLD=${1:-tools/kbench/conv_bench_old}; NEW=${2:-tools/kbench/conv_bench}
for a in "16 128 128 0 1 512" "16 256 128 0 0 512" "16 128 128 0 0 512" "16 64 64 0 1 512" "16 128 64 0 0 512" "8 128 128 0 2 512" "16 256 128 0 2 512" "32 192 64 0 0 512" "16 128 128 0 1 256"; do
  echo -n "old: "; RGFM_KB_R=256 timeout -k 10 120 $OLD $a hx2p | tail -1 || exit 1
  echo -n "new: "; RGFM_KB_R=256 timeout -k 10 120 $NEW $a hx2p | tr "\n" " " || exit 1
  echo
done
