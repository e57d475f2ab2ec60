#!/bin/bash
# A/B of two kbench builds on the tile-stream kernel's layers (per-tile constants hoisted): q19.sh OLD NEW
export REPS=${REPS:-1000}
OLD=${1:-tools/kbench/conv_bench_old}; NEW=${2:-tools/kbench/conv_bench}
for B in 256 512; do
for a in "32 64 64 0 1" "32 64 64 0 0" "32 128 64 0 0" "32 32 32 0 1" "32 32 32 0 0" "32 96 32 0 0" "32 32 32 0 2"; do
  echo -n "old: "; RGFM_KB_SC=1 RGFM_KB_R=64 timeout -k 10 120 $OLD $a $B hx2q | tail -1 || exit 1
  echo -n "new: "; RGFM_KB_SC=1 RGFM_KB_R=64 timeout -k 10 120 $NEW $a $B hx2q | tr "\n" " " || exit 1
  echo
done
done
