#!/bin/bash
# A/B of two kbench builds over the layer sweep: q13.sh OLD NEW  (VALU diet of the staging path)
export REPS=${REPS:-1000}
OLD=${1:-tools/kbench/conv_bench_old}; NEW=${2:-tools/kbench/conv_bench}
for a in "32 64 64 0 1 512 hx2q" "32 192 64 0 0 512 hx2p" "32 128 64 0 2 512 hx2q" "32 32 32 0 1 512 hx2q" "32 96 32 0 0 512 hx2q" "16 128 128 0 1 512 hx2p" "16 256 128 0 0 512 hx2p" "8 128 128 0 1 512 hx2p" "16 64 64 0 1 512 hx2p"; do
  echo -n "old: "; timeout -k 10 120 $OLD $a || exit 1
  echo -n "new: "; timeout -k 10 120 $NEW $a || exit 1
done
