#!/bin/bash
K=tools/kbench/conv_bench_qskew
export REPS=${REPS:-500}
for args in "32 64 64 0 0" "32 64 64 0 1" "16 128 128 0 1"; do
  for sk in 0 4 8 12 16 24; do
    echo "skew $sk x 4096 cycles"; SKEW=$sk timeout -k 10 120 $K $args 512 hx2q | grep -v "^check" || exit 1
  done
done
