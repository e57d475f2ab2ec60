#!/bin/bash
K=tools/kbench/conv_bench_qprof
export REPS=${REPS:-500}
for args in "32 64 64 0 0" "32 64 64 0 1" "32 192 64 0 2" "16 128 128 0 1" "16 256 128 0 2"; do
  timeout -k 10 120 $K $args 512 hx2q || exit 1
done
