#!/bin/bash
# layer sweep used while tuning the conv kernels (run on the GPU box from the repo root): run.sh [bx3|hx2|f32]
K=tools/kbench/conv_bench
W=${1:-hx2}
for args in "32 64 64 0 1" "32 192 64 0 0" "32 64 64 0 2" "16 128 128 0 1" "16 256 128 0 0" "8 128 128 0 1" "8 256 128 0 0" "32 128 128 2 0" "32 32 32 0 1" "32 96 32 0 0" "16 64 64 0 1"; do
  $K $args 512 $W || exit 1
done
