#!/bin/bash
# layer sweep used while tuning conv_mfma_bx3 (run on the GPU box from the repo root)
K=tools/kbench/conv_bench
for args in "32 64 64 0 1" "32 192 64 0 0" "32 64 64 0 2" "16 128 128 0 1" "16 256 128 0 0" "8 128 128 0 1" "32 128 128 2 0" "32 32 32 0 1" "32 96 32 0 0"; do
  $K $args 512 bx3
done
$K 32 64 64 0 1 512 f32
$K 16 256 128 0 0 512 f32
