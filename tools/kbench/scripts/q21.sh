#!/bin/bash
# the Downsample convs (stride 2, raw input): conv_mfma_hx2_kernel<*, CONV_S2, *> against conv_mfma_hx2s_kernel
export REPS=${REPS:-500}
K=tools/kbench/conv_bench
for a in "16 64 64 1 0 5" "8 128 128 1 0 5" "16 32 32 1 0 7" "8 64 64 1 0 3" "16 64 64 1 0 512" "8 128 128 1 0 512" "16 32 32 1 0 512" "16 64 64 1 0 256" "8 128 128 1 0 256" "16 32 32 1 0 256" "16 64 64 1 0 32" "8 128 128 1 0 32" "16 32 32 1 0 32"; do
  for k in hx2 hx2s; do
    echo -n "$k: "; timeout -k 10 60 $K $a $k | tr "\n" " " || exit 1
    echo
  done
done
