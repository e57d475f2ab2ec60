#!/bin/bash
# the 8x8 level: conv_mfma_hx2p_kernel against conv_mfma_hx2c_kernel (one barrier per chunk)
export REPS=${REPS:-500}
K=tools/kbench/conv_bench
for B in 5 512 256 32; do
for a in "8 128 128 0 1" "8 128 128 0 0" "8 256 128 0 0" "8 128 128 0 2"; do
  for k in hx2p hx2c; do
    echo -n "$k: "; RGFM_KB_R=256 RGFM_KB_SC=1 timeout -k 10 60 $K $a $B $k | tr "\n" " " || exit 1
    echo
  done
done
done
