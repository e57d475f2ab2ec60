#!/bin/bash
# cost of the output's low-range check (ConvArgs::small_check) in the epilogue
export REPS=${REPS:-1000}
K=tools/kbench/conv_bench
for a in "32 64 64 0 1 512 hx2q" "32 32 32 0 1 512 hx2q" "16 128 128 0 1 512 hx2p" "16 256 128 0 0 512 hx2p" "8 128 128 0 1 512 hx2p" "16 64 64 0 1 512 hx2p" "32 192 64 0 0 512 hx2p"; do
  for sc in 0 1; do
    echo -n "small_check $sc: "; RGFM_KB_SC=$sc timeout -k 10 120 $K $a | tail -1 || exit 1
  done
done
