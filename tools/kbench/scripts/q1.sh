#!/bin/bash
K=tools/kbench/conv_bench_q
export REPS=1000
for args in "32 64 64 0 0" "32 64 64 0 1" "32 128 64 0 2" "16 64 64 0 1" "32 192 64 0 2"; do
  echo "== base $args"; timeout -k 10 120 $K $args 512 hx2p || exit 1
  echo "== Q $args"; RGFM_HX2P_Q=1 timeout -k 10 120 $K $args 512 hx2p || exit 1
done
