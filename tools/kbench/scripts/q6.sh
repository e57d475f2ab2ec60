#!/bin/bash
K=tools/kbench/conv_bench
export REPS=${REPS:-1000}
for args in "32 64 64 0 0" "32 64 64 0 1" "32 128 64 0 0" "32 192 64 0 2" "16 64 64 0 1" "16 128 128 0 1" "16 256 128 0 2"; do
  timeout -k 10 120 $K $args 512 hx2p | grep -v "^check" || exit 1
  for t in 1 2 4; do
    echo "tpw $t"; RGFM_HX2Q_TPW=$t timeout -k 10 120 $K $args 512 hx2q || exit 1
  done
done
REPS=300
for args in "32 64 64 0 0" "32 64 64 0 1"; do
  RGFM_HX2Q_TPW=4 timeout -k 10 120 tools/kbench/conv_bench_qprof $args 512 hx2q | grep -v "^check" || exit 1
done
