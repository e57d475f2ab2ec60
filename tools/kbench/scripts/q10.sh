#!/bin/bash
export REPS=${REPS:-1000}
K=tools/kbench/conv_bench
for args in "32 32 32 0 0" "32 32 32 0 1" "32 64 32 0 0" "32 96 32 0 0" "32 32 32 0 2" "32 64 64 0 0" "32 64 64 0 1"; do
  timeout -k 10 120 $K $args 512 hx2p | grep -v "^check" || exit 1
  for t in 1 4; do
    echo -n "tpw $t: "; RGFM_HX2Q_TPW=$t timeout -k 10 120 $K $args 512 hx2q || exit 1
  done
done
for args in "32 32 32 0 1" "32 96 32 0 0"; do
  for b in 256 128; do
  timeout -k 10 120 $K $args $b hx2p | grep -v "^check" || exit 1
  echo -n "q: "; RGFM_HX2Q_MIN=1 timeout -k 10 120 $K $args $b hx2q | grep -v "^check" || exit 1
  done
done
