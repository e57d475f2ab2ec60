#!/bin/bash
export REPS=${REPS:-1000}
for args in "32 64 64 0 0" "32 64 64 0 1" "32 128 64 0 0" "32 192 64 0 2" "16 64 64 0 1"; do
  timeout -k 10 120 tools/kbench/conv_bench $args 512 hx2p | grep -v "^check" || exit 1
  for v in conv_bench conv_bench_qs conv_bench_qp conv_bench_qsp; do
    echo -n "$v: "; timeout -k 10 120 tools/kbench/$v $args 512 hx2q | grep -v "^check" || exit 1
  done
done
