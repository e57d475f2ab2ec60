#!/bin/bash
export REPS=${REPS:-1000}
for args in "32 64 64 0 0" "32 128 64 0 0"; do
  for v in conv_bench_qs conv_bench_qa1 conv_bench_qa2 conv_bench_qa3 conv_bench_qa4; do
    echo -n "$v: "; timeout -k 10 120 tools/kbench/$v $args 512 hx2q | grep -v "^check" || exit 1
  done
done
