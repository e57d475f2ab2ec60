#!/bin/bash
export REPS=${REPS:-1000}
for args in "32 64 64 0 0" "32 128 64 0 0"; do
  for v in conv_bench conv_bench_qs conv_bench_qm; do
    echo -n "$v: "; timeout -k 10 120 tools/kbench/$v $args 512 hx2q || exit 1
  done
done
