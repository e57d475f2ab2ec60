#!/bin/bash
# fixed cost of an 8x8-level launch: time against the number of chunks (Cin = 32 .. 256), both kernels, 32 and 512 rows
export REPS=${REPS:-500}
K=tools/kbench/conv_bench
for B in 32 512; do
for cin in 32 64 128 256; do
  for k in hx2p hx2c; do
    echo -n "$k: "; timeout -k 10 60 $K 8 $cin 128 0 0 $B $k | tail -1 || exit 1
  done
done
done
