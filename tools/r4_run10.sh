#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
export REPS=1000
KD=tools/kbench/conv_bench_d
( for B in 512 32; do for a in "16 128 128 0 0" "16 128 128 0 1" "16 64 64 0 0" "16 64 64 0 1" "8 128 128 0 0" "8 256 128 0 0"; do
  for v in 2 1; do echo -n "$a $B hx2d v$v: "; RGFM_HX2D=$v timeout -k 10 60 $KD $a $B hx2d | tr "\n" " "; echo; done
done; done ) > $O/r4_kbench_d5.txt 2>&1
cat $O/r4_kbench_d5.txt | sed 's/check vs f32 kernel: //' | cut -c1-230
