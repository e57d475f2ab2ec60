#!/bin/bash
# where the Winograd kernel's fixed cost is: no epilogue (ABL 1) / no K loop (ABL 2) against the whole kernel
cd $GRAFT_REPO_ROOT/tools/kbench; O=$GRAFT_REPO_ROOT/gpurun_out
( for a in "16 128 128 0 0" "16 256 128 0 0" "32 192 64 0 0"; do
  for k in conv_bench_w conv_bench_wa1 conv_bench_wa2; do echo -n "$a $k: "; RGFM_KB_GN=1 REPS=300 timeout -k 10 60 ./$k $a 512 hx2w 2>&1 | tail -1; done
done ) 2>&1 | tee $O/r4_wino7.txt
