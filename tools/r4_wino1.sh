#!/bin/bash
# Winograd prototype (conv_mfma_hx2w.hip) against the dispatched direct kernels, one layer at a time
cd $GRAFT_REPO_ROOT/tools/kbench; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
K=./conv_bench_w
( for B in 512; do for a in "16 128 128 0 0" "16 128 128 0 1" "16 256 128 0 0" "32 64 64 0 0" "32 64 64 0 1" "32 192 64 0 0" "32 128 64 0 0" "16 64 64 0 1"; do
  for w in hx2w hx2p hx2q; do echo -n "$a $B $w: "; REPS=${REPS:-300} timeout -k 10 60 $K $a $B $w 2>&1 | tr "\n" " "; echo; done
done; done ) > $O/r4_wino1.txt 2>&1
cat $O/r4_wino1.txt | sed 's/check vs f32 kernel: //' | cut -c1-250
