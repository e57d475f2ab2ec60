#!/bin/bash
# the four-wave cut of the Winograd kernel (RGFM_HX2W_CUT=2) against the eight-wave cut and the direct kernel
cd $GRAFT_REPO_ROOT/tools/kbench; O=$GRAFT_REPO_ROOT/gpurun_out
( for B in ${BS:-512}; do for a in "16 128 128 0 0" "16 128 128 0 1" "16 256 128 0 0" "32 192 64 0 0" "32 128 64 0 0" "32 64 64 0 0"; do
  for w in "hx2w 1" "hx2p 0"; do set -- $w; echo -n "$a $B $1 cut$2: "; RGFM_HX2W_CUT=$2 RGFM_KB_GN=1 REPS=${REPS:-300} timeout -k 10 60 ./conv_bench_w $a $B $1 2>&1 | tr "\n" " "; echo; done
done; done ) > $O/r4_wino4.txt 2>&1
sed 's/check vs f32 kernel: max|diff| //; s/stats rel diff //; s/(fp32-equivalent)//; s/range flag 0 //' $O/r4_wino4.txt | cut -c1-210
