#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
export REPS=1000
KD=tools/kbench/conv_bench_d
( for a in "16 128 128 0 0" "16 256 128 0 0" "16 64 64 0 0" "16 128 64 0 0" "16 32 64 0 0" "8 128 128 0 0"; do for B in 512 256; do for po in 0 1 2; do
  k=hx2p; [ "${a:0:1}" = "8" ] && k=hx2c
  echo -n "$a $B $k pout=$po: "; if [ $po = 0 ]; then timeout -k 10 60 $KD $a $B $k | tail -1; else RGFM_KB_POUT=$po timeout -k 10 60 $KD $a $B $k | tail -1; fi
done; done; done ) > $O/r4_kbench_pout.txt 2>&1
cat $O/r4_kbench_pout.txt | cut -c1-200
