#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
export REPS=1000
KD=tools/kbench/conv_bench_d
( for B in 512 32; do for a in "8 128 128 0 1" "8 128 128 0 0" "8 256 128 0 0" "16 128 128 0 0" "16 64 64 0 1" "16 64 64 0 0"; do
  for k in hx2d; do echo -n "$a $B $k: "; timeout -k 10 60 $KD $a $B $k | tr "\n" " "; echo; done
done; done ) > $O/r4_kbench_d2.txt 2>&1
cat $O/r4_kbench_d2.txt | sed 's/check vs f32 kernel: //' | cut -c1-220
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/r4_t2.log 2>&1; echo "pytest rc $?" | tee -a $O/r4_t2.log
tail -8 $O/r4_t2.log
