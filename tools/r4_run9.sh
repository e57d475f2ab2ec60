#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
export REPS=1000
KD=tools/kbench/conv_bench_d
( for B in 512 32; do for a in "16 128 128 0 0" "16 128 128 0 1" "16 128 128 0 2" "16 64 64 0 0" "16 64 64 0 1" "16 64 64 0 2"; do
  echo -n "$a $B hx2d v2: "; RGFM_KB_R=256 RGFM_HX2D=2 timeout -k 10 60 $KD $a $B hx2d | tr "\n" " "; echo
done; done ) > $O/r4_kbench_d4.txt 2>&1
cat $O/r4_kbench_d4.txt | sed 's/check vs f32 kernel: //' | cut -c1-230
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/r4_t9.log 2>&1; echo "pytest rc $?" | tee -a $O/r4_t9.log
tail -5 $O/r4_t9.log
bash tools/ab_lib.sh tools/ab_prev/librgfm_hip_r03.so 2 2>&1 | tee $O/r4_ab_r03_r04b.txt
