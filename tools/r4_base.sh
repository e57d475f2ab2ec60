#!/bin/bash
# round-4 baseline: tests, bench, kbench baseline of the layers this round works on
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r4_t1.log 2>&1; echo "pytest rc $?" | tee -a $O/r4_t1.log
tail -5 $O/r4_t1.log
timeout -k 10 600 python bench.py > $O/r4_bench1.json 2> $O/r4_bench1.err; echo "bench rc $?"
tail -c 600 $O/r4_bench1.json
export REPS=1000
K=tools/kbench/conv_bench
( for a in "8 128 128 0 1 512 hx2c" "8 128 128 0 0 512 hx2c" "8 256 128 0 0 512 hx2c" "8 128 128 0 1 32 hx2c" "8 128 128 0 0 256 hx2c" \
  "16 128 128 0 1 512 hx2p" "16 128 128 0 0 512 hx2p" "16 256 128 0 0 512 hx2p" "16 64 64 0 1 512 hx2p" "16 64 64 0 0 512 hx2p" \
  "32 64 64 0 1 512 hx2q" "32 64 64 0 2 512 hx2p" "32 192 64 0 0 512 hx2p"; do
  echo -n "$a: "; RGFM_KB_R=192 timeout -k 10 60 $K $a | tr "\n" " "; echo
done ) > $O/r4_kbench_base.txt 2>&1
cat $O/r4_kbench_base.txt
KD=tools/kbench/conv_bench_d
( for B in 512 32; do for a in "8 128 128 0 1" "8 128 128 0 0" "8 256 128 0 0" "16 128 128 0 1" "16 128 128 0 0" "16 256 128 0 0" "16 64 64 0 1" "16 64 64 0 0"; do
  for k in hx2d hx2c hx2p; do echo -n "$a $B $k: "; timeout -k 10 60 $KD $a $B $k | tr "\n" " "; echo; done
done; done ) > $O/r4_kbench_d.txt 2>&1
cat $O/r4_kbench_d.txt
