#!/bin/bash
# stride-1 layers at 32 rows (what one rank of an 8-GPU run integrates in the pre-phase): the pipelined kernel with
# unit-sized / chunk-sized units, and the tile stream with its tile threshold lifted
export REPS=${REPS:-500}
K=tools/kbench/conv_bench
for a in "32 64 64 0 1 32" "32 192 64 0 0 32" "16 128 128 0 1 32" "16 256 128 0 0 32" "8 128 128 0 1 32" "8 256 128 0 0 32" "16 64 64 0 1 32" "32 32 32 0 1 32"; do
  echo "== $a"
  for c in 0 1; do echo -n "  hx2p chunk-units $c: "; RGFM_HX2P_CHUNK=$c timeout -k 10 60 $K $a hx2p | tail -1 || exit 1; done
  echo -n "  hx2q tpw 1:         "; RGFM_HX2Q_MIN=1 RGFM_HX2Q_TPW=1 timeout -k 10 60 $K $a hx2q | tail -1
done
