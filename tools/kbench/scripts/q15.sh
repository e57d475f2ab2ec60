#!/bin/bash
# where the fused 1x1-skip chunks' time goes: timing ablations (conv_bench_qa1..4 = -DRGFM_HX2Q_ABL=1..4) of the tile-stream
# kernel on 64 -> 64 at 32x32 with no skip / skip from 128 / skip from 192 channels (results wrong by construction)
export REPS=${REPS:-1000}
for r in "1 64" "2 128" "2 192"; do
  set -- $r
  for k in conv_bench conv_bench_qa1 conv_bench_qa2 conv_bench_qa3 conv_bench_qa4; do
    echo -n "res=$1 R=$2 $k: "; RGFM_KB_R=$2 timeout -k 10 120 tools/kbench/$k 32 64 64 0 $1 512 hx2q | tail -1 || exit 1
  done
done
