#!/bin/bash
# phase stamps of the tile-stream kernel after the epilogue changes (conv_bench_qprof = -DRGFM_HX2Q_PROF)
export REPS=${REPS:-300}
for a in "32 64 64 0 1 512" "32 64 64 0 0 512" "32 32 32 0 1 512" "32 128 64 0 0 512"; do
  for sc in 0 1; do
    echo "== $a small_check $sc"; RGFM_KB_SC=$sc timeout -k 10 120 tools/kbench/conv_bench_qprof $a hx2q | grep -v "^check" || exit 1
  done
done
