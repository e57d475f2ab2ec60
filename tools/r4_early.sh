#!/bin/bash
# early request of chunk 0 (ahead of the GroupNorm table) in the pipelined kernel: small rows and the bench, previous library against this tree's
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
PREV=$PWD/tools/ab_prev/librgfm_hip_prev.so
( echo "prev:"; RGFM_LIB=$PREV timeout -k 10 200 python3 tools/small_rows.py 32 64 256 2>/dev/null
  echo "new:"; timeout -k 10 200 python3 tools/small_rows.py 32 64 256 2>/dev/null
  echo "prev:"; RGFM_LIB=$PREV timeout -k 10 200 python3 tools/small_rows.py 32 64 256 2>/dev/null
  echo "new:"; timeout -k 10 200 python3 tools/small_rows.py 32 64 256 2>/dev/null ) | tee $O/r4_early_small.txt
bash tools/ab_lib.sh tools/ab_prev/librgfm_hip_prev.so 2 | tee $O/r4_early_ab.txt
