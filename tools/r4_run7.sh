#!/bin/bash
# same-box A/B: round 3's library (commit 9dd5327, built from its sources) against this tree's, alternating
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
bash tools/ab_lib.sh tools/ab_prev/librgfm_hip_r03.so 3 2>&1 | tee $O/r4_ab_r03_r04.txt
for v in 1 0; do RGFM_HX2D=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-alt-mode --no-arith-check 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("RGFM_HX2D='$v'", round(d["value"],1))' | tee -a $O/r4_ab_r03_r04.txt; done
timeout -k 10 300 python tools/small_rows.py 32 64 > $O/r4_small_rows.txt 2>&1; cat $O/r4_small_rows.txt
RGFM_HX2D=0 timeout -k 10 300 python tools/small_rows.py 32 64 > $O/r4_small_rows_off.txt 2>&1; cat $O/r4_small_rows_off.txt
