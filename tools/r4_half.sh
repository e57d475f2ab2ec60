#!/bin/bash
# threshold below which launches are cut into twice the workgroups (RGFM_HX2P_HALF): bench, alternating
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
F="--no-cpu-baseline --no-alt-mode --no-arith-check --no-parity-check"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(sys.argv[1], round(d["value"],1), "img/s", round(d["ms_per_step"],1), "ms")'
for i in 1 2; do for T in 256 129 65 0; do
  (RGFM_HX2P_HALF=$T timeout -k 10 300 python3 bench.py $F 2>/dev/null | python3 -c "$P" "half<$T:") || exit 1
done; done | tee $O/r4_half_ab.txt
