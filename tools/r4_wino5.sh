#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
F="--no-cpu-baseline --no-alt-mode --no-arith-check --no-parity-check"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"],1), "img/s", round(d["ms_per_step"],1), "ms")'
for i in 1 2; do
  (timeout -k 10 300 python3 bench.py $F 2>/dev/null | python3 -c "$P" "default:           ") || exit 1
  (RGFM_WINO=1 RGFM_WINO_W32=1 timeout -k 10 300 python3 bench.py $F 2>/dev/null | python3 -c "$P" "winograd at 32x32: ") || exit 1
  (RGFM_WINO=1 timeout -k 10 300 python3 bench.py $F 2>/dev/null | python3 -c "$P" "winograd, all:     ") || exit 1
done 2>&1 | tee $O/r4_wino5_ab.txt
