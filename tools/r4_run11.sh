#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
F="--no-cpu-baseline --no-alt-mode --no-arith-check --no-parity-check"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(sys.argv[1], round(d["value"],1), "img/s", round(d["ms_per_step"],1), "ms; conv busy", round(r["busy_ms"]/d["steps"],1), "ms")'
for rep in 1 2; do
for e in "" "RGFM_GN=table" "RGFM_GRAPH=1" "RGFM_HX2C=0" ; do
  ( env $e timeout -k 10 300 python3 bench.py $F 2>/dev/null | python3 -c "$P" "[$e]" ) | tee -a $O/r4_switch_ab.txt
done; done
