#!/bin/bash
# same-box A/B of two builds of the library (same ABI): ab_lib.sh PREV.so [runs]  -- alternating bench.py calls
PREV=${1:-tools/ab_prev/librgfm_hip.so}; N=${2:-2}
F="--no-cpu-baseline --no-alt-mode --no-arith-check"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(sys.argv[1], round(d["value"],1), "img/s", round(d["ms_per_step"],1), "ms; conv busy", round(r["busy_ms"]/d["steps"],1), "ms")'
for i in $(seq $N); do
  (RGFM_LIB=$PWD/$PREV timeout -k 10 300 python3 bench.py $F 2>/dev/null | python3 -c "$P" "prev:") || exit 1
  (timeout -k 10 300 python3 bench.py $F 2>/dev/null | python3 -c "$P" "new: ") || exit 1
done
