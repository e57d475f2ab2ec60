#!/bin/bash
# upsampling convs as four parity classes (RGFM_UP_T2): parity subset, then same-box bench A/B against the nine-tap form
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet or sampler or variants or golden or probe or fm or flow" > $O/r4_up1_tests.log 2>&1 || { tail -30 $O/r4_up1_tests.log; exit 1; }
tail -3 $O/r4_up1_tests.log
F="--no-cpu-baseline --no-alt-mode --no-arith-check"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(sys.argv[1], round(d["value"],1), "img/s", round(d["ms_per_step"],1), "ms; conv busy", round(r["busy_ms"]/d["steps"],1), "ms; parity", d.get("parity_check"))'
for i in 1 2; do
  (RGFM_UP_T2=0 timeout -k 10 300 python3 bench.py $F 2>/dev/null | python3 -c "$P" "nine taps:     ") || exit 1
  (timeout -k 10 300 python3 bench.py $F 2>/dev/null | python3 -c "$P" "parity classes:") || exit 1
done 2>&1 | tee $O/r4_up1_ab.txt
