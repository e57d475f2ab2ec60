#!/bin/bash
# Winograd kernel in the library: kbench with the consumer-side norm, parity subset, bench A/B (RGFM_WINO=0 / default)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
( cd tools/kbench; for a in "16 128 128 0 1" "32 192 64 0 0"; do echo -n "$a gn: "; RGFM_KB_GN=1 REPS=200 timeout -k 10 60 ./conv_bench_w $a 512 hx2w 2>&1 | tr "\n" " "; echo; done ) | sed 's/check vs f32 kernel: //' | cut -c1-220
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/r4_wino2_tests.log 2>&1 || { tail -30 $O/r4_wino2_tests.log; exit 1; }
tail -2 $O/r4_wino2_tests.log
F="--no-cpu-baseline --no-alt-mode --no-arith-check"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(sys.argv[1], round(d["value"],1), "img/s", round(d["ms_per_step"],1), "ms; conv busy", round(r["busy_ms"]/d["steps"],1), "ms; parity", d["parity_check"]["max_abs"])'
for i in 1 2; do
  (RGFM_WINO=0 timeout -k 10 300 python3 bench.py $F 2>/dev/null | python3 -c "$P" "direct:  ") || exit 1
  (timeout -k 10 300 python3 bench.py $F 2>/dev/null | python3 -c "$P" "winograd:") || exit 1
done 2>&1 | tee $O/r4_wino2_ab.txt
