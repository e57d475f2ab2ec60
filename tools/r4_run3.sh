#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
export REPS=1000
KD=tools/kbench/conv_bench_d
( for B in 512 32; do for a in "8 128 128 0 1" "8 128 128 0 0" "8 256 128 0 0" "8 128 128 0 2" "16 128 128 0 0" "16 128 128 0 1" "16 128 128 0 2" "16 64 64 0 1" "16 64 64 0 2"; do
  for v in 1 2; do echo -n "$a $B hx2d v$v: "; RGFM_KB_R=256 RGFM_HX2D=$v timeout -k 10 60 $KD $a $B hx2d | tr "\n" " "; echo; done
done; done
for a in "8 128 128 0 0" "8 256 128 0 0"; do for po in 0 1 2; do
  echo -n "$a 512 hx2c pout=$po: "; if [ $po = 0 ]; then timeout -k 10 60 $KD $a 512 hx2c | tr "\n" " "; else RGFM_KB_POUT=$po timeout -k 10 60 $KD $a 512 hx2c | tr "\n" " "; fi; echo
done; done ) > $O/r4_kbench_d3.txt 2>&1
cat $O/r4_kbench_d3.txt | sed 's/check vs f32 kernel: //' | cut -c1-230
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/r4_t3.log 2>&1; echo "pytest rc $?" | tee -a $O/r4_t3.log
tail -8 $O/r4_t3.log
timeout -k 10 600 python bench.py --no-cpu-baseline --no-alt-mode > $O/r4_bench3.json 2> $O/r4_bench3.err; echo "bench rc $?"
RGFM_HX2D=0 timeout -k 10 600 python bench.py --no-cpu-baseline --no-alt-mode > $O/r4_bench3_off.json 2> $O/r4_bench3_off.err; echo "bench rc $?"
python - <<'PY'
import json
for f in ("gpurun_out/r4_bench3.json","gpurun_out/r4_bench3_off.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["roofline"]["achieved"], d["parity_check"]["max_abs"])
PY
