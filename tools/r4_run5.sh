#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/r4_t5.log 2>&1; echo "pytest rc $?" | tee -a $O/r4_t5.log
tail -5 $O/r4_t5.log
for i in 1 2; do
timeout -k 10 600 python bench.py --no-cpu-baseline --no-alt-mode --no-arith-check > $O/r4_bench5_$i.json 2> $O/r4_bench5.err; echo "bench rc $?"
done
python - <<'PY'
import json
for f in ("1","2"):
    d=json.loads(open(f"gpurun_out/r4_bench5_{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["roofline"]["achieved"], d["parity_check"]["max_abs"], {k:round(v["avg_launch_us"],1) for k,v in d["roofline"]["hbm_kernels"].items() if isinstance(v,dict)}, d["roofline"]["hbm_kernels"]["share_of_call"])
PY
