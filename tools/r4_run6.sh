#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/r4_t6.log 2>&1; echo "pytest rc $?" | tee -a $O/r4_t6.log
tail -5 $O/r4_t6.log
for i in 1 2; do
timeout -k 10 600 python bench.py --no-cpu-baseline --no-alt-mode --no-arith-check > $O/r4_bench6_on$i.json 2> $O/r4_bench6.err; echo "bench rc $?"
RGFM_DECOUPLE=0 timeout -k 10 600 python bench.py --no-cpu-baseline --no-alt-mode --no-arith-check > $O/r4_bench6_off$i.json 2>> $O/r4_bench6.err; echo "bench rc $?"
done
python - <<'PY'
import json
for f in ("on1","off1","on2","off2"):
    d=json.loads(open(f"gpurun_out/r4_bench6_{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["roofline"]["achieved"], d["parity_check"]["max_abs"], {k:round(v["avg_launch_us"],1) for k,v in d["roofline"]["hbm_kernels"].items() if isinstance(v,dict)}, d["roofline"]["hbm_kernels"]["share_of_call"], d["roofline"]["kernel_time_share"])
PY
