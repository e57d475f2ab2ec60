#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/r4_t4.log 2>&1; echo "pytest rc $?" | tee -a $O/r4_t4.log
tail -8 $O/r4_t4.log
for i in 1 2; do
timeout -k 10 600 python bench.py --no-cpu-baseline --no-alt-mode --no-arith-check > $O/r4_bench4_on$i.json 2> $O/r4_bench4.err; echo "bench rc $?"
RGFM_HX2D=0 timeout -k 10 600 python bench.py --no-cpu-baseline --no-alt-mode --no-arith-check > $O/r4_bench4_off$i.json 2>> $O/r4_bench4.err; echo "bench rc $?"
done
python - <<'PY'
import json
for f in ("on1","off1","on2","off2"):
    d=json.loads(open(f"gpurun_out/r4_bench4_{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["roofline"]["achieved"], d["parity_check"]["max_abs"])
PY
