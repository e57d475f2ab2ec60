#!/bin/bash
# per-kernel time of one bench call (run on the GPU box from the repo root): kstats.sh [pattern]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf /tmp/ks && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- \
  python3 bench.py --steps 1 --warmup 1 > gpurun_out/kstats.log 2>&1 || exit 1
python3 - "$1" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/ks/**/*kernel_stats.csv", recursive=True)[0]
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for r in csv.DictReader(open(f)):
    if pat in r["Name"]:
        print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:8.1f} us  {r["Percentage"]:>6s} %')
PY
