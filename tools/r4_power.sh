#!/bin/bash
# is the chip at its power cap during the bench?  rocm-smi samples beside a running bench.py
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
python3 -c "import torch" 2>/dev/null
(timeout -k 10 400 python3 bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-alt-mode --no-arith-check --no-parity-check > $O/r4_power_bench.json 2>/dev/null) &
BP=$!
: > $O/r4_power_during.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>&1 | grep -i "Power (W)\|sclk" | tr '\n' ' ' >> $O/r4_power_during.txt; echo >> $O/r4_power_during.txt
  sleep 3
done
wait $BP
head -c 200 $O/r4_power_bench.json; echo
cat $O/r4_power_during.txt
