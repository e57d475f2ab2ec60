#!/bin/bash
# The round's profile set, run on the GPU box in ONE gpurun call:
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r02'
# writes gpurun_out/<tag>_*; copy what should be judged into profiles/.
# rocprofv3 is given the python program itself (no env/bash hop after `--`), counters in their own passes.
set -o pipefail
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-alt-mode --no-arith-check --no-parity-check"

# 1. the official line (default flags: cpu_baseline, alternative mode, arithmetic check all on)
( cd $R && timeout -k 10 900 python3 bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench_err.txt ) || exit 1
echo "bench line done"

# 2. kernel stats of one call, two streams (per-launch durations include overlap with the other net)
rm -rf /tmp/ks && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- \
  python3 $R/bench.py --steps 1 --warmup 1 --no-kernel-timers $B > /dev/null 2>&1 || exit 1
cp $(find /tmp/ks -name '*kernel_stats.csv' | head -1) $O/${TAG}_bench_kernel_stats.csv
echo "kernel stats done"

# 3. serial trace (one stream) -> per-layer table of one main-loop step
rm -rf /tmp/kt && RGFM_OVERLAP=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- \
  python3 $R/bench.py --steps 1 --warmup 1 --euler-steps 4 --no-kernel-timers $B > /dev/null 2>&1 || exit 1
cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) $O/${TAG}_bench_kernel_stats_serial.csv
# per-layer roofline against the ceilings the bench line of step 1 measured in-process (f16 MFMA loop / 3, float4 copy)
CEIL=$(python3 -c "
import json
d=json.loads(open('$O/${TAG}_bench_line.json').read().strip().splitlines()[-1])['roofline']['measured_ceilings']
print(round(d['mfma_f16_tflops']/3,1), round(d['hbm_copy_gbs']/1000,2))")
python3 $R/tools/trace_layers.py $(find /tmp/kt -name '*kernel_trace.csv' | head -1) 512 $CEIL > $O/${TAG}_bench_layers_serial.txt 2>&1
echo "layers done"

# 4. PMC: clock + MFMA busy (own pass, kernel trace only)
rm -rf /tmp/pm && RGFM_OVERLAP=0 timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY \
  --kernel-trace --output-format csv -d /tmp/pm -- \
  python3 $R/bench.py --steps 1 --warmup 1 --euler-steps 3 --no-kernel-timers $B > /dev/null 2>&1 || exit 1
python3 $R/tools/pmc_mfma.py /tmp/pm > $O/${TAG}_conv_pmc_summary_hx2.json 2> $O/${TAG}_pmc_err.txt
echo "pmc mfma done"

# 5. PMC: HBM bytes (FETCH_SIZE, WRITE_SIZE in separate passes)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c && RGFM_OVERLAP=0 timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- \
    python3 $R/bench.py --steps 1 --warmup 1 --euler-steps 3 --no-kernel-timers $B > /dev/null 2>&1 || exit 1
done
( cd $R && python3 tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE 512 > $O/${TAG}_conv_traffic_hx2.json 2>> $O/${TAG}_pmc_err.txt )
echo "pmc traffic done"

# 6. phases of a call and the weak-scaling projection
( cd $R && timeout -k 10 300 python3 tools/phase_split.py --json $O/${TAG}_phase_split.json > $O/${TAG}_phase_split.txt 2>&1 )
tail -5 $O/${TAG}_phase_split.txt
python3 -c "
import json
d=json.loads(open('$O/${TAG}_bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline'].get('traffic'))"
