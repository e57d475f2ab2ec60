#!/bin/bash
# per-layer times (serial trace, one 512-row main-loop step) with and without the Winograd path
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-alt-mode --no-arith-check --no-parity-check"
run() {
  local tag=$1; shift
  rm -rf /tmp/kt_$tag
  ( export "$@" RGFM_OVERLAP=0; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$tag -- \
    python3 $R/bench.py --steps 1 --warmup 1 --euler-steps 4 --no-kernel-timers $B > /dev/null 2>&1 ) || exit 1
  python3 $R/tools/trace_layers.py $(find /tmp/kt_$tag -name '*kernel_trace.csv' | head -1) 512 538 5.69 > $O/r4_wino6_layers_$tag.txt 2>&1
}
run direct RGFM_WINO=0
run wino RGFM_WINO=1
paste <(awk '{print $1,$2,$3,$4,$6}' $O/r4_wino6_layers_direct.txt) <(awk '{print $6}' $O/r4_wino6_layers_wino.txt) | awk 'NR>1 && $5+0>0 && $6+0>0 && ($6/$5>1.03 || $6/$5<0.97) {print}' | head -40
grep "step total" $O/r4_wino6_layers_direct.txt $O/r4_wino6_layers_wino.txt
