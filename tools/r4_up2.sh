#!/bin/bash
# per-layer times of the upsampling convs (serial trace, one main-loop step at 512 rows): nine taps / parity classes on the
# pipelined kernel / parity classes on the un-pipelined kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-alt-mode --no-arith-check --no-parity-check"
run() {  # tag, env...
  local tag=$1; shift
  rm -rf /tmp/kt_$tag
  env "$@" RGFM_OVERLAP=0 true
  ( export "$@" RGFM_OVERLAP=0; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$tag -- \
    python3 $R/bench.py --steps 1 --warmup 1 --euler-steps 4 --no-kernel-timers $B > /dev/null 2>&1 ) || exit 1
  python3 $R/tools/trace_layers.py $(find /tmp/kt_$tag -name '*kernel_trace.csv' | head -1) 512 538 5.69 > $O/r4_up2_layers_$tag.txt 2>&1
  echo "== $tag"; grep "up[0-9]\|step total" $O/r4_up2_layers_$tag.txt
}
run nine RGFM_UP_T2=0
run t2p RGFM_DUMMY=1
run t2u RGFM_HX2P=0
