#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for nb in 32 64; do
  rm -rf /tmp/sr_$nb
  (cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sr_$nb -- python3 tools/small_rows_trace.py run $nb > /dev/null 2>&1) || exit 1
  python3 $R/tools/small_rows_trace.py show /tmp/sr_$nb > $O/r4_small_trace_$nb.txt
  head -1 $O/r4_small_trace_$nb.txt
done
cd $R && timeout -k 10 200 python3 tools/small_rows.py 32 64 2>/dev/null | tee $O/r4_small_rows.txt
