#!/bin/bash
# samples sclk / power while one layer runs back to back (development tool)
K=tools/kbench/conv_bench
which=${1:-bx3}
( for i in 1 2 3 4 5 6 7 8 9 10 11 12; do $K 16 256 128 0 0 512 $which; done > /tmp/pp_$which.log 2>&1 ) &
PID=$!
sleep 2
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | head -4
  sleep 1
done
wait $PID
tail -2 /tmp/pp_$which.log
