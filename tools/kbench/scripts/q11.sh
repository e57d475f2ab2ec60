#!/bin/bash
# every stride-1 GroupNorm'd layer shape of the two U-Nets: hx2p against the stream kernel's workgroup cuts
export REPS=${REPS:-600}
K=tools/kbench/conv_bench
B=${B:-512}
run() {  # args..., cuts...
  local args="$1"; shift
  timeout -k 10 120 $K $args $B hx2p | grep -v "^check" || exit 1
  for c in "$@"; do
    echo -n "  cut $c: "; RGFM_HX2Q_MIN=1 RGFM_HX2Q_CUT=$c timeout -k 10 120 $K $args $B hx2q | grep -v "^check" | sed 's/hx2q S=.*B=[0-9]*://' || exit 1
  done
}
for a in "32 64 64 0 0" "32 64 64 0 1" "32 128 64 0 0" "32 192 64 0 0" "32 64 64 0 2"; do run "$a" 21 12; done
for a in "16 64 128 0 0" "16 128 128 0 1" "16 128 128 0 0" "16 256 128 0 0" "16 192 128 0 0" "16 128 128 0 2"; do run "$a" 22 12 21; done
for a in "16 32 64 0 0" "16 64 64 0 1" "16 64 64 0 0" "16 128 64 0 0" "16 96 64 0 0" "16 64 64 0 2"; do run "$a" 21 12; done
