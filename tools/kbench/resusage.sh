#!/bin/bash
# register / scratch use of the conv kernels as hipcc reports it: resusage.sh [extra -D flags]  (device-only compile)
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I../../ratio_guided_multimodal_fm_amd/csrc \
  --cuda-device-only -c conv_bench.hip -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  awk '/Function Name:/ {n=$0; sub(/.*Function Name: /,"",n); sub(/ \[.*/,"",n)}
       / VGPRs:/ {v=$0; sub(/.*VGPRs: /,"",v); sub(/ \[.*/,"",v)}
       /ScratchSize/ {s=$0; sub(/.*lane\]: /,"",s); sub(/ \[.*/,"",s); if (n ~ /conv_mfma_hx2[pqscd]?_kernel/) print n, "vgpr", v, "scratch", s}'
