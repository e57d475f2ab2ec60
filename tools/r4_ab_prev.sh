#!/bin/bash
# parity subset, then the previous library (tools/ab_prev/librgfm_hip_prev.so) against this tree's on the bench, alternating
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet or sampler or variants or golden or probe or bitwise or batch" > $O/r4_abp_tests.log 2>&1 || { tail -30 $O/r4_abp_tests.log; exit 1; }
tail -2 $O/r4_abp_tests.log
bash tools/ab_lib.sh tools/ab_prev/librgfm_hip_prev.so ${1:-3} | tee $O/r4_abp_ab.txt
