cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for D in 0 1 2 4 8 3 7 15; do
  RGFM_WINO_DBG=$D RGFM_WINO=1 RGFM_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$D -- python bench.py --steps 1 --warmup 0 --euler-steps 2 --no-cpu-baseline --no-kernel-timers > gpurun_out/abl_$D.log 2>&1
  echo "dbg=$D $(grep 'conv_wino_kernel<0, false>' gpurun_out/abl_$D/*/*_kernel_stats.csv | cut -d, -f2-4,7)"
done
