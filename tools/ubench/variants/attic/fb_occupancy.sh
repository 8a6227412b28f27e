# FB kernels by blocks per CU (NCHMM_EXP_FB_BLOCKS_PER_CU, experiment switch of nchmm_create): rocprofv3 kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in ${BLOCKS:-1 2}; do
  export NCHMM_EXP_FB_BLOCKS_PER_CU=$b
  rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/fbocc$b -o fb -- python3 $R/tools/bench_fwbw.py > $R/gpurun_out/fbocc$b.log 2>&1
  echo "blocks/CU=$b"; grep -h "scaled_kernel" $R/gpurun_out/fbocc$b/fb_kernel_stats.csv | cut -c1-200; grep -o '"shader_clock_mhz_under_load": [0-9]*' $R/gpurun_out/fbocc$b.log
done
