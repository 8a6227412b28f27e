# Round 6, VERDICT item 7: ONE bounded session on the backward sweep's 1.9 -> 1.6 ms gap.
#   1. the phase timer (fb_phases.py): where a wave's event goes
#   2. same-box A/B of the variants under tools/ubench/_fbv/ (fb_ab_multi.sh: bench leg x3 + rocprofv3 per-kernel averages)
#   3. parity of the two furthest-reaching variants (tests/test_fwbw_gpu.py with the variant built in)
# bash tools/ubench/fb_r06_session.sh   (GPU box, repo root); everything lands in gpurun_out/r06/fb_session.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p $OUT
cd $R
{
echo "#### phases (tree)"; python tools/ubench/fb_phases.py 2>&1 | tail -12
echo "#### A/B"; bash tools/ubench/fb_ab_multi.sh tools/ubench/_fbv/v1_pm_fma.hip tools/ubench/_fbv/v2_pm_fma_prio.hip tools/ubench/_fbv/v3_pm_fma_emis_fold.hip tools/ubench/_fbv/v4_all_prio.hip 2>&1
for V in v3_pm_fma_emis_fold v4_all_prio; do
  echo "#### parity with $V built in"
  cp nanocall_amd/csrc/fwbw_scaled_kernel.hip /tmp/fwbw_scaled_kernel.hip.keep
  cp tools/ubench/_fbv/$V.hip nanocall_amd/csrc/fwbw_scaled_kernel.hip
  (cd nanocall_amd/csrc && make -s > /dev/null 2>&1)
  python -m pytest tests/test_fwbw_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
  cp /tmp/fwbw_scaled_kernel.hip.keep nanocall_amd/csrc/fwbw_scaled_kernel.hip
  (cd nanocall_amd/csrc && make -s > /dev/null 2>&1)
done
} > $OUT/fb_session.txt 2>&1
tail -70 $OUT/fb_session.txt
