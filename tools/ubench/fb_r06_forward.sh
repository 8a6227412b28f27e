# Round 6 side experiment: does the priority trick that took 3 % off the backward sweep do anything for the forward one (store-bound)?
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
{ bash tools/ubench/fb_ab_multi.sh tools/ubench/_fbv/f1_fwd_prio.hip tools/ubench/_fbv/f2_fwd_fold.hip tools/ubench/_fbv/f3_fwd_prio_fold.hip tools/ubench/_fbv/f4_fwd_store_prio.hip 2>&1; } > $OUT/fb_forward_session.txt 2>&1
tail -45 $OUT/fb_forward_session.txt
