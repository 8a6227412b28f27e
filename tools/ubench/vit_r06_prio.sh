# Round 6 side experiment: the per-phase priority that took 3-6 % off the two forward-backward sweeps, on the wide Viterbi sweep
# (which balances its two blocks per chunk of events today).  RUNS=3 bash tools/ubench/vit_r06_prio.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
{ RUNS=3 TESTS=1 bash tools/ubench/vit_ab_multi.sh tools/ubench/_vitv/a_phase_prio_only.hip tools/ubench/_vitv/b_phase_prio_plus_chunk.hip tools/ubench/_vitv/c_phase_prio_3_1.hip 2>&1; } > $OUT/vit_prio_session.txt 2>&1
cat $OUT/vit_prio_session.txt
