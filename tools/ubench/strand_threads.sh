# The reference's call shapes from 1 .. 4096 worker threads, through the header swap (how many calls in flight fill the GPU):
#   tools/bench_cpp_layer      Viterbi::fill, one strand per call            -> nchmm_viterbi_strand
#   tools/bench_train_threads  Parameter_Trainer::train_one_round, one read  -> nchmm_fwbw_windows
# -> profiles/r04b_strand_combiner_threads.txt
cd $GRAFT_REPO_ROOT
make -C tools bench_cpp_layer bench_train_threads > /dev/null 2>&1
for T in 1 16 64 256 1024 2048 4096; do R=$((T*4)); [ $R -lt 64 ] && R=64; [ $R -gt 8192 ] && R=8192; timeout 300 tools/bench_cpp_layer $R 5000 $T 2>&1 | tail -1; done
timeout 600 tools/bench_train_threads 8192 1 16 64 256 1024 2>&1 | tail -5
