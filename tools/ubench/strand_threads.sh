# Viterbi::fill, one strand per call, from 1 .. 4096 worker threads (tools/bench_cpp_layer): how many strands in flight fill the GPU through nchmm_viterbi_strand.
# -> profiles/r04b_strand_combiner_threads.txt
cd $GRAFT_REPO_ROOT
make -C tools bench_cpp_layer > /dev/null 2>&1
for T in 1 16 64 256 1024 2048 4096; do R=$((T*4)); [ $R -lt 64 ] && R=64; [ $R -gt 8192 ] && R=8192; timeout 300 tools/bench_cpp_layer $R 5000 $T 2>&1 | tail -1; done
