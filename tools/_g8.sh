cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
READS=2000 EVENTS=3000 THREADS=32 python tools/bench_cli.py > gpurun_out/r02g_bench_cli.json 2> gpurun_out/r02g_bench_cli.err; cat gpurun_out/r02g_bench_cli.json; tail -3 gpurun_out/r02g_bench_cli.err
READS=8000 EVENTS=5000 THREADS=32 NCHMM_DEBUG=1 python tools/bench_cli.py > gpurun_out/r02g_bench_cli_8k.json 2>> gpurun_out/r02g_bench_cli.err; cat gpurun_out/r02g_bench_cli_8k.json; grep nchmm_base gpurun_out/r02g_bench_cli.err | tail -3
