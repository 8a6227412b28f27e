# 2 / 3 / 4 compute lanes (kVitLanes in nchmm_ctx.hpp, edited here and restored; batches in flight follow) on one box:
# tools/bench_ragged.py with DEPTH batches in flight and bench.py config 2.   bash tools/ubench/lanes_ab.sh  -> profiles/r04_lanes_ab.txt
cd $GRAFT_REPO_ROOT/nanocall_amd/csrc
cp nchmm_ctx.hpp /tmp/nchmm_ctx.hpp.tree
trap 'cp /tmp/nchmm_ctx.hpp.tree nchmm_ctx.hpp; make -s > /dev/null 2>&1' EXIT
run() { (cd $GRAFT_REPO_ROOT && DEPTH=$1 python tools/bench_ragged.py 2>/dev/null | sed "s/^/lanes $2 depth $1 /"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fwbw 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $2 Mevents/s', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'clock', d['device']['shader_clock_mhz_under_load'], 'end_to_end', d.get('end_to_end', {}).get('value'), 'one_call', d.get('end_to_end', {}).get('one_call', {}).get('value'))"); }
for L in 2 3 4; do
  sed -i "s/constexpr int kVitLanes = [0-9]*;/constexpr int kVitLanes = $L;/" nchmm_ctx.hpp
  make -s > /dev/null 2>&1
  [ $L = 3 ] && run 2 3
  run $L $L
done
