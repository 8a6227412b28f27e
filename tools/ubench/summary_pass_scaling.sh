# Round 6: what bounds the summary pass (FAST5 -> summaries) on the box?  32 000 FAST5 files (24 distinct reads of 2 x 5000 events),
# `nanocall --no-train --no-basecall` (host work only, no device) by reader processes and host threads; then four and eight
# such runs at once over disjoint quarters / eighths of the files (what the workers of a multi-GPU run do).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
T=$(mktemp -d ${SP_BASE:-/tmp}/sp_XXXX)
python - $T <<'PY'
import os, sys, shutil, subprocess
sys.path[:0] = ['.', 'oracle', 'tests']
import oracle_pipeline as op
tmp = sys.argv[1]
os.makedirs(tmp + '/reads')
if not os.path.exists('tools/make_fast5'):
    subprocess.run(['make', '-C', 'tools', 'make_fast5'], check=True, capture_output=True)
for k in range(24):
    ed = op.synth_ed_table("r73", 5000, 5000, seed=100 + k, hairpin=8, complement_model="r73.c.p1.006.ont.model" if k % 2 else "r73.c.p2.006.ont.model")
    ev = f'{tmp}/seed{k}.events'; op.write_events_table(ev, ed, 4000.0, f'read-{k}')
    subprocess.run(['tools/make_fast5', ev, f'{tmp}/seed{k}.fast5'], check=True)
for r in range(32000):
    shutil.copy(f'{tmp}/seed{r % 24}.fast5', f'{tmp}/reads/r{r:06d}.fast5')
for w in (4, 8):
    for k in range(w):
        with open(f'{tmp}/fofn_{w}_{k}', 'w') as f:
            f.write(''.join(f'{tmp}/reads/r{r:06d}.fast5\n' for r in range(k, 32000, w)))
PY
run() { nanocall_amd/bin/nanocall --pore r73 --no-train --no-basecall "$@" 2>&1 | grep stage_wall | sed 's/.*init_files_s=\([0-9.]*\).*init_reads_s=\([0-9.]*\).*/init_files \1 init_reads \2/'; }
for rp in 8 16 32 64 128; do echo "one process, -t 128, reader-procs $rp: $(run -t 128 --reader-procs $rp $T/reads)"; done
for t in 16 32 64; do echo "one process, -t $t, reader-procs 64: $(run -t $t --reader-procs 64 $T/reads)"; done
for w in 4 8; do
  echo "== $w processes at once, each -t $((128 / w)) --reader-procs $((128 / w)) over 1/$w of the files"
  s=$(date +%s.%N)
  for k in $(seq 0 $((w - 1))); do ( echo "   $k: $(run -t $((128 / w)) --reader-procs $((128 / w)) $T/fofn_${w}_$k)" ) & done
  wait
  e=$(date +%s.%N); python3 -c "print('   wall %.2f s for 32000 files = %.1f k files/s' % ($e - $s, 32 / ($e - $s)))"
done
rm -rf $T
