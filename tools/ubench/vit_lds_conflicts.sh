# Round 5: are the LDS bank conflicts of viterbi_kernel (SQ_LDS_BANK_CONFLICT ~ 75 % of its LDS-active cycles) the publish
# stores, and do they cost anything?  PMC pass (SQ_LDS_BANK_CONFLICT, SQ_ACTIVE_INST_LDS, SQ_WAIT_ANY) of the tree's kernel and of
# variants/viterbi_padded_exchange.hip.txt (exchange planes of 256 + 8 entries: the h = 0 / h = 1 halves of a 16-lane store group on
# different banks), same library otherwise, serialised config-2 launches.    bash tools/ubench/vit_lds_conflicts.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/nanocall_amd/csrc
# whatever ends this script (an error, an interrupt, a time limit): the tree's own object is rebuilt, never a variant's left behind
trap 'rm -f viterbi_kernel.o; make -s > /dev/null 2>&1' EXIT
FLAGS=$(make -s print-hipflags)
BENCH="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fwbw --no-end-to-end --serial-launches --no-shard-leg"
pmc() {   # $1 = label
  rm -rf /tmp/pmc_$1; (cd /tmp && TMPDIR=/tmp rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES -d /tmp/pmc_$1 -o vit -- $BENCH > /tmp/pmc_$1.log 2>&1)
  python3 - $1 <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = 0
for f in glob.glob(f"/tmp/pmc_{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "viterbi_kernel" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n += 1
launches = max(1, n // max(1, len(tot)))
print(sys.argv[1], {k: f"{v / launches:.4g}" for k, v in sorted(tot.items())}, "per launch over", launches, "launches")
PY
}
echo "== tree"; pmc tree
cp $R/tools/ubench/variants/viterbi_padded_exchange.hip.txt /tmp/viterbi_variant.hip
/opt/rocm/bin/hipcc $FLAGS -I$R/nanocall_amd/csrc -c /tmp/viterbi_variant.hip -o viterbi_kernel.o && make -s > /dev/null 2>&1
echo "== padded exchange"; pmc padded
rm -f viterbi_kernel.o; make -s > /dev/null 2>&1
