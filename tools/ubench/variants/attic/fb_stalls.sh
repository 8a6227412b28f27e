# Where does one block of the FB backward sweep wait?  Rebuilds fwbw_scaled_kernel.o with one experiment switch at a time
# (results are garbage by construction) and times the kernels at 1 and 2 blocks per CU.   bash tools/ubench/fb_stalls.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/nanocall_amd/csrc
# the switches are kept out of the kernel source; the shipped file is kept aside and comes back on any exit
patch --dry-run -s -p0 < ../../tools/ubench/exp_switches_fwbw_scaled.patch > /dev/null || { echo "exp_switches_fwbw_scaled.patch no longer applies to the shipped kernel" >&2; exit 1; }
cp fwbw_scaled_kernel.hip /tmp/fwbw_scaled_kernel.hip.orig
trap 'cp /tmp/fwbw_scaled_kernel.hip.orig fwbw_scaled_kernel.hip; rm -f *.rej *.orig; make -s > /dev/null 2>&1' EXIT
patch -s -p0 < ../../tools/ubench/exp_switches_fwbw_scaled.patch
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -Wall -Wno-unused-result -fno-slp-vectorize"
for v in ${VARIANTS:-BASE NCHMM_EXP_NOLOAD NCHMM_EXP_NOBARRIER NCHMM_EXP_NOEXP NCHMM_EXP_NODPP}; do
  /opt/rocm/bin/hipcc $FLAGS -D$v -c fwbw_scaled_kernel.hip -o fwbw_scaled_kernel.o && make -s > /dev/null 2>&1
  echo "== $v"
  (cd $R && BLOCKS="1 2" bash tools/ubench/fb_occupancy.sh 2>&1 | grep -E "blocks/CU|backward_scaled|forward_scaled" | cut -d, -f1,4 )
done

