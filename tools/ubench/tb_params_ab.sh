# In-block traceback parameters on one box: segment length (kTbSeg, edited into a copy of the kernel by vit_ab_defs.sh), speculative
# run-in (NCHMM_TB_MARGIN, run time), with the phase ticks of NCHMM_PROFILE=1 (traceback ticks, re-walked segments).
#   PROFILE=1 bash tools/ubench/tb_params_ab.sh      -> profiles/r04_inblock_tb_params_ab.txt
# (that record also has a run with a deliberate lead of one block of a CU over the other, NCHMM_VIT_SKEW, which changed nothing and
# was taken out of the kernel afterwards)
cd $GRAFT_REPO_ROOT
echo "### margin 128, variants of segment length"
NCHMM_TB_MARGIN=128 REPS=2 bash tools/ubench/vit_ab_defs.sh "" "kTbSeg=40" "kTbSeg=128"
echo "### margin 64 (the default)"
NCHMM_TB_MARGIN=64 REPS=2 bash tools/ubench/vit_ab_defs.sh "" "kTbSeg=40"
echo "### margin 32"
NCHMM_TB_MARGIN=32 REPS=2 bash tools/ubench/vit_ab_defs.sh ""
