# In-block traceback parameters on one box: segment length (NCHMM_TB_SEG, compile time), speculative run-in (NCHMM_TB_MARGIN, run time),
# with the phase ticks of NCHMM_PROFILE=1 (traceback ticks, re-walked segments).   PROFILE=1 bash tools/ubench/tb_params_ab.sh
# -> profiles/r04_inblock_tb_params_ab.txt
cd $GRAFT_REPO_ROOT
echo "### margin 128 (default), variants of segment length and skew"
REPS=2 bash tools/ubench/vit_ab_defs.sh "" "-DNCHMM_TB_SEG=40" "-DNCHMM_TB_SEG=128" "-DNCHMM_VIT_SKEW=0"
echo "### margin 64"
NCHMM_TB_MARGIN=64 REPS=2 bash tools/ubench/vit_ab_defs.sh "" "-DNCHMM_TB_SEG=40"
echo "### margin 32"
NCHMM_TB_MARGIN=32 REPS=2 bash tools/ubench/vit_ab_defs.sh ""
