cd $GRAFT_REPO_ROOT
echo "### margin 128 (default), variants of segment length and skew"
REPS=2 bash tools/ubench/vit_ab_defs.sh "" "-DNCHMM_TB_SEG=40" "-DNCHMM_TB_SEG=128" "-DNCHMM_VIT_SKEW=0"
echo "### margin 64"
NCHMM_TB_MARGIN=64 REPS=2 bash tools/ubench/vit_ab_defs.sh "" "-DNCHMM_TB_SEG=40"
echo "### margin 32"
NCHMM_TB_MARGIN=32 REPS=2 bash tools/ubench/vit_ab_defs.sh ""
