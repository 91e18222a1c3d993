# A/B of rank-smoother variants on the GPU box: GNX_* knobs are read once per context, one process per variant
cd "${GRAFT_REPO_ROOT:-.}"
for v in "GNX_SM_PAIR=0" "GNX_SM_PAIR=1" "GNX_SM_PAIR=1 GNX_RK_RPL=2" "GNX_SM_PAIR=1 GNX_RK_RPL=4" "GNX_SM_PAIR=0 GNX_RK_RPL=2" $EXTRA_VARIANTS; do
  env $v TAG="$v" WHICH=infer python scripts/dev/bench_kernels.py 2>&1 | grep smooth
done
