# A/B of logistic-pass variants on the GPU box (knobs are read once per context: one process per variant)
cd "${GRAFT_REPO_ROOT:-.}"
for v in "GNX_LR_DL=0" "GNX_LR_DL=1" "GNX_LR_DL=1 GNX_LR_NBUF=2" $EXTRA_VARIANTS; do
  env $v TAG="$v" WHICH=base python scripts/dev/bench_kernels.py 2>&1 | grep base_logistic
done
for v in "GNX_LR_DL=0" "GNX_LR_DL=1" "GNX_LR_DL=1 GNX_LR_TUNE=1,8" "GNX_LR_DL=1 GNX_LR_TUNE=1,8 GNX_LR_NBUF=2" "GNX_LR_DL=0 GNX_LR_TUNE=1,8"; do
  env $v TAG="$v" python scripts/dev/bench_a12.py 2>&1 | grep "A=12"
done
