cd "${GRAFT_REPO_ROOT:-.}"
for v in "GNX_LR_DL=0" "GNX_LR_DL=1" "GNX_LR_DL=1 GNX_LR_TUNE=2,12" "GNX_LR_DL=1 GNX_LR_TUNE=2,12 GNX_LR_BPC=3" "GNX_LR_DL=1 GNX_LR_TUNE=2,12 GNX_LR_BPC=6"; do
  env $v TAG="$v" WHICH=base python scripts/dev/bench_kernels.py 2>&1 | grep base_logistic
done
