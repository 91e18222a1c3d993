cd "${GRAFT_REPO_ROOT:-.}"
for v in "X=1" "GNX_LR_BPC=1" "GNX_LR_TUNE=2,8"; do
  env $v TAG="$v" python scripts/dev/bench_a12.py 2>&1 | grep "A=12"
done
A=20 TAG="A20" python scripts/dev/bench_anyA.py | tail -1
A=20 GNX_LR_BPC=4 TAG="A20 bpc4" python scripts/dev/bench_anyA.py | tail -1
