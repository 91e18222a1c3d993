cd "${GRAFT_REPO_ROOT:-.}"
for v in "GNX_FOREST_WRUN=16 ROUNDS=20" "GNX_FOREST_WRUN=12 ROUNDS=20" "GNX_FOREST_WRUN=24 ROUNDS=20" "GNX_FOREST_WRUN=16 ROUNDS=20 GNX_FOREST_FLAGS=1" $EXTRA_VARIANTS; do
  echo "$v: $(env $v python scripts/dev/forest_occ.py 2>&1 | tail -1)"
done
