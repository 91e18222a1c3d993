# phase ablation of k_smooth_xgb_bs at config 2 (timing only: flagged runs compute garbage)
# GNX_BS_FLAGS: 1 = no trees (phases B and C), 2 = no C, 4 = no counter read / sort (A1, A3), 8 = no row build (A4)
for f in ${FLAGS:-0 1 2 3 4 12 15}; do
  GNX_BS_FLAGS=$f python scripts/dev/bs_check.py bench 2>&1 | grep "^bs" | sed "s/^/flags=$f /"
done
