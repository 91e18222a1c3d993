# ablations of the 2-bit logistic pass at config 2 and config 5a (GNX_LR_FLAGS, timing only: 1 raw logits (no sigmoid), 2 no MFMA,
# 4 no flush (no combine / park / epilogue / stores), 8 no X loads, 16 no plane loads)
for f in ${FLAGS:-0 1 4 8 16 2}; do for m in ${MODES:-bench c5}; do GNX_LR_FLAGS=$f python scripts/dev/p2_check.py $m 2>&1 | grep -E "config" | sed "s/^/flags=$f /" | cut -c1-150; done; done
