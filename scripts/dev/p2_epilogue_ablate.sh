for f in 0 1 4; do for m in bench c5; do GNX_LR_FLAGS=$f python scripts/dev/p2_check.py $m 2>&1 | grep -E "config" | sed "s/^/flags=$f /" | cut -c1-150; done; done
