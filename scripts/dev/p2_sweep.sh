python scripts/dev/p2_check.py check 2>&1 | tail -1
for c in "4,8,2" "4,8,12" "4,8,11" "2,8,2" "2,8,12" "2,8,1"; do IFS=, read m w x <<< "$c"; for f in 0 6; do GNX_LR_NBUF=$x GNX_LR_FLAGS=$f GNX_LR_TUNE=$m,$w GNX_LR_BPC=4 python scripts/dev/p2_check.py bench 2>&1 | grep config2 | sed "s/^/xsn=$x flags=$f /"; done; done
