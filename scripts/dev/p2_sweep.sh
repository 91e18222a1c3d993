python scripts/dev/p2_check.py check 2>&1 | tail -1
for c in "4,8,1" "4,8,2" "4,8,13" "4,8,102" "2,8,2" "2,8,3" "2,8,102"; do IFS=, read m w x <<< "$c"; for f in 0 6; do GNX_LR_NBUF=$x GNX_LR_FLAGS=$f GNX_LR_TUNE=$m,$w GNX_LR_BPC=4 python scripts/dev/p2_check.py bench 2>&1 | grep config2 | sed "s/^/xsn=$x flags=$f /" | cut -c1-150; done; done
GNX_PMC_SETS=tcp python scripts/dev/pmc.py 'k_base_logistic_p2' -- python scripts/dev/p2_check.py bench 2>&1 | grep -v "^   derived" | cut -c1-400
