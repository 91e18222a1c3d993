python scripts/dev/p2_check.py check 2>&1 | tail -1
for c in "2,8,4,4" "2,8,3,4" "2,10,3,3" "2,12,3,3" "2,14,3,3" "2,14,2,3" "2,12,2,4" "4,6,2,3" "4,6,3,3" "1,14,4,3"; do for b in 2 4; do GNX_P2_TUNE=$c GNX_LR_BPC=$b python scripts/dev/p2_check.py bench 2>&1 | grep -E "config2|rror" | cut -c1-175; done; done
for c in "2,6,2,3" "2,6,3,3" "1,8,4,3" "1,14,4,3" "1,14,3,3" "1,12,4,3"; do GNX_P2_TUNE=$c python scripts/dev/p2_check.py bench12 2>&1 | grep -E "chr22|rror" | cut -c1-175; done
