# run length of the 2-bit logistic pass at the default tile shape: config 2 and config 5a (GNX_LR_P2_RUN = 256 | 512)
for rs in 256 512; do for m in bench c5; do GNX_LR_P2_RUN=$rs GNX_DEBUG=1 python scripts/dev/p2_check.py $m 2>&1 | grep -E "config|k_base_logistic_p2<" | sort -u | sed "s/^/run=$rs /" | cut -c1-200; done; done
