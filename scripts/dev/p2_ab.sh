#!/bin/bash
# A/B of two builds of the library on ONE box, alternating: the 2-bit base pass at MODE = bench (config 2: chr22, A = 7), c5 (config 5a:
# chr1, A = 12), bench12, aligned (scripts/dev/p2_check.py).  gnomix_amd/libgnomix_hip_prev.so = the build to compare with (copy it
# there before editing; GNX_LIBRARY picks the library).  Prints the runs and the two means.
cd "${GRAFT_REPO_ROOT:-.}"
A=$PWD/gnomix_amd/libgnomix_hip.so; B=$PWD/gnomix_amd/libgnomix_hip_prev.so
for i in $(seq 1 ${REPS:-4}); do
  for L in $A $B; do
    echo -n "$(basename $L) "; GNX_LIBRARY=$L "$@" python scripts/dev/p2_check.py ${MODE:-c5} 2>&1 | grep -E "config|chr22|aligned" | sed -E 's/.*int8 ([0-9.]+) ms  p2 ([0-9.]+) ms.*identical=(\w+).*/int8 \1 p2 \2 \3/'
  done
done | tee /tmp/ab.txt
awk '{s[$1]+=$5; n[$1]++} END {for (k in s) printf "%s mean p2 %.4f ms over %d runs\n", k, s[k]/n[k], n[k]}' /tmp/ab.txt
