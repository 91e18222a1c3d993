#!/bin/bash
# compile-time variants of k_base_logistic_p2f, rebuilt on the box and timed at config 5a:
#   VARIANTS="-DGNX_P2F_PD=1;-DGNX_P2F_PD=3;-DGNX_P2F_FLUSH_GROUP=2" bash scripts/dev/p2f_defs.sh
# GNX_P2F_PD: prefetch distance (tiles) of the plane-read pipeline; GNX_P2F_FLUSH_GROUP: accumulator registers gathered at a time at a
# window's end; GNX_P2F_PBR / GNX_P2F_NSP: classes per phase-1 unit / store parts of the epilogue waves.  The last build stays on the box only.
cd "${GRAFT_REPO_ROOT:-.}"
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-fast-math -ffp-contract=off"
run() { "$@" python scripts/dev/p2_check.py c5 2>&1 | grep -E "config5|p2f cycles|MISMATCH|BAD" | tail -2 | sed -E 's/\(.*GB\/s of int8 X\)  //' | cut -c1-330; }
IFS=';' read -ra V <<< "${VARIANTS:-}"
for d in "" "${V[@]}"; do
  echo "== ${d:-default}"
  (cd gnomix_amd/csrc && rm -f k_base_logistic_p2.o && make CXXFLAGS="$BASE $d" >/dev/null 2>&1) || { echo build failed; continue; }
  run env; run env
  [ -n "$DBGTOO" ] && run env GNX_DEBUG=2 | grep "p2f cycles"
done
