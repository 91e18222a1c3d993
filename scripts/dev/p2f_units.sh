#!/bin/bash
# epilogue unit granularity of k_base_logistic_p2f at config 5a: GNX_LR_FLAGS bits 16-18 = classes per phase-1 unit, bits 20-22 = store parts
cd "${GRAFT_REPO_ROOT:-.}"
run() { "$@" python scripts/dev/p2_check.py c5 2>&1 | grep -E "config5|p2f cycles|MISMATCH|BAD" | tail -2 | sed -E 's/\(.*GB\/s of int8 X\)  //' | cut -c1-330; }
for pb in 4 3 2 1; do for nsp in 2 4; do
  f=$(( (pb << 16) | (nsp << 20) ))
  echo "== classes per unit $pb, store parts $nsp (GNX_LR_FLAGS=$f)"
  run env GNX_LR_FLAGS=$f
  run env GNX_LR_FLAGS=$f GNX_DEBUG=2 | grep "p2f cycles"
done; done
for f in 1 4 64 1024; do echo "== ablation $f"; run env GNX_LR_FLAGS=$f; done
