#!/bin/bash
# config 5a's 2-bit base pass (k_base_logistic_p2f), ablations and per-role cycle counters on ONE box (timing only: the ablated runs'
# outputs are wrong by construction).  gpurun -- 'bash scripts/dev/p2f_ablate.sh > gpurun_out/r06_c5a_ablations.txt'
# GNX_LR_FLAGS: 1 raw logits (no sigmoid / normaliser), 2 no MFMA, 4 no flush at all, 8 no X loads, 16 no plane loads, 64 no limb gather /
# combine / park, 128 no epilogue-wave priority, 1024 no stores to B; switches that KEEP the output: 2048 no 16-byte float32 stores,
# 1 << 25 the flat kernel declines (int8 kernels run), bits 16-18 / 20-22 classes per sigmoid unit / store parts.
# GNX_P2_TUNE="2,8,ew,2,nbuf": block shape.  GNX_LR_P2_FLAT=0: the two-pass slot-tile kernel of round 5.  (Compile-time variants —
# prefetch distance of the plane reads — are scripts/dev/p2f_defs.sh; two builds alternating on one box: scripts/dev/p2_ab.sh.)
cd "${GRAFT_REPO_ROOT:-.}"
run() { "$@" python scripts/dev/p2_check.py c5 2>&1 | grep -E "config5|p2f cycles" | tail -3 | sed -E 's/\(.*GB\/s of int8 X\)  //' | cut -c1-330; }
echo "== default (flat tiles, 4 epilogue waves, 3-step plane ring, ~18 blocks per CU)"; run env
echo "== two-pass slot tiles (round 5's kernel)"; run env GNX_LR_P2_FLAT=0
for f in 1 2 4 8 16 64 128 1024; do echo "== GNX_LR_FLAGS=$f"; run env GNX_LR_FLAGS=$f; done
for f in 2048 33554432 131072 393216; do echo "== GNX_LR_FLAGS=$f (output kept)"; run env GNX_LR_FLAGS=$f; done
for t in 2,8,2,2,3 2,8,0,2,3 2,8,4,2,2; do echo "== GNX_P2_TUNE=$t"; run env GNX_P2_TUNE=$t; done
for w in 24 32 64; do echo "== GNX_LR_WANT=$w (window ranges)"; run env GNX_LR_WANT=$w GNX_DEBUG=1; done
echo "== cycle counters per wave role (GNX_DEBUG=2: the instrumented instantiation), default / raw logits / no flush"
for f in 0 1 4; do run env GNX_DEBUG=2 GNX_LR_FLAGS=$f | grep "p2f cycles"; done
echo "== one block, step by step (GNX_DEBUG=6): cycles from a barrier's release to each role's arrival at the next"
GNX_DEBUG=6 python scripts/dev/p2_check.py c5 2>&1 | grep -A44 "p2f trace" | cut -c1-200
