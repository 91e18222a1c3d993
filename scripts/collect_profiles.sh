#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'bash scripts/collect_profiles.sh'
# then on the dev side:  python scripts/summarize_prof.py gpurun_out profiles r01c
# One --kernel-trace --stats pass, then SEPARATE --pmc passes (counters only, never combined with sys/runtime traces),
# as /opt/skills/guides/MI355X_MICROARCH.md prescribes for HBM traffic (FETCH_SIZE / WRITE_SIZE) on gfx950.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
BENCH="python bench.py --steps 3 --warmup 1 --cpu-seconds 0"
mkdir -p gpurun_out
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq gpurun_out/prof_tc
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o stats -- python bench.py --steps 20 --warmup 3 --cpu-seconds 0 > gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o c -- $BENCH > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -o c -- $BENCH > gpurun_out/prof_write.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/prof_sq -o c -- $BENCH > gpurun_out/prof_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d gpurun_out/prof_tc -o c -- $BENCH > gpurun_out/prof_tc.log 2>&1
find gpurun_out/prof_* -name "*.csv" | head -20
tail -1 gpurun_out/prof_stats.log | cut -c1-200
