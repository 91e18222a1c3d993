#!/bin/bash
# Collect the rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 3000 -- 'bash scripts/collect_profiles.sh [bench] [modes] [cf] [rf] [c5a] [c5b] [c3] [c4]'
# (the script condenses the traces on the box: gpurun_out/prof_summary/ comes back; copy what is to be judged into profiles/)
# Per workload: ONE --kernel-trace --stats pass, then SEPARATE --pmc passes (counters only, never combined with sys/runtime
# traces), as /opt/skills/guides/MI355X_MICROARCH.md prescribes for HBM traffic (FETCH_SIZE / WRITE_SIZE) on gfx950.
#   bench   = bench.py (config 2: logistic + xgb smoother)          modes = chr22 with the CRF and CNN smoothers
#   cf / rf = chr22 with the boosted-tree / random-forest bases      c5a_p2 / c5a_int8 = chr1 WGS, A = 12, logistic + CRF, 25 000 haplotypes
#                                                                    (ONE launch size and ONE logistic kernel per process: a
#                                                                    kernel-stats average over launches of different sizes means nothing)
#   c5b     = Gnofix re-phasing loop (host staging included)         c3    = chr1 array, CovRSK/SVC base + xgb
#   c5br    = the same Gnofix workload device-resident, int8 and 2-bit rows (the counters README / DESIGN quote for k_gnofix)
#   smbs    = the tree smoother alone, k_smooth_xgb_rk against the parked k_smooth_xgb_bs (scripts/dev/bs_check.py bench; needs a
#             library built with `make -C gnomix_amd/csrc EXPERIMENTS=1`)
#   c4      = whole genome, 22 chromosome models (JSON only: scripts/bench_configs.py c4)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
WHICH="${@:-bench modes cf rf c5a_p2 c5a_int8 c5b c5br c3 c4}"
OUT=gpurun_out/prof
mkdir -p $OUT
SQ="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
TC="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
for cfg in $WHICH; do
  if [ "$cfg" = "bench" ]; then
    CMD="python bench.py --steps 3 --warmup 1 --passes 1 --cpu-seconds 0 --e2e-steps 0 --vcf-reps 0 --trained 0 --configs 0"
    STATS="python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --e2e-steps 0 --vcf-reps 0 --trained 0 --configs 0"
  elif [ "$cfg" = "smbs" ]; then   # the tree smoother alone: rank walk (pointer nodes) against the bit-sliced kernel (k_smooth_xgb_bs)
    CMD="python scripts/dev/bs_check.py bench"
    STATS="$CMD"
  elif [ "$cfg" = "c4" ]; then
    python scripts/bench_configs.py c4 > $OUT/c4.log 2>&1
    cp gpurun_out/bench_configs.json $OUT/c4.json
    tail -1 $OUT/c4.log | cut -c1-300
    continue
  else
    CMD="python scripts/bench_configs.py $cfg"
    STATS="$CMD"
  fi
  rm -rf $OUT/$cfg; mkdir -p $OUT/$cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$cfg/stats -o s -- $STATS > $OUT/$cfg/stats.log 2>&1
  grep '"config"' $OUT/$cfg/stats.log | tail -2 > $OUT/$cfg/result.jsonl
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$cfg/fetch -o c -- $CMD > $OUT/$cfg/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$cfg/write -o c -- $CMD > $OUT/$cfg/write.log 2>&1
  rocprofv3 --pmc $SQ --output-format csv -d $OUT/$cfg/sq -o c -- $CMD > $OUT/$cfg/sq.log 2>&1
  if [ "$cfg" = "bench" ] || [ "$cfg" = "c5a_p2" ] || [ "$cfg" = "c5a_int8" ]; then   # the L1 fill counters of the logistic passes (int8 and 2-bit) side by side
    rocprofv3 --pmc $TC --output-format csv -d $OUT/$cfg/tc -o c -- $CMD > $OUT/$cfg/tc.log 2>&1
  fi
  # the raw traces are large: keep the per-kernel csv files only
  find $OUT/$cfg -name "*.csv" | grep -v -e kernel_stats -e counter_collection | xargs -r rm -f
  echo "$cfg done: $(find $OUT/$cfg -name '*.csv' | wc -l) csv files"
done
# the per-dispatch counter CSVs run to > 100 MB: condense them HERE and bring back only the summaries (gpurun merges at most
# 64 MiB of gpurun_out/)
python scripts/summarize_prof.py $OUT gpurun_out/prof_summary ${PROF_TAG:-r06} > gpurun_out/prof_summary.log 2>&1
tail -3 gpurun_out/prof_summary.log | cut -c1-400
rm -rf $OUT
du -sh gpurun_out/prof_summary | tail -1
