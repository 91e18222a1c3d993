# memory-path counters of the bench kernels (TLB, L1 stalls, request latency): development aid
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
CMD="python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --e2e-steps 0"
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum"; do
  i=$((i+1))
  rm -rf /tmp/pm$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pm$i -o c -- $CMD > /tmp/pm$i.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pm$i/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"]
    k = "logistic" if "k_base_logistic" in n else "smooth" if "k_smooth_xgb" in n else None
    if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: "%.4g" % (sum(x) / len(x)) for c, x in v.items()})
PY
done
