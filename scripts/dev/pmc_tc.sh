# usage: pmc_tc.sh <tag>: cache-path counters of the logistic kernel under the current environment
export TMPDIR=/tmp
out=gpurun_out/pmc_tc_$1; rm -rf $out; mkdir -p $out
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum --output-format csv -d $out -o c -- python bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $out/log.txt 2>&1
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "logistic" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$1", {k: round(sum(v)/len(v)/1e6,1) for k,v in sorted(acc.items())})
PY
