# usage: pmc_sm.sh <outdir> <impl>: SQ counter passes over the smoother alone (scripts/dev/smoother_impls.py)
export TMPDIR=/tmp
out=$1; impl=$2; mkdir -p $out
IMPLS=$impl rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $out/prof_sq --output-format csv -- python scripts/dev/smoother_impls.py > $out/log.txt 2>&1
IMPLS=$impl rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES -d $out/prof_sq2 --output-format csv -- python scripts/dev/smoother_impls.py >> $out/log.txt 2>&1
python scripts/summarize_prof.py $out $out x >/dev/null 2>&1
python - <<PY
import json
d=json.load(open("$out/x_pmc.json"))
for k,v in d.items():
    if "smooth" in k:
        print(k[:50], {c: round(x["avg_per_launch"]/1e6,1) for c,x in v.items() if isinstance(x,dict)})
PY
