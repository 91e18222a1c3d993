cd "${GRAFT_REPO_ROOT:-.}"
for hb in 0 2500 1250 1000 626 312; do
  GNX_HOST_BATCH=$hb python bench.py --cpu-seconds 0 --steps 3 --warmup 1 --e2e-steps 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['e2e']; print('HOST_BATCH=$hb', 'int8 %.0f hap/s %.1f ms' % (e['int8']['haplotypes_per_s'], e['int8']['ms']), 'packed %.0f hap/s %.1f ms %.1f GB/s' % (e['packed2bit']['haplotypes_per_s'], e['packed2bit']['ms'], e['packed2bit']['x_GBps']))"
done
