#!/bin/bash
# PMC passes over the gridder (counters only; no tracing domains besides kernel-trace)
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$1
shift
mkdir -p $OUT
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-secondary $BENCH_ARGS > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    if 'grid_mfma' not in r['Kernel_Name']:
        continue
    a = agg[r['Counter_Name']]
    a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in sorted(agg.items()):
    print('%-28s per-launch %.4g  (launches %d)' % (k, v / n, n))
PY
done
