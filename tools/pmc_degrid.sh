export TMPDIR=/tmp
OUT=gpurun_out/pmc_degrid
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 tools/bench_degrid.py --arith ${ARITH:-fp32} > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/pmc_degrid/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'degrid_mfma' not in r['Kernel_Name']:
            continue
        a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k in sorted(agg):
    print('%-28s %.6g' % (k, agg[k][0] / agg[k][1]))
PY
grep "Mvis" $OUT/p1.log
