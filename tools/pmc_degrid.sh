# PMC counters of degrid_mfma_kernel for ONE launch over the whole 50 M-visibility slice (the way the
# resident-store driver runs it); writes the per-launch averages to $OUT/summary.txt
# usage: ARITH=fp32|split_fp16 bash tools/pmc_degrid.sh
export TMPDIR=/tmp
OUT=gpurun_out/pmc_degrid
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 tools/exp_degrid_slice.py ${ARITH:-fp32} > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - <<'PY' > gpurun_out/pmc_degrid/summary.txt
import csv, glob, collections, os
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/pmc_degrid/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'degrid_mfma' not in r['Kernel_Name']:
            continue
        a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
print('# rocprofv3 --pmc (four passes, kernel-trace only), degrid_mfma_kernel, %s form, average per launch' % os.environ.get('ARITH', 'fp32'))
print('# (ONE launch over the 50 M-visibility W-slice of C2; tools/pmc_degrid.sh)')
for k in sorted(agg):
    print('%-28s %.6g' % (k, agg[k][0] / agg[k][1]))
PY
cat gpurun_out/pmc_degrid/summary.txt; grep "Gvis" $OUT/p1.log
