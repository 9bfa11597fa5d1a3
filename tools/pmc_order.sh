# PMC counters + kernel stats of grid_mfma_kernel / degrid_mfma_kernel on one ORDER of the 16.8 M
# order-sweep stream (tools/exp_order_stream.py); per-launch averages -> $OUT/summary_$ORDER.txt
# usage: bash tools/pmc_order.sh loader_blocks|baseline_major|store_order
export TMPDIR=/tmp
ORDER=${1:-loader_blocks}
OUT=gpurun_out/pmc_order_$ORDER
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 tools/exp_order_stream.py $ORDER > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/exp_order_stream.py $ORDER > $OUT/stats.log 2>&1
echo "stats rc=$?"
ORDER=$ORDER python3 - <<'PY' > gpurun_out/pmc_order_$ORDER/summary_$ORDER.txt
import csv, glob, collections, os
order = os.environ['ORDER']
out = 'gpurun_out/pmc_order_' + order
print('# rocprofv3 --pmc (five passes, kernel-trace only) + --stats, window kernels on the %s stream' % order)
print('# (16.8 M-visibility order sweep of bench.py, C2 geometry; tools/pmc_order.sh; average per launch)')
for line in open(out + '/stats.log'):
    if 'Grec/s' in line:
        print('# unprofiled-equivalent timing inside the stats pass: ' + line.strip())
for kern in ('grid_mfma', 'degrid_mfma'):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if ('::' + kern + '_kernel<') not in r['Kernel_Name']:
                continue
            a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    print('[%s_kernel]' % kern)
    for k in sorted(agg):
        print('%-28s %.6g' % (k, agg[k][0] / agg[k][1]))
    for f in glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if ('::' + kern + '_kernel<') in r['Name']:
                print('%-28s calls %s avg_ns %s min_ns %s max_ns %s' % ('kernel_stats', r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs']))
PY
cat $OUT/summary_$ORDER.txt
