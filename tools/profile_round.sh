#!/bin/bash
# Profiles committed under profiles/: kernel-trace stats of the default bench, and PMC passes
# (HBM traffic and SQ instruction mix) over the gridder.  Counters are collected in their own
# runs (kernel-trace only), as the pool requires.
export TMPDIR=/tmp
R=${1:-r01}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
# headline kernels only (the per-chunk launches bench.py times): the average duration of
# grid_mfma_kernel here must agree with roofline.avg_launch_us of the same command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-secondary > $OUT/stats.log 2>&1
echo "stats rc=$?"
grep -h "^{\"metric\"" $OUT/stats.log | tail -1 > $OUT/${R}_bench_profiled.json
# everything else the bench exercises (secondary measurements + the major-cycle loop)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_full -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 --major-loop > $OUT/stats_full.log 2>&1
echo "stats_full rc=$?"
if [ "$2" = "stats-only" ]; then python tools/summarize_profiles.py $OUT $R; exit 0; fi
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$name -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-secondary > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
done
python tools/summarize_profiles.py $OUT $R
