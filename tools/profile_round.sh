#!/bin/bash
# Profiles committed under profiles/ (copy the summaries from gpurun_out/prof_<round>/):
#   1. PMC passes over the headline command (HBM traffic and SQ instruction mix of both gridder
#      forms); counters are collected in their own runs with --kernel-trace only, as the pool
#      requires; FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit one);
#   2. kernel-trace stats of the same command, which is handed the traffic measured in (1) so
#      that its JSON line carries roofline.traffic from THIS round's counters;
#   3. kernel-trace stats of the default bench (secondary measurements + major-cycle loop).
# Usage: tools/profile_round.sh r02 [stats-only]
export TMPDIR=/tmp
R=${1:-r02}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
HEAD="python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-secondary"
# 0. the same command unprofiled, on this lease (boxes differ by several per cent: the profiled
#    steady state below is to be compared with THIS line, not with another box's)
$HEAD > $OUT/unprofiled.log 2>&1
grep -h "^{\"metric\"" $OUT/unprofiled.log | tail -1 > $OUT/${R}_bench_unprofiled.json
echo "unprofiled rc=$?"
if [ "$2" != "stats-only" ]; then
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
             "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
    name=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$name -- $HEAD > $OUT/pmc_$name.log 2>&1
    echo "pmc $name rc=$?"
  done
  python3 tools/summarize_profiles.py $OUT $R pmc
fi
# the average duration of grid_mfma_kernel<..., false> (fp32 form) here must agree with
# roofline.avg_launch_us of the JSON line of the same command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $HEAD --traffic-json $OUT/gridder_traffic_fp32.json > $OUT/stats.log 2>&1
echo "stats rc=$?"
grep -h "^{\"metric\"" $OUT/stats.log | tail -1 > $OUT/${R}_bench_profiled.json
# everything else the bench exercises (secondary measurements + the major-cycle loop)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_full -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > $OUT/stats_full.log 2>&1
echo "stats_full rc=$?"
grep -h "^{\"metric\"" $OUT/stats_full.log | tail -1 > $OUT/${R}_bench_full_profiled.json
python3 tools/summarize_profiles.py $OUT $R stats
