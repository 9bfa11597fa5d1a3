export TMPDIR=/tmp
OUT=gpurun_out/pmc_quick
rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-secondary > $OUT/pmc_$name.log 2>&1
done
python3 tools/summarize_profiles.py $OUT quick pmc
