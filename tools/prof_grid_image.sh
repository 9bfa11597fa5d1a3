#!/bin/bash
# kernel durations of tools/exp_grid_image.py (run on the GPU box from the repo root)
set -e
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $ROOT/gpurun_out/prof_g2i -o g2i -- python3 $ROOT/tools/exp_grid_image.py "$@" > $ROOT/gpurun_out/exp_g2i_prof.log 2>&1
python3 $ROOT/tools/rocpd_kernels.py $ROOT/gpurun_out/prof_g2i/g2i_results.db | grep -i "g2i\|i2g" 
