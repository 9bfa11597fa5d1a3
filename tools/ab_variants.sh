#!/bin/bash
# same-box A/B: the regular libkimg.so and every build_variants/libkimg_NAME.so named on the command
# line through one experiment script.   tools/ab_variants.sh "python3 tools/exp_....py args" NAME ...
CMD=$1
shift
echo "== base"; $CMD 2>&1 | grep -v "amdgpu.ids"
for v in "$@"; do
  echo "== $v"; KIMG_VARIANT_LIB=$v $CMD 2>&1 | grep -v "amdgpu.ids"
done
echo "== base again"; $CMD 2>&1 | grep -v "amdgpu.ids"
