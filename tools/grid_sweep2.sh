#!/bin/bash
for dbg in 0 3 7; do for vb in 524288 1048576 4194304; do
  echo "dbg=$dbg vb=$vb: $(KIMG_GRID_DEBUG=$dbg python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-secondary --vis-block $vb 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['roofline']['avg_launch_us'])")"
done; done
