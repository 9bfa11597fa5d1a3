#!/bin/bash
# timing-only experiments on the gridder (results are wrong when KIMG_GRID_DEBUG != 0)
for dbg in 0 1 2 3; do
  echo "dbg=$dbg: $(KIMG_GRID_DEBUG=$dbg python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['roofline']['avg_launch_us'])")"
done
for b in; do
  echo "blocks=$b: $(KIMG_GRID_BLOCKS=$b python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['roofline']['avg_launch_us'])")"
done
