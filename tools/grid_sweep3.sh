#!/bin/bash
run() { python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-secondary "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['roofline']['avg_launch_us'])"; }
for st in 0 15 30 50 80; do echo "stagger=$st: $(KIMG_GRID_STAGGER=$st run)"; done
for st in 0 30; do echo "stagger=$st noflushend: $(KIMG_GRID_STAGGER=$st KIMG_GRID_DEBUG=1 run)"; done
