#!/bin/bash
run() { python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-secondary "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['roofline']['avg_launch_us'])"; }
for nw in 4 8 12; do echo "waves=$nw: $(KIMG_GRID_WAVES=$nw run)   noflush: $(KIMG_GRID_WAVES=$nw KIMG_GRID_DEBUG=3 run)"; done
