#!/bin/bash
# Kernel trace of the multi-component CLEAN loop (tools/exp_clean_multi.py): durations of
# cycle_multi_kernel by grid size (= components per launch allowed) and the gaps between
# consecutive launches.  Usage (on the GPU box): bash tools/prof_clean_multi.sh [out dir]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/prof_multi}
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/exp_clean_multi.py 111 133 1000 > $OUT/run.log 2>&1
echo rc=$?
grep -v amdgpu.ids $OUT/run.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
head -6 $f | cut -c1-200
f2=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f2" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mk = [r for r in rows if 'cycle_multi' in r['Kernel_Name']]
print(len(mk), 'multi launches')
g = collections.defaultdict(list)
for r in mk:
    g[int(r['Grid_Size_Z'])].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
for k in sorted(g):
    v = g[k]
    d = sorted(e - s for s, e in v)
    pitch = sorted(v[i + 1][0] - v[i][0] for i in range(len(v) - 1) if v[i + 1][0] - v[i][0] < 100000)
    gaps = sorted(v[i + 1][0] - v[i][1] for i in range(len(v) - 1) if v[i + 1][0] - v[i][0] < 100000)
    print('grid z %2d: %4d launches, duration median %.2f us (p10 %.2f, p90 %.2f); start-to-start %.2f us; gap %.2f us' % (
        k, len(v), d[len(d) // 2] / 1e3, d[len(d) // 10] / 1e3, d[9 * len(d) // 10] / 1e3,
        pitch[len(pitch) // 2] / 1e3, gaps[len(gaps) // 2] / 1e3))
PY
rm -f $OUT/*/*.db
