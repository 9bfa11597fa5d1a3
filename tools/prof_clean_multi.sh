cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_multi
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_multi -- python3 tools/exp_clean_multi.py 111 133 1000 > gpurun_out/prof_multi/run.log 2>&1
echo rc=$?
find gpurun_out/prof_multi -name "*kernel_stats.csv" | head
f=$(find gpurun_out/prof_multi -name "*kernel_stats.csv" | head -1)
head -20 $f
f2=$(find gpurun_out/prof_multi -name "*kernel_trace.csv" | head -1)
python3 - "$f2" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
mk = [r for r in rows if 'cycle_multi' in r['Kernel_Name']]
print(len(mk), 'multi launches')
# group by grid size
import collections
g = collections.defaultdict(list)
for r in mk:
    g[(r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
for k, v in g.items():
    d = [e - s for s, e in v]
    gaps = [v[i+1][0] - v[i][1] for i in range(len(v)-1)]
    gaps = [x for x in gaps if x < 100000]
    d.sort(); gaps.sort()
    print(k, len(v), 'dur med %.2f us p10 %.2f p90 %.2f' % (d[len(d)//2]/1e3, d[len(d)//10]/1e3, d[9*len(d)//10]/1e3), 'gap med %.2f us' % (gaps[len(gaps)//2]/1e3 if gaps else -1))
PY
rm -f gpurun_out/prof_multi/*/*.db
