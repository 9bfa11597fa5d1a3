"""From a rocprofv3 --kernel-trace CSV: per kernel name, calls, average duration and the average gap
to the previous kernel of the same name on the same queue (the price of the boundary between two
dependent launches).    python tools/trace_gaps.py DIR [name-substring ...]"""
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
want = sys.argv[2:]
by = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name']
    if want and not any(w in name for w in want):
        continue
    by[(name[:70], r['Queue_Id'])].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
for (name, queue), v in sorted(by.items(), key=lambda kv: -len(kv[1]))[:30]:
    v.sort()
    dur = sum(b - a for a, b in v) / len(v)
    gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    small = [g for g in gaps if g < 50000]
    print('%-72s q%-3s calls %6d  avg %8.0f ns  gap(avg of <50us) %8.0f ns  pitch %8.0f ns' % (
        name, queue, len(v), dur, sum(small) / max(len(small), 1),
        (sum(small) / max(len(small), 1)) + dur))
