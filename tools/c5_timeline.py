"""From a rocprofv3 --kernel-trace CSV of `bench.py --no-secondary --cpu-sample 0`: the device's timeline
over the LAST run of frontend.process_channel (config 5 from the resident store in store order: the
last ~8 ms of kernels before the trace ends): busy time by kernel, and every idle gap above 10 us
with the kernels on either side -- where the host keeps the device waiting.

    python tools/c5_timeline.py DIR [window in ms, default 7.8]"""
import csv
import glob
import sys
import collections

rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
window = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 7.8e6
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
end = ev[-1][1]
ev = [e for e in ev if e[0] >= end - window]
busy = collections.defaultdict(lambda: [0, 0])
covered = 0
last_end = ev[0][0]
gaps = []
for i, (a, b, name) in enumerate(ev):
    short = name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:60]
    busy[short][0] += b - a
    busy[short][1] += 1
    if a > last_end:
        if a - last_end > 10000:
            prev = ev[i - 1][2].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:40]
            gaps.append((a - last_end, (a - ev[0][0]) / 1e6, prev, short))
    covered += max(0, b - max(a, last_end))
    last_end = max(last_end, b)
span = last_end - ev[0][0]
print('window %.3f ms: device busy %.3f ms, idle %.3f ms' % (span / 1e6, covered / 1e6, (span - covered) / 1e6))
for name, (t, n) in sorted(busy.items(), key=lambda kv: -kv[1][0])[:22]:
    print('  %8.1f us  %5d x  %s' % (t / 1e3, n, name))
print('idle gaps above 10 us (%.1f us in all):' % (sum(g[0] for g in gaps) / 1e3))
for g, at, prev, nxt in gaps:
    print('  %7.1f us at %6.3f ms   after %-40s before %s' % (g / 1e3, at, prev, nxt))
