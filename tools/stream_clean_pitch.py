"""From a rocprofv3 rocpd database of a channel stream: the pitch and duration of the CLEAN cycle
kernels per queue (how much the concurrent chains and the gridding slow a chain).
    python tools/stream_clean_pitch.py results.db"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
print('kernels %d, span %.1f ms' % (len(rows), (rows[-1][2] - rows[0][1]) / 1e6))
by = collections.defaultdict(list)
for n, s, e, q, st in rows:
    if 'cycle_fused' in n:
        by[(q, st, 'batch' if 'batch' in n else 'solo')].append((s, e))
for k, v in sorted(by.items()):
    pitches = sorted(v[i + 1][0] - v[i][0] for i in range(len(v) - 1) if v[i + 1][0] - v[i][0] < 50000)
    durs = sorted(e - s for s, e in v)
    print('queue %s stream %s %-5s: %6d launches, pitch p10 %.2f p50 %.2f p90 %.2f us, duration p50 %.2f us' % (
        k[0], k[1], k[2], len(v), pitches[len(pitches) // 10] / 1e3, pitches[len(pitches) // 2] / 1e3,
        pitches[int(len(pitches) * 0.9)] / 1e3, durs[len(durs) // 2] / 1e3))
