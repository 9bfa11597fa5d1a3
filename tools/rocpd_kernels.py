#!/usr/bin/env python3
"""Per-kernel durations out of a rocprofv3 rocpd database (its default output): the mean per
kernel name, or with --sequence the launches in order."""
import argparse
import collections
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--sequence', type=int, default=0, help='print the last N launches in order')
    ap.add_argument('--skip', default='', help='with --sequence: fold runs of kernels whose name has this')
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    rows = db.execute('select name, start, end, grid_x, grid_y, grid_z, workgroup_x '
                      'from kernels order by start').fetchall()
    if args.sequence:
        rows = rows[-args.sequence:]
        t0 = rows[0][1]
        prev_end = t0
        run = None      # (count, start, end) of a folded run
        for name, s, e, gx, gy, gz, wx in rows:
            if args.skip and args.skip in name:
                run = (run[0] + 1, run[1], e) if run else (1, s, e)
                prev_end = e
                continue
            if run:
                print('%10.1f  ... %d x %s, %.1f us in all' % ((run[1] - t0) / 1e3, run[0], args.skip,
                                                             (run[2] - run[1]) / 1e3))
                run = None
            print('%10.1f  gap %8.1f  %9.2f us  grid %5d x %4d x %2d / %4d  %s' % (
                (s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, gx, gy, gz, wx, name[:60]))
            prev_end = max(prev_end, e)
        return
    acc = collections.defaultdict(list)
    for name, s, e, *_ in rows:
        acc[name].append(e - s)
    for name, d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print('%6d x %9.2f us (min %8.2f max %8.2f)  %s' % (
            len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, name[:80]))


if __name__ == '__main__':
    main()
