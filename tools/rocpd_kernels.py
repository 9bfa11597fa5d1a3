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
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    rows = db.execute('select name, start, end, grid_x, grid_y, grid_z, workgroup_x '
                      'from kernels order by start').fetchall()
    if args.sequence:
        for name, s, e, gx, gy, gz, wx in rows[-args.sequence:]:
            print('%9.2f us  grid %5d x %4d x %2d / %4d  %s' % ((e - s) / 1e3, gx, gy, gz, wx, name[:70]))
        return
    acc = collections.defaultdict(list)
    for name, s, e, *_ in rows:
        acc[name].append(e - s)
    for name, d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print('%6d x %9.2f us (min %8.2f max %8.2f)  %s' % (
            len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, name[:80]))


if __name__ == '__main__':
    main()
