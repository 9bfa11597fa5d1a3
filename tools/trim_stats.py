#!/usr/bin/env python3
"""Condense a rocprofv3 --stats kernel_stats.csv into a short table (kernel names cut at the
first '(' and the at::native noise dropped) for committing under profiles/."""
import csv
import sys


def main(src, dst):
    rows = list(csv.DictReader(open(src)))
    with open(dst, 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats summary (source: {})\n'.format(src))
        f.write('{:<60s} {:>7s} {:>14s} {:>12s} {:>8s} {:>10s} {:>10s}\n'.format(
            'kernel', 'calls', 'total_ns', 'avg_ns', 'pct', 'min_ns', 'max_ns'))
        for r in rows:
            name = r['Name'].replace('void ', '').replace('(anonymous namespace)::', '')
            name = name.split('(')[0][:60]
            f.write('{:<60s} {:>7s} {:>14s} {:>12.1f} {:>8s} {:>10s} {:>10s}\n'.format(
                name, r['Calls'], r['TotalDurationNs'], float(r['AverageNs']), r['Percentage'],
                r['MinNs'], r['MaxNs']))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
