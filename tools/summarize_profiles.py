#!/usr/bin/env python3
"""Condense the raw rocprofv3 output of tools/profile_round.sh into small text/JSON files
(written next to the raw data; copy them into profiles/ to commit)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import trim_stats


def main(out, tag):
    stats = glob.glob(os.path.join(out, 'stats', '**', '*kernel_stats.csv'), recursive=True)
    if stats:
        trim_stats.main(stats[0], os.path.join(out, tag + '_bench_kernel_stats.txt'))
    stats = glob.glob(os.path.join(out, 'stats_full', '**', '*kernel_stats.csv'), recursive=True)
    if stats:
        trim_stats.main(stats[0], os.path.join(out, tag + '_bench_full_kernel_stats.txt'))
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'grid_mfma' not in r['Kernel_Name']:
                continue
            a = agg[r['Counter_Name']]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
    per_launch = {k: v / n for k, (v, n) in agg.items()}
    with open(os.path.join(out, tag + '_gridder_pmc.txt'), 'w') as f:
        f.write('# rocprofv3 --pmc, grid_mfma_kernel, average per launch (1 048 576 visibilities)\n')
        for k in sorted(per_launch):
            f.write('{:<28s} {:.6g}\n'.format(k, per_launch[k]))
    if 'FETCH_SIZE' in per_launch and 'WRITE_SIZE' in per_launch:
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
        fetch = per_launch['FETCH_SIZE'] * 1024
        write = per_launch['WRITE_SIZE'] * 1024
        json.dump({'kernel': 'grid_mfma_kernel', 'fetch_bytes_per_launch': fetch,
                   'write_bytes_per_launch': write, 'bytes_per_launch': fetch + write,
                   'note': 'FETCH_SIZE*1024 + WRITE_SIZE*1024 as reported; MI355X_MICROARCH.md '
                           'says FETCH_SIZE under-reports wide (16 B/lane) streaming reads by 2x; '
                           'the gridder loads 8/2/4 B per lane (uncalibrated widths). WRITE_SIZE '
                           'is exact for float atomics.'},
                  open(os.path.join(out, 'gridder_traffic.json'), 'w'), indent=1)
    print(open(os.path.join(out, tag + '_gridder_pmc.txt')).read())


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
