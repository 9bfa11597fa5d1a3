#!/usr/bin/env python3
"""Condense the raw rocprofv3 output of tools/profile_round.sh into small text/JSON files
(written next to the raw data; copy them into profiles/ to commit)."""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import trim_stats


def gridder_form(kernel_name):
    """'fp32' / 'split_fp16' from the last template argument (F16) of grid_mfma_kernel<...>."""
    m = re.search(r'grid_mfma_kernel<([^>]*)>', kernel_name)
    if not m:
        return None
    return 'split_fp16' if m.group(1).split(',')[-1].strip() in ('true', '1') else 'fp32'


def pmc(out, tag):
    agg = {'fp32': collections.defaultdict(lambda: [0.0, 0]),
           'split_fp16': collections.defaultdict(lambda: [0.0, 0])}
    for f in glob.glob(os.path.join(out, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            form = gridder_form(r['Kernel_Name'])
            if form is None:
                continue
            a = agg[form][r['Counter_Name']]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
    for form, counters in agg.items():
        per_launch = {k: v / n for k, (v, n) in counters.items()}
        if not per_launch:
            continue
        path = os.path.join(out, '{}_gridder_pmc_{}.txt'.format(tag, form))
        with open(path, 'w') as f:
            f.write('# rocprofv3 --pmc, grid_mfma_kernel ({} form), average per launch of\n'
                    '# `bench.py --no-secondary` (one launch = the whole 50 M-visibility W-slice)\n'
                    .format(form))
            for k in sorted(per_launch):
                f.write('{:<28s} {:.6g}\n'.format(k, per_launch[k]))
        print(open(path).read())
        if 'FETCH_SIZE' in per_launch and 'WRITE_SIZE' in per_launch:
            # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
            fetch = per_launch['FETCH_SIZE'] * 1024
            write = per_launch['WRITE_SIZE'] * 1024
            json.dump({'kernel': 'grid_mfma_kernel', 'arith': form, 'kernel_width': 28,
                       'polarizations': 1, 'vis_per_launch': 50000000,
                       'fetch_bytes_per_launch': fetch, 'write_bytes_per_launch': write,
                       'bytes_per_launch': fetch + write,
                       'note': 'FETCH_SIZE*1024 + WRITE_SIZE*1024 as reported, separate --pmc '
                               'passes; MI355X_MICROARCH.md: FETCH_SIZE under-reports 16 B/lane '
                               'streaming reads by 2x and is uncalibrated for other widths (the '
                               'gridder loads 8 / 2 / 8 / 4 B per lane); WRITE_SIZE is exact for '
                               'float atomics.'},
                      open(os.path.join(out, 'gridder_traffic_{}.json'.format(form)), 'w'), indent=1)


def stats(out, tag):
    for sub, name in (('stats', '_bench_kernel_stats.txt'), ('stats_full', '_bench_full_kernel_stats.txt')):
        found = glob.glob(os.path.join(out, sub, '**', '*kernel_stats.csv'), recursive=True)
        if found:
            trim_stats.main(found[0], os.path.join(out, tag + name))
            print(open(os.path.join(out, tag + name)).read()[:3000])


if __name__ == '__main__':
    what = sys.argv[3] if len(sys.argv) > 3 else 'all'
    if what in ('pmc', 'all'):
        pmc(sys.argv[1], sys.argv[2])
    if what in ('stats', 'all'):
        stats(sys.argv[1], sys.argv[2])
