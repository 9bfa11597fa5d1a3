#!/usr/bin/env python3
"""Print the headline numbers of a bench.py JSON line (file argument)."""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d['roofline']
print('fp32  : %8.1f Mvis/s  %.3f ms/step  frac %.4f  launch %.1f us' % (
    d['value'], d['ms_per_step'], r['frac'], r['avg_launch_us']))
if 'split_fp16' in d:
    s = d['split_fp16']
    print('split : %8.1f Mvis/s  %.3f ms/step  frac %.4f (of fp32 peak %.4f)' % (
        s['value'], s['ms_per_step'], s['roofline']['frac'],
        s['roofline'].get('frac_of_fp32_mfma_peak', 0)))
for k in ('max_norm_error_vs_fp64',):
    if k in d:
        print(k, d[k], d.get('split_fp16', {}).get(k))
for k, v in d.get('secondary', {}).items():
    print('  ', k, v)
if 'major_cycle_loop' in d:
    print('   major_cycle_loop', d['major_cycle_loop'])
if 'cpu_baseline' in d:
    print('   cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('all_cores'))
    print('   cpu', {k: v for k, v in d['cpu_baseline'].items() if k.endswith(('_per_s', '_ms'))})
po = d.get('production_order')
if po:
    print('production order:', po['stream'])
    for k in ('as_delivered', 'store_order', 'store_merged'):
        b = po[k]
        print('   %-13s %8d records  grid %.3f ms %8.1f Mrec/s (frac %.3f)  degrid %.3f ms %8.1f Mrec/s%s' % (
            k, b['records'], b['grid_ms'], b['records_per_s_M'], b['roofline']['frac'], b['degrid_ms'],
            b['degrid_records_per_s_M'],
            '  reorder %.3f ms' % b['reorder_ms_once_per_channel'] if 'reorder_ms_once_per_channel' in b else ''))
sd = d.get('major_cycle_loop', {}).get('store_driven')
if sd:
    print('store-driven C5: %d inputs -> %d stored; preprocess %.0f Mvis/s; reorder %.3f ms; process_channel %.2f ms in store order (%s cycles), %.2f ms in arrival order' % (
        sd['input_visibilities'], sd['records_stored'], sd['preprocess_Mvis_per_s'],
        sd['store_reorder_ms_once_per_channel'], sd['store_order_total_ms'], sd['store_order_minor_cycles'],
        sd['arrival_order_total_ms']))
    print('   stages (store order):', sd['store_order_stage_ms'])
    print('   stages (arrival)    :', sd['arrival_order_stage_ms'])
