"""Four channels of one GPU in flight (frontend.process_channels, batched CLEAN): wall time and the
host-side timeline of the worker threads (trace.record_timeline).

    python tools/exp_four_channels.py [vis] [workers ...]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import synth
from katsdpimager_amd import _lib as _kl
if os.environ.get('KIMG_VARIANT_LIB'):
    _kl.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build_variants',
                                'libkimg_%s.so' % os.environ['KIMG_VARIANT_LIB'])
from katsdpimager_amd import accel, frontend, imaging, parameters, preprocess, trace, weight

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
workers_list = [int(a) for a in sys.argv[2:]] or [1, 4]
G, W, P, K = 4096, 32, 1, 28
ctx = accel.create_some_context()
q = ctx.create_command_queue()
dev = ctx.device
obs = synth.make_observation(G, n, W, P, device=dev, cover=0.30, seed=2)
ipd, gpd, apd = synth.make_parameters(obs, P, K, degrid=True)
synth.add_point_sources(obs, 200, seed=4, noise=0.01)
raw_vis = obs.raw_vis
d_uvw = accel.DeviceArray(ctx, (n, 3), np.float32, tensor=obs.uvw)
d_wts = accel.DeviceArray(ctx, (1, n, P), np.float32, tensor=obs.weights[None].contiguous())
d_vis = accel.DeviceArray(ctx, (1, n, P), np.complex64, tensor=raw_vis[None].contiguous())
torch.cuda.synchronize()
coll = preprocess.VisibilityCollectorDevice(q, [ipd], [gpd], 1 << 20)
coll.add(d_uvw, d_wts, d_vis, None, None, np.identity(P, np.complex64), None)
coll.close()
reader = coll.reader()
print('store: %d of %d records' % (coll.num_stored, coll.num_input))
block = max(reader.len(0, s) for s in range(reader.num_w_slices(0)))
cp2 = parameters.CleanParameters(1000, 0.1, 1.0, 0.0, 0, 0.01, 0.5, 0.02)
wparm = parameters.WeightParameters(weight.WeightType.ROBUST, 0.0)
template2 = imaging.ImagingTemplate(ctx, apd, ipd.fixed, wparm, gpd.fixed, cp2)
jobs = []
for _ in range(4):
    qi = ctx.create_command_queue()
    imi = template2.instantiate(qi, ipd, gpd, block, 0, 2)
    imi.ensure_all_bound()
    jobs.append(dict(reader=reader, rel_channel=0, imager=imi, image_p=ipd, grid_p=gpd,
                     clean_p=cp2, weight_type=wparm.weight_type, vis_block=block, major=2, degrid=True))
frontend.process_channels(jobs, workers=4)
from katsdpimager_amd import clean as _clean
_orig = _clean.enqueue_cycles_batch
_calls = []


def _timed(cleans, patches, thresholds, cycles, queue=None):
    t0 = time.perf_counter()
    qq = _orig(cleans, patches, thresholds, cycles, queue)
    t1 = time.perf_counter()
    qq.finish()
    _calls.append((len(cleans), (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
    return qq


_clean.enqueue_cycles_batch = _timed
for workers in workers_list:
    for rep in range(2):
        entries = []
        trace.record_timeline(entries)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = frontend.process_channels(jobs, workers=workers)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        trace.record_timeline(None)
    print('workers=%d: %.2f ms  batches %s  minor %s' % (
        workers, dt * 1e3, frontend.process_channel_stream.last_batches, [int(r['minor']) for r in res]))
    print('   batch calls (channels, enqueue ms, then wait ms):', [(c, round(a, 3), round(b, 3)) for c, a, b in _calls])
    _calls.clear()
    for th, name, a, b in sorted(entries, key=lambda e: (e[0], e[2])):
        print('   %-16s %-14s %8.3f -> %8.3f  (%.3f ms)' % (th, name, (a - t0) * 1e3, (b - t0) * 1e3, (b - a) * 1e3))
