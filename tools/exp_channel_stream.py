"""A stream of channels on one GPU (frontend.process_channel_stream), `workers` of them in flight:
wall time per channel with the channels in step (all share the device while gridding, then CLEAN
together in one batch) against taking turns at the throughput-bound stages (CleanBatcher phased).

    python tools/exp_channel_stream.py [vis] [channels] [workers ...]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import synth
from katsdpimager_amd import _lib as _kl
if os.environ.get("KIMG_VARIANT_LIB"):
    _kl.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_variants",
                                "libkimg_%s.so" % os.environ["KIMG_VARIANT_LIB"])
from katsdpimager_amd import accel, clean, frontend, imaging, parameters, preprocess, weight

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
channels = int(sys.argv[2]) if len(sys.argv) > 2 else 12
workers_list = [int(a) for a in sys.argv[3:]] or [2, 4, 6]
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
for _ in range(max(workers_list)):
    qi = ctx.create_command_queue()
    imi = template2.instantiate(qi, ipd, gpd, block, 0, 2)
    imi.ensure_all_bound()
    jobs.append(dict(reader=reader, rel_channel=0, imager=imi, image_p=ipd, grid_p=gpd,
                     clean_p=cp2, weight_type=wparm.weight_type, vis_block=block, major=2, degrid=True))


def make_job(channel, worker):
    return jobs[worker]


frontend.process_channels(jobs, workers=len(jobs))          # graph capture per imager
t0 = time.perf_counter()
frontend.process_channel_stream(make_job, range(4), workers=1)
torch.cuda.synchronize()
print('one channel at a time: %.2f ms per channel' % ((time.perf_counter() - t0) * 1e3 / 4))
_make = clean.CleanBatcher
if os.environ.get('KIMG_WINDOW_CUS'):      # (experiment: CUs the window kernels fill while channels share the device)
    frontend.WINDOW_CUS_SHARED = int(os.environ['KIMG_WINDOW_CUS'])
for workers in workers_list:
    for name, kw in (('in step', None), ('turns', dict(overlap=True)), ('turns, two at a time', dict(phase_permits=2))):
        if kw is not None:
            clean.CleanBatcher = lambda parties, phased=False, kw=kw: _make(parties, phased=True, **kw)
        else:
            clean.CleanBatcher = _make
        best = None
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = frontend.process_channel_stream(make_job, range(channels), workers=workers,
                                                  stagger=kw is not None)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        sizes = [b[0] for b in frontend.process_channel_stream.last_batches]
        print('workers=%d %-28s %7.2f ms = %.2f ms per channel; shared launches of %s channels; minor %s' % (
            workers, name, best * 1e3, best * 1e3 / channels,
            {c: sizes.count(c) for c in sorted(set(sizes))}, sorted({int(r['minor']) for r in res})))
clean.CleanBatcher = _make
