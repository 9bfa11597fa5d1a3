"""Per-wave begin / end times of ONE gridder launch over the re-ordered store (6.3 M records of the
order sweep), and the launch time against the number of records (prefixes of the same stream): where a
short launch loses against a long one.  Needs the timing build:

    python tools/build_variant.py timing grid_mfma.hip -DKIMG_GRID_TIMING
    KIMG_VARIANT_LIB=timing python tools/exp_timing_store.py"""
import ctypes
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
from katsdpimager_amd import accel, grid, preprocess

G, n, W, P, K = 4096, 16 * 1048576, 32, 1, 28
ctx = accel.create_some_context()
q = ctx.create_command_queue()
dev = ctx.device
obs = synth.order_loader_blocks(synth.make_observation(G, n, W, P, device=dev, seed=6))['obs']
ip, gp, ap = synth.make_parameters(obs, P, K, degrid=True)
n = obs.n_vis
arrays = dict(uv=accel.DeviceArray(ctx, (n, 4), np.int16, tensor=obs.uv),
              w_plane=accel.DeviceArray(ctx, (n,), np.int16, tensor=obs.w_plane),
              weights=accel.DeviceArray(ctx, (n, P), np.float32, tensor=obs.weights),
              vis=accel.DeviceArray(ctx, (n, P), np.complex64, tensor=obs.vis))
out, n = preprocess.reorder_device_arrays(q, P, n, arrays, K, obs.oversample, W, False)
q.finish()
g = grid.GridderTemplate(ctx, ip.fixed, gp.fixed, {'variant': 'mfma'}).instantiate(q, ap, ip, gp, n)
shape = g.slots['grid'].shape
g.bind(grid=accel.DeviceArray(ctx, shape, np.complex64),
       weights_grid=accel.DeviceArray(ctx, shape, np.float32, tensor=torch.ones(shape, device=dev)),
       uv=accel.DeviceArray(ctx, (n, 4), np.int16, tensor=out['uv'].tensor[:n]),
       w_plane=accel.DeviceArray(ctx, (n,), np.int16, tensor=out['w_plane'].tensor[:n]),
       vis=accel.DeviceArray(ctx, (n, P), np.complex64, tensor=out['vis'].tensor[:n].clone()))
g.ensure_all_bound()
waves = 256 * 12
tim = torch.zeros((waves, 2), dtype=torch.int64, device=dev)
lib = ctypes.CDLL(_kl.lib()._name)
have_timing = hasattr(lib, 'kimg_debug_grid_timing')
if have_timing:
    lib.kimg_debug_grid_timing.argtypes = [ctypes.c_void_p]
    assert lib.kimg_debug_grid_timing(tim.data_ptr()) == 0
for count in (n // 16, n // 8, n // 4, n // 2, n):
    g.num_vis = count
    g._run()
    q.finish()
    t0 = time.perf_counter()
    for _ in range(10):
        g._run()
    q.finish()
    dt = (time.perf_counter() - t0) / 10
    line = 'records %8d  %.3f ms  %.2f Grec/s' % (count, dt * 1e3, count / dt / 1e9)
    if have_timing:
        t = tim.cpu().numpy().astype(np.float64) / 100.0          # us (100 MHz)
        t = t[t[:, 1] > 0]
        t -= t[:, 0].min()
        ends = np.sort(t[:, 1])
        line += ' | span %.0f us, waves begin by %.0f us, ends: p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f' % (
            ends[-1], t[:, 0].max(), *np.percentile(ends, [10, 50, 90, 99]), ends[-1])
    print(line, flush=True)
