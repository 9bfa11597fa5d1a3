"""Run the window gridder and degridder (variant 'mfma', float32) a few times over ONE order of
the bench's order-sweep observation (16.8 M visibilities, C2 geometry), for rocprofv3 passes:

    python tools/exp_order_stream.py loader_blocks|baseline_major|store_order [reps]

`store_order` is the loader-block stream after the resident store's once-per-channel re-order
(preprocess.VisibilityReaderDevice.reorder), i.e. what every gridding pass of a channel runs on."""
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
from katsdpimager_amd import accel, grid

order = sys.argv[1] if len(sys.argv) > 1 else 'loader_blocks'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
G, n, W, P, K = 4096, 16 * 1048576, 32, 1, 28
ctx = accel.create_some_context()
q = ctx.create_command_queue()
dev = ctx.device
obs = synth.make_observation(G, n, W, P, device=dev, seed=6)
if order in ('loader_blocks', 'store_order', 'store_merged'):
    obs = synth.order_loader_blocks(obs)['obs']
ip, gp, ap = synth.make_parameters(obs, P, K, degrid=True)
n = obs.n_vis
if order in ('store_order', 'store_merged'):
    from katsdpimager_amd import preprocess
    arrays = dict(uv=accel.DeviceArray(ctx, (n, 4), np.int16, tensor=obs.uv),
                  w_plane=accel.DeviceArray(ctx, (n,), np.int16, tensor=obs.w_plane),
                  weights=accel.DeviceArray(ctx, (n, P), np.float32, tensor=obs.weights),
                  vis=accel.DeviceArray(ctx, (n, P), np.complex64, tensor=obs.vis))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, n = preprocess.reorder_device_arrays(q, P, n, arrays, K, obs.oversample, W,
                                              order == 'store_merged')
    q.finish()
    print('%s: re-order of %d records -> %d in %.3f ms (first call: includes code loading)' % (
        order, obs.n_vis, n, (time.perf_counter() - t0) * 1e3), flush=True)
    obs = synth._copy_with(obs, out['uv'].tensor[:n], out['w_plane'].tensor[:n], out['vis'].tensor[:n],
                           out['weights'].tensor[:n])
g = grid.GridderTemplate(ctx, ip.fixed, gp.fixed, {'variant': 'mfma'}).instantiate(q, ap, ip, gp, n)
d = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed, {'variant': 'mfma'}).instantiate(q, ap, ip, gp, n)
shape = g.slots['grid'].shape
gen = torch.Generator(device=dev)
gen.manual_seed(3)
model = torch.view_as_complex(torch.randn(shape + (2,), generator=gen, device=dev))
gbuf = accel.DeviceArray(ctx, shape, np.complex64, tensor=model)
common = dict(grid=gbuf, uv=accel.DeviceArray(ctx, (n, 4), np.int16, tensor=obs.uv),
              w_plane=accel.DeviceArray(ctx, (n,), np.int16, tensor=obs.w_plane),
              vis=accel.DeviceArray(ctx, (n, P), np.complex64, tensor=obs.vis.clone()))
g.bind(weights_grid=accel.DeviceArray(ctx, shape, np.float32, tensor=torch.ones(shape, device=dev)), **common)
d.bind(weights=accel.DeviceArray(ctx, (n, P), np.float32, tensor=torch.ones((n, P), device=dev)), **common)
for op in (g, d):
    op.ensure_all_bound()
    op.num_vis = n
torch.cuda.synchronize()
for name, op in (('grid', g), ('degrid', d)):
    op._run()
    q.finish()
    t0 = time.perf_counter()
    for _ in range(reps):
        op._run()
    q.finish()
    dt = (time.perf_counter() - t0) / reps
    print('%s %-6s records %d  %.3f ms  %.2f Grec/s' % (order, name, n, dt * 1e3, n / dt / 1e9), flush=True)
