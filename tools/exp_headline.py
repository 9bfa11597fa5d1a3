"""The headline launch (50 M visibilities of the C2 channel in one gridder launch, float32 form),
timed a few times; for same-box A/B runs of build variants (tools/ab_variants.sh)."""
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

G, n, W, P, K = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000, 32, 1, 28
ctx = accel.create_some_context()
q = ctx.create_command_queue()
dev = ctx.device
obs = synth.make_observation(G, n, W, P, device=dev, seed=1)
ip, gp, ap = synth.make_parameters(obs, P, K)
op = grid.GridderTemplate(ctx, ip.fixed, gp.fixed, {'variant': 'mfma', 'arith': 'fp32'}).instantiate(q, ap, ip, gp, n)
shape = op.slots['grid'].shape
op.bind(grid=accel.DeviceArray(ctx, shape, np.complex64),
        weights_grid=accel.DeviceArray(ctx, shape, np.float32, tensor=torch.ones(shape, device=dev)),
        uv=accel.DeviceArray(ctx, (n, 4), np.int16, tensor=obs.uv),
        w_plane=accel.DeviceArray(ctx, (n,), np.int16, tensor=obs.w_plane),
        vis=accel.DeviceArray(ctx, (n, P), np.complex64, tensor=obs.vis))
op.ensure_all_bound()
op.num_vis = n
op._run()
q.finish()
times = []
for _ in range(7):
    t0 = time.perf_counter()
    op._run()
    q.finish()
    times.append(time.perf_counter() - t0)
best, med = min(times), sorted(times)[len(times) // 2]
print('%d visibilities: best %.3f ms = %.2f Gvis/s, median %.3f ms = %.2f Gvis/s' % (
    n, best * 1e3, n / best / 1e9, med * 1e3, n / med / 1e9))
