"""Whole-slice degridder launch timing (50 M visibilities, C2 geometry)."""
import os, sys, time
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

G, n, W, P, K = 4096, 50_000_000, 32, 1, 28
ctx = accel.create_some_context()
q = ctx.create_command_queue()
dev = ctx.device
obs = synth.make_observation(G, n, W, P, device=dev, seed=1)
ip, gp, ap = synth.make_parameters(obs, P, K, degrid=True)
for arith in (sys.argv[1:] or ('fp32', 'split_fp16')):
    dg = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed, {'arith': arith, 'variant': 'mfma'}).instantiate(q, ap, ip, gp, n)
    shape = dg.slots['grid'].shape
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    model = torch.view_as_complex(torch.randn(shape + (2,), generator=gen, device=dev))
    dg.bind(grid=accel.DeviceArray(ctx, shape, np.complex64, tensor=model),
            weights=accel.DeviceArray(ctx, (n, P), np.float32, tensor=torch.ones((n, P), device=dev)),
            uv=accel.DeviceArray(ctx, (n, 4), np.int16, tensor=obs.uv),
            w_plane=accel.DeviceArray(ctx, (n,), np.int16, tensor=obs.w_plane),
            vis=accel.DeviceArray(ctx, (n, P), np.complex64, tensor=obs.vis.clone()))
    dg.ensure_all_bound()
    dg.num_vis = n
    torch.cuda.synchronize()
    dg._run(); q.finish()
    t0 = time.perf_counter()
    for _ in range(3):
        dg._run()
    q.finish()
    dt = (time.perf_counter() - t0) / 3
    print('%-10s %.3f ms  %.2f Gvis/s' % (arith, dt * 1e3, n / dt / 1e9), flush=True)
    del dg
