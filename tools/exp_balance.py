"""Experiment: does interleaving chunks of the stream over the waves' spans (a static form of load
balancing) shorten the launch?  Each wave's contiguous span is made of `parts` chunks taken from
`parts` regions of the baseline-major stream."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import synth
from katsdpimager_amd import accel, grid

G, n, W, P, K = 4096, 50_000_000, 32, 1, 28
ctx = accel.create_some_context()
q = ctx.create_command_queue()
dev = ctx.device
obs = synth.make_observation(G, n, W, P, device=dev, seed=1)
ip, gp, ap = synth.make_parameters(obs, P, K)
op = grid.GridderTemplate(ctx, ip.fixed, gp.fixed, {'variant': 'mfma', 'arith': 'fp32'}).instantiate(q, ap, ip, gp, n)
shape = op.slots['grid'].shape
gbuf = accel.DeviceArray(ctx, shape, np.complex64)
wg = accel.DeviceArray(ctx, shape, np.float32, tensor=torch.ones(shape, device=dev))

def run(order, label):
    uv, wp, vis = obs.uv, obs.w_plane, obs.vis
    if order is not None:
        uv, wp, vis = uv[order].contiguous(), wp[order].contiguous(), vis[order].contiguous()
    op.bind(grid=gbuf, weights_grid=wg, uv=accel.DeviceArray(ctx, (n, 4), np.int16, tensor=uv),
            w_plane=accel.DeviceArray(ctx, (n,), np.int16, tensor=wp),
            vis=accel.DeviceArray(ctx, (n, P), np.complex64, tensor=vis))
    op.ensure_all_bound()
    op.num_vis = n
    torch.cuda.synchronize()
    op._run(); q.finish()
    t0 = time.perf_counter()
    for _ in range(3):
        op._run()
    q.finish()
    dt = (time.perf_counter() - t0) / 3
    print('%-28s %.3f ms  %.2f Gvis/s' % (label, dt * 1e3, n / dt / 1e9), flush=True)

run(None, 'as generated')
waves = 256 * 12
for parts in (2, 4, 8, 16):
    chunk = (n // (waves * parts)) // 64 * 64
    nc = (n // chunk) // parts * parts
    ids = torch.arange(nc, device=dev)
    # new position p (chunk granularity): wave w = p // parts, part j = p % parts -> source chunk j * (nc / parts) + w
    src = (ids % parts) * (nc // parts) + ids // parts
    order = (src[:, None] * chunk + torch.arange(chunk, device=dev)[None, :]).reshape(-1)
    order = torch.cat([order, torch.arange(nc * chunk, n, device=dev)])
    assert order.numel() == n
    run(order, 'interleaved, %d parts (%d)' % (parts, chunk))
