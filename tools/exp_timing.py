"""Per-wave busy times of the gridder launch (test build with -DKIMG_GRID_TIMING)."""
import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import synth
from katsdpimager_amd import accel, grid, _lib

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
waves = 256 * 12
tim = torch.zeros((waves, 2), dtype=torch.int64, device=dev)
lib = ctypes.CDLL(_lib.lib()._name)
lib.kimg_debug_grid_timing.argtypes = [ctypes.c_void_p]
assert lib.kimg_debug_grid_timing(tim.data_ptr()) == 0

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
    op._run(); q.finish()
    dt = time.perf_counter() - t0
    t = tim.cpu().numpy().astype(np.float64) / 100.0          # us (100 MHz)
    t -= t[:, 0].min()
    busy = t[:, 1] - t[:, 0]
    print('%-24s wall %.3f ms | launch span %.0f us | wave busy: mean %.0f  p5 %.0f  p50 %.0f  p95 %.0f  max %.0f | starts up to %.0f us'
          % (label, dt * 1e3, t[:, 1].max(), busy.mean(), np.percentile(busy, 5), np.percentile(busy, 50),
             np.percentile(busy, 95), busy.max(), t[:, 0].max()), flush=True)
    # by wave slot within the block, and by block
    byslot = busy.reshape(256, 12).mean(axis=0)
    print('   mean busy by wave slot:', ' '.join('%.0f' % x for x in byslot))
    ends = t[:, 1].reshape(256, 12).max(axis=1)
    print('   block end times: min %.0f p50 %.0f max %.0f' % (ends.min(), np.percentile(ends, 50), ends.max()))

run(None, 'as generated')
parts = 16
chunk = (n // (waves * parts)) // 64 * 64
nc = (n // chunk) // parts * parts
ids = torch.arange(nc, device=dev)
src = (ids % parts) * (nc // parts) + ids // parts
order = (src[:, None] * chunk + torch.arange(chunk, device=dev)[None, :]).reshape(-1)
order = torch.cat([order, torch.arange(nc * chunk, n, device=dev)])
run(order, 'interleaved 16 parts')
