"""Stand-alone degridding loop (used under rocprofv3 for counters of degrid_mfma_kernel).
python tools/bench_degrid.py [--vis N] [--w-planes W] [--kernel-width K] [--pols P]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--vis', type=int, default=50_000_000)
    ap.add_argument('--pixels', type=int, default=4096)
    ap.add_argument('--w-planes', type=int, default=32)
    ap.add_argument('--kernel-width', type=int, default=28)
    ap.add_argument('--pols', type=int, default=1)
    ap.add_argument('--chunks', type=int, default=8)
    ap.add_argument('--arith', default='fp32', choices=['fp32', 'split_fp16'])
    args = ap.parse_args()
    import torch
    import synth
    from katsdpimager_amd import accel, grid
    ctx = accel.create_some_context()
    q = ctx.create_command_queue()
    P = args.pols
    obs = synth.make_observation(args.pixels, args.vis, args.w_planes, P, device=ctx.device)
    ip, gp, ap_ = synth.make_parameters(obs, P, args.kernel_width, degrid=True)
    vb = 1 << 20
    dg = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed, {'arith': args.arith}).instantiate(
        q, ap_, ip, gp, vb)
    Gg = dg.slots['grid'].shape[1]
    gen = torch.Generator(device=ctx.device)
    gen.manual_seed(1)
    G = torch.complex(torch.rand((P, Gg, Gg), generator=gen, device=ctx.device) - 0.5,
                      torch.rand((P, Gg, Gg), generator=gen, device=ctx.device) - 0.5)
    dg.bind(grid=accel.DeviceArray(ctx, (P, Gg, Gg), np.complex64, tensor=G),
            weights=accel.DeviceArray(ctx, (vb, P), np.float32,
                                      tensor=torch.ones((vb, P), device=ctx.device)))
    dg.ensure_all_bound()
    torch.cuda.synchronize()

    def run():
        for c in range(args.chunks):
            sl = slice(c * vb, (c + 1) * vb)
            dg.bind(uv=accel.DeviceArray(ctx, (vb, 4), np.int16, tensor=obs.uv[sl]),
                    w_plane=accel.DeviceArray(ctx, (vb,), np.int16, tensor=obs.w_plane[sl]),
                    vis=accel.DeviceArray(ctx, (vb, P), np.complex64, tensor=obs.vis[sl]))
            dg.num_vis = vb
            dg._run()
    run()
    q.finish()
    t0 = time.perf_counter()
    run()
    q.finish()
    dt = time.perf_counter() - t0
    print('%.1f Mvis/s, %.1f us per 1M-visibility launch' % (args.chunks * vb / dt / 1e6,
                                                            dt / args.chunks * 1e6))


if __name__ == '__main__':
    main()
