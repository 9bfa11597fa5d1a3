"""Stand-alone timing of the device preprocessor (used under rocprofv3 for per-kernel times).
python tools/bench_preprocess.py [--vis N] [--pols P] [--in-pols Q] [--w-slices S] [--reps R]"""
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
    ap.add_argument('--pols', type=int, default=1)
    ap.add_argument('--in-pols', type=int, default=None)
    ap.add_argument('--w-slices', type=int, default=1)
    ap.add_argument('--w-planes', type=int, default=32)
    ap.add_argument('--vis-block', type=int, default=1 << 20)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--feed-angles', action='store_true')
    args = ap.parse_args()
    import torch
    import synth
    from katsdpimager_amd import accel, preprocess
    P = args.pols
    Q = args.in_pols or P
    ctx = accel.create_some_context()
    q = ctx.create_command_queue()
    obs = synth.make_observation(args.pixels, args.vis, args.w_planes, 1, device=ctx.device,
                                 w_slices=args.w_slices)
    ip, gp, _ = synth.make_parameters(obs, P, 28)
    n = obs.n_vis
    gen = torch.Generator(device=ctx.device)
    gen.manual_seed(1)
    vis = torch.complex(torch.rand((1, n, Q), generator=gen, device=ctx.device) - 0.5,
                        torch.rand((1, n, Q), generator=gen, device=ctx.device) - 0.5)
    wts = torch.rand((1, n, Q), generator=gen, device=ctx.device) + 0.5
    d_uvw = accel.DeviceArray(ctx, (n, 3), np.float32, tensor=obs.uvw)
    d_wts = accel.DeviceArray(ctx, (1, n, Q), np.float32, tensor=wts)
    d_vis = accel.DeviceArray(ctx, (1, n, Q), np.complex64, tensor=vis)
    fa = None
    circ = None
    rs = np.random.RandomState(0)
    if args.feed_angles:
        fa = accel.DeviceArray(ctx, (n,), np.float32,
                               tensor=torch.rand((n,), generator=gen, device=ctx.device))
        stokes = (rs.standard_normal((P, 4)) + 1j * rs.standard_normal((P, 4))).astype(np.complex64)
        circ = (rs.standard_normal((4, Q)) + 1j * rs.standard_normal((4, Q))).astype(np.complex64)
    else:
        stokes = (rs.standard_normal((P, Q)) + 1j * rs.standard_normal((P, Q))).astype(np.complex64)
    torch.cuda.synchronize()
    for rep in range(args.reps):
        coll = preprocess.VisibilityCollectorDevice(q, [ip], [gp], args.vis_block)
        q.finish()
        t0 = time.perf_counter()
        coll.add(d_uvw, d_wts, d_vis, fa, fa, stokes, circ)
        q.finish()
        dt = time.perf_counter() - t0
        in_bytes = n * (12 + 12 * Q)
        out_bytes = coll.num_output * (10 + 12 * P)
        print('rep %d: %.2f ms  %.1f Minput-vis/s  kept %.3f  compulsory traffic %.1f GB/s' % (
            rep, dt * 1e3, n / dt / 1e6, coll.num_output / coll.num_input,
            (in_bytes + out_bytes) / dt / 1e9), flush=True)
        del coll


if __name__ == '__main__':
    main()
