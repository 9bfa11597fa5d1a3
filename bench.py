#!/usr/bin/env python3
"""Benchmark of the imaging hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): one spectral channel per GPU, 4096^2 image, 32
W-planes, ~50 M synthetic visibilities (tools/synth.py: 64-antenna array, earth-rotation tracks,
baseline-major order, uncompressed), float32, Stokes I, kernel width 28, oversample 8, every input
resident in HBM.  One "step" = clear the grid + grid the channel's W-slice the way the product's
resident-store driver does it (frontend.make_dirty over VisibilityReaderDevice.iter_slice_device
with one block per slice, i.e. ONE gridder launch per slice; the hot loop #1 of the reference's
frontend.make_dirty, frontend.py:126-139).  The same visibilities streamed in the reference's
--vis-block chunks of 1 048 576 (frontend.py:357), one launch per chunk, are measured too
(`chunked`).

`value` / `roofline` are for the gridder's float32 arithmetic (KIMG_ARITH_FP32,
v_mfma_f32_32x32x2_f32: every product and sum in float32 like the reference's kernel).  The
opt-in fp16 hi/lo form (KIMG_ARITH_SPLIT_FP16) is a second block, `split_fp16`, scored against
the F16 matrix pipe it runs on.  Both blocks carry their max-norm error against a float64
evaluation of one 1 M-visibility chunk.

With N > 1 (one process per GPU, torchrun) every rank images its own channel of an 8-channel band
(rank 0 always the top channel, the one the N = 1 run images; same geometry for every N): the
path shards by channel with no data-path collective (weak scaling); rank 0 broadcasts the
channel-independent baseline table over RCCL once at start-up.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
F16_MFMA_PEAK_TFLOPS = 2516.6     # same table: BF16/F16 dense, v_mfma_f32_32x32x16_f16
HBM_PEAK_GBS = 8000.0
BAND_CHANNELS = 8                 # BASELINE config 3: 8 spectral-line channels
BAND_SPREAD = 0.03


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=5)
    p.add_argument('--warmup', type=int, default=2)
    p.add_argument('--vis', type=int, default=50_000_000)
    p.add_argument('--pixels', type=int, default=4096)
    p.add_argument('--w-planes', type=int, default=32)
    p.add_argument('--kernel-width', type=int, default=28)
    p.add_argument('--polarizations', type=int, default=1)
    p.add_argument('--vis-block', type=int, default=1048576,
                   help='chunk size of the `chunked` measurements (the reference\'s --vis-block)')
    p.add_argument('--variant', default='auto', choices=['auto', 'generic', 'mfma'])
    p.add_argument('--clean-cycles', type=int, default=1000)
    p.add_argument('--cpu-sample', type=int, default=3_200_000,
                   help='visibilities per timed pass of the CPU baseline (0 disables it)')
    p.add_argument('--cpu-passes', type=int, default=5)
    p.add_argument('--no-secondary', action='store_true',
                   help='only the headline gridder measurement (and the CPU baseline)')
    p.add_argument('--no-major-loop', action='store_true',
                   help='skip the major-cycle loop (BASELINE config 5)')
    p.add_argument('--extras', action='store_true',
                   help='also: preprocessing + store-driven driver, several channels in flight, '
                        'host-chunk (PCIe-inclusive) gridding')
    p.add_argument('--traffic-json',
                   default=os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles',
                                        'gridder_traffic_fp32.json'),
                   help='HBM bytes per launch from a rocprofv3 --pmc pass of this same command '
                        '(tools/profile_round.sh; default: the committed pass of the current '
                        'kernel); included as roofline.traffic only if its recorded configuration '
                        'matches this run')
    p.add_argument('--roctx', action='store_true',
                   help='named roctx ranges around the stages of the major-cycle loop '
                        '(rocprofv3 --marker-trace)')
    p.add_argument('--force-dist', action='store_true',
                   help='take the distributed code path (RCCL process group, broadcast, '
                        'gathers, barriers) also with ONE rank: how that path is exercised on a '
                        'one-GPU box')
    p.add_argument('--rehearse', action='store_true',
                   help='CPU rehearsal of the multi-process plumbing (rendezvous, broadcast, channel '
                        'assignment, barriers, reductions, the JSON line) with no device work; for '
                        'the gloo tests')
    return p.parse_args()


def rank_channel(rank, world):
    """Channel of the 8-channel band imaged by `rank`: rank 0 always takes the top channel (so the
    N = 1 run is one of the channels of every N > 1 run); the others spread over the band."""
    stride = max(BAND_CHANNELS // max(world, 1), 1)
    return (BAND_CHANNELS - 1 - rank * stride) % BAND_CHANNELS


def channel_scale(channel):
    """Relative frequency of a band channel, normalised to the top one: uvw in wavelengths scale
    with it, so every channel's footprint stays inside the grid sized for the top channel."""
    from katsdpimager_amd import parallel
    return parallel.channel_frequency_scale(channel, BAND_CHANNELS, BAND_SPREAD) / (1 + BAND_SPREAD)


class Timer:
    """K steps bracketed by barrier + synchronize; per-step HIP events on the operator's stream."""

    def __init__(self, q, barrier, dev):
        self.q, self.barrier, self.dev = q, barrier, dev

    def run(self, step, steps, warmup, reduce=True):
        import torch
        from katsdpimager_amd import parallel
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        for _ in range(warmup):
            step(None, None)
        torch.cuda.synchronize()
        self.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(ev0[k], ev1[k])
        torch.cuda.synchronize()
        self.local_sec_per_step = (time.perf_counter() - t0) / steps      # this rank's own clock
        self.barrier()
        t1 = time.perf_counter()
        elapsed = parallel.max_over_ranks(t1 - t0, self.dev) if reduce else t1 - t0
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
        return elapsed / steps, kern_ms


def grid_roofline(arith, K, P, n_vis, launches, kern_ms, variant):
    """Roofline block of the gridder kernel: algorithmic flops of one launch / its average
    duration (HIP events on the launch stream), against the peak of the matrix pipe it uses."""
    flop_per_vis = 8.0 * K * K * P               # one complex MAC per tap and polarization
    bytes_per_vis = 8 + 2 + 8 * P + 4 * P        # uv + w_plane + vis + weight gather
    launch_us = kern_ms * 1e3 / launches
    secs = kern_ms * 1e-3
    achieved = flop_per_vis * n_vis / secs / 1e12
    blocks = 1 if K <= 32 else 4                 # 2 x 2 tap blocks for wide kernels
    split = arith == 'split_fp16' and variant != 'generic' and K <= 64
    mfma = variant != 'generic' and K <= 64
    # executed: a 32x32 window per visibility (and tap block); the split form spends a whole
    # 32x32x16 instruction (6 of 16 k-slots used) on two visibilities
    executed_per_vis = blocks * P * (2 * 32 * 32 * 16 * 2 / 2 if split else 2 * 32 * 32 * 2 * 2)
    peak = F16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
    executed = executed_per_vis * n_vis / secs / 1e12
    out = {
        'kernel': 'grid_mfma_kernel' if mfma else 'grid_generic_kernel',
        'instruction': ('v_mfma_f32_32x32x16_f16 (fp16 hi/lo operand pairs, 2 visibilities each)'
                        if split else 'v_mfma_f32_32x32x2_f32' if mfma else 'per-tap float atomics'),
        'bound': 'mfma', 'achieved': round(achieved, 3), 'peak': peak, 'unit': 'TFLOP/s',
        'frac': round(achieved / peak, 4), 'traffic': None,
        'flop_per_vis': flop_per_vis, 'vis_per_launch': int(round(n_vis / launches)),
        'launches_per_step': launches, 'avg_launch_us': round(launch_us, 2),
        'executed_tflops': round(executed, 2), 'executed_frac': round(executed / peak, 4),
        'hbm_algorithmic_GBps': round(bytes_per_vis * n_vis / secs / 1e9, 1),
        'hbm_frac_of_8TBps': round(bytes_per_vis * n_vis / secs / 1e9 / HBM_PEAK_GBS, 4),
        # the reference bench's own figure of merit: grid-point additions per second
        # (tests/imager_bench.py:204-208), N K^2 P / t
        'GGAPS': round(n_vis * K * K * P / secs / 1e9, 1),
    }
    if split:
        # the split form's loop is bound by VALU issue (operand re-join / split / permute), not by
        # the F16 pipe: also state the algorithmic fp32 work against the fp32 matrix peak, i.e.
        # against the best the exact instruction could do
        out['frac_of_fp32_mfma_peak'] = round(achieved / FP32_MFMA_PEAK_TFLOPS, 4)
    return out


def main():
    args = parse_args()
    if args.rehearse:
        return rehearse(args)
    import torch
    import torch.distributed as dist
    import synth
    from katsdpimager_amd import accel, grid, parallel, _lib

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    distributed = world > 1 or args.force_dist
    if distributed:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        if 'MASTER_ADDR' not in os.environ:         # (--force-dist without a launcher)
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(parallel.free_port()),
                              RANK='0', WORLD_SIZE='1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    if args.gpus != world:
        raise SystemExit('--gpus {} but WORLD_SIZE is {}: launch with torchrun'.format(
            args.gpus, world))
    _lib.lib()      # fail loudly if the HIP extension is missing
    ctx = accel.Context(local_rank)
    q = ctx.create_command_queue()
    dev = ctx.device
    if distributed:
        # one process per GPU, every rank on its own device (one node: LOCAL_RANK = device index)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        rank_devices = parallel.check_rank_devices(torch.cuda.current_device(), dev)
    else:
        rank_devices = [torch.cuda.current_device()]
    P, K, W, G = args.polarizations, args.kernel_width, args.w_planes, args.pixels

    def barrier():
        if distributed:
            dist.barrier(device_ids=[local_rank])

    # ---- the channel-independent inputs: made on rank 0, broadcast over RCCL / xGMI (SURVEY 8e:
    # UVW in metres float32 [N][3] -- 0.6 GB for 50 M visibilities -- and the image taper); every
    # rank then quantises the tracks for its own channel's cell size --------------------------------
    t_bcast, bcast_bytes = 0.0, 0
    uvw = None
    if distributed:
        # (ConvolutionKernel.taper, grid.py:404-423: it depends on the anti-aliasing width and the
        # oversampling only, not on the channel)
        xs = np.arange(G) / G - 0.5
        taper = (grid.kaiser_bessel_fourier(xs, 7.0, grid.antialias_beta(7.0))
                 * np.sinc(xs / 8)).astype(np.float32) if rank == 0 else None
        shared = {
            'uvw': synth.track_uvw(args.vis, dev) if rank == 0
            else torch.empty((args.vis, 3), dtype=torch.float32, device=dev),
            'taper1d': torch.from_numpy(taper).to(dev) if rank == 0
            else torch.empty((G,), dtype=torch.float32, device=dev)}
        check = float(shared['uvw'][::9973].double().sum()) if rank == 0 else 0.0
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        parallel.broadcast_shared(shared, src=0)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
        bcast_bytes = sum(t.numel() * t.element_size() for t in shared.values())
        # (every rank holds rank 0's bytes: the gathered spot checksums agree)
        sums = parallel.gather_stats([float(shared['uvw'][::9973].double().sum())], dev)[:, 0].tolist()
        assert all(x == sums[0] for x in sums) and (rank != 0 or sums[0] == check), sums
        uvw = shared['uvw']

    # ---- this rank's channel -------------------------------------------------------------
    channel = rank_channel(rank, world)
    obs = synth.make_observation(G, args.vis, W, P, device=dev, cover=0.30,
                                 channel_scale=channel_scale(channel), seed=2 + rank, uvw=uvw)
    ip, gp, ap = synth.make_parameters(obs, P, K)
    n_vis = obs.n_vis
    vb = args.vis_block
    n_chunks = -(-n_vis // vb)
    pad = n_chunks * vb - n_vis

    def padded(t):
        if pad == 0:
            return t
        z = torch.zeros((pad,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        return torch.cat([t, z])
    uv_all, wp_all, vis_all = padded(obs.uv), padded(obs.w_plane), padded(obs.vis)
    n_pad = n_chunks * vb

    templates = {a: grid.GridderTemplate(ctx, ip.fixed, gp.fixed,
                                         {'variant': args.variant, 'arith': a})
                 for a in ('fp32', 'split_fp16')}
    slice_ops = {a: t.instantiate(q, ap, ip, gp, n_pad) for a, t in templates.items()}
    fn = slice_ops['fp32']
    Gg = fn.slots['grid'].shape[1]
    gen = torch.Generator(device=dev)
    gen.manual_seed(2)
    wg = accel.DeviceArray(ctx, (P, Gg, Gg), np.float32,
                           tensor=torch.rand((P, Gg, Gg), generator=gen, device=dev))
    grid_buf = accel.DeviceArray(ctx, (P, Gg, Gg), np.complex64)
    whole = dict(uv=accel.DeviceArray(ctx, (n_pad, 4), np.int16, tensor=uv_all),
                 w_plane=accel.DeviceArray(ctx, (n_pad,), np.int16, tensor=wp_all),
                 vis=accel.DeviceArray(ctx, (n_pad, P), np.complex64, tensor=vis_all))
    for op in slice_ops.values():
        op.bind(grid=grid_buf, weights_grid=wg, **whole)
        op.ensure_all_bound()
        op.num_vis = n_vis
        # the resident store measures a slice's order once (VisibilityReaderDevice._locality) and
        # hands the answer to the gridder with every chunk: do the same here, outside the timing
        jump_fraction = op.measure_locality()
    chunk_ops = {a: t.instantiate(q, ap, ip, gp, vb) for a, t in templates.items()}
    for op in chunk_ops.values():
        op.bind(grid=grid_buf, weights_grid=wg)
        op.ensure_all_bound()
        op.locality_hint = fn.locality_hint
    chunks = []
    for i in range(n_chunks):
        s = slice(i * vb, (i + 1) * vb)
        chunks.append((accel.DeviceArray(ctx, (vb, 4), np.int16, tensor=uv_all[s]),
                       accel.DeviceArray(ctx, (vb,), np.int16, tensor=wp_all[s]),
                       accel.DeviceArray(ctx, (vb, P), np.complex64, tensor=vis_all[s]),
                       min(vb, n_vis - i * vb)))
    torch.cuda.synchronize()        # torch produced the inputs on its own stream

    def slice_step(arith):
        op = slice_ops[arith]

        def step(e0, e1):
            grid_buf.zero(q)                            # imager.clear_grid()
            if e0 is not None:
                e0.record(q.stream)
            op._run()                                   # imager.grid(): the whole slice, one launch
            if e1 is not None:
                e1.record(q.stream)
        return step

    def chunk_step(arith):
        op = chunk_ops[arith]

        def step(e0, e1):
            grid_buf.zero(q)
            if e0 is not None:
                e0.record(q.stream)
            for uv_c, wp_c, vis_c, n in chunks:
                op.bind(uv=uv_c, w_plane=wp_c, vis=vis_c)
                op.num_vis = n
                op._run()
            if e1 is not None:
                e1.record(q.stream)
        return step

    timer = Timer(q, barrier, dev)
    # ---- the headline: float32 arithmetic, one launch per slice ---------------------------
    sec_per_step, kern_ms = timer.run(slice_step('fp32'), args.steps, args.warmup)
    mvis = world * n_vis / sec_per_step / 1e6
    roofline = grid_roofline('fp32', K, P, n_vis, 1, kern_ms, args.variant)
    # every rank's own numbers, so that the first multi-GPU run localises a slow rank
    per_rank = parallel.gather_stats(
        [float(channel), timer.local_sec_per_step * 1e3, kern_ms, roofline['frac'],
         float(torch.cuda.current_device())], dev if distributed else 'cpu')
    traffic = load_traffic(args, roofline)
    if traffic is not None:
        roofline['traffic'] = traffic

    workload = ('{0}: 1 channel per GPU, {1}^2 image, {2} W-planes, {3} vis, K={4}, P={5}, '
                'one gridder launch per W-slice (resident store)').format(
        'C2' if (G, W, P) == (4096, 32, 1) else 'C4' if (G, W, P) == (8192, 64, 4) else 'custom',
        G, W, n_vis, K, P)
    result = {
        'metric': 'Mvis/s gridded (4096^2 grid, 32 W-planes)', 'value': round(mvis, 2),
        'unit': 'Mvis/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(sec_per_step * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': workload, 'grid_size': Gg, 'channels': world,
                   'band_channel_of_rank0': channel, 'parallelism': 'channel-sharded',
                   'window_jump_fraction': round(jump_fraction, 5),
                   'broadcast_ms': round(t_bcast * 1e3, 3), 'broadcast_MB': round(bcast_bytes / 1e6, 1),
                   'broadcast_GBps': round(bcast_bytes / t_bcast / 1e9, 1) if t_bcast > 0 else None,
                   'process_group': 'nccl (RCCL), {} rank(s)'.format(world) if distributed else None},
        'roofline': roofline,
        'per_rank': [{'rank': r, 'band_channel': int(row[0]), 'ms_per_step': round(row[1], 3),
                      'kernel_ms': round(row[2], 3), 'roofline_frac': round(row[3], 4),
                      'device': int(row[4])} for r, row in enumerate(per_rank.tolist())],
    }
    assert [p['device'] for p in result['per_rank']] == rank_devices

    # ---- the opt-in fp16 hi/lo form, same step, scored against the pipe it uses -----------
    if args.variant != 'generic' and K <= 64:
        s_split, k_split = timer.run(slice_step('split_fp16'), args.steps, args.warmup)
        result['split_fp16'] = {
            'value': round(world * n_vis / s_split / 1e6, 2), 'unit': 'Mvis/s',
            'ms_per_step': round(s_split * 1e3, 3),
            'dtype': 'f32 accumulation, operands as fp16 hi/lo pairs (22 bits, lo*lo dropped)',
            'roofline': grid_roofline('split_fp16', K, P, n_vis, 1, k_split, args.variant),
        }

    if rank == 0 and world == 1 and args.cpu_sample > 0:
        result['cpu_baseline'] = cpu_baseline(args, obs, fn, wg, Gg)

    if rank == 0 and not args.no_secondary:
        sec = {}
        # accuracy of both forms against float64 on one vis_block chunk of this workload
        mid = (n_chunks // 2) * vb
        mid = min(mid, max(n_vis - vb, 0))
        cn = min(vb, n_vis - mid)
        truth = synth.grid_truth_fp64(fn.convolve_kernel.data, obs.uv[mid:mid + cn],
                                      obs.w_plane[mid:mid + cn], obs.vis[mid:mid + cn],
                                      wg.tensor, K)
        tmax = float(truth.abs().max())
        for arith in ('fp32', 'split_fp16'):
            if arith == 'split_fp16' and 'split_fp16' not in result:
                continue
            op = chunk_ops[arith]
            grid_buf.zero(q)
            s = slice(mid, mid + vb)
            op.bind(uv=accel.DeviceArray(ctx, (vb, 4), np.int16, tensor=uv_all[s]),
                    w_plane=accel.DeviceArray(ctx, (vb,), np.int16, tensor=wp_all[s]),
                    vis=accel.DeviceArray(ctx, (vb, P), np.complex64, tensor=vis_all[s]))
            op.num_vis = cn
            op._run()
            q.finish()
            err = float((grid_buf.tensor.to(torch.complex128) - truth).abs().max()) / tmax
            block = result if arith == 'fp32' else result['split_fp16']
            block['max_norm_error_vs_fp64'] = float('%.3g' % err)
        del truth
        # the reference's chunking: vis_block-sized launches on one stream
        for arith in ('fp32', 'split_fp16'):
            if arith == 'split_fp16' and 'split_fp16' not in result:
                continue
            s_c, k_c = timer.run(chunk_step(arith), 3, 1, reduce=False)
            r = grid_roofline(arith, K, P, n_vis, n_chunks, k_c, args.variant)
            sec['chunked_' + arith] = {
                'Mvis_per_s': round(n_vis / s_c / 1e6, 1), 'vis_block': vb,
                'avg_launch_us': r['avg_launch_us'], 'frac': r['frac']}
        sec.update(secondary(args, ctx, q, obs, ip, gp, ap, fn, grid_buf, chunks, whole, n_vis))
        sec.update(geometry_sweep(args, ctx, q, dev))
        sec.update(other_configs(args, ctx, q, dev))
        extra_cpu = sec.pop('_cpu_baseline', None)
        if extra_cpu and 'cpu_baseline' in result:
            result['cpu_baseline'].update(extra_cpu)
        result['secondary'] = sec
        result['production_order'] = production_order(args, ctx, q, obs, ip, gp, ap, wg, grid_buf)
        if args.extras:
            result['extras'] = extras(args, ctx, q, obs, ip, gp, ap, templates['fp32'], grid_buf,
                                      wg, uv_all, wp_all, vis_all)
    if rank == 0 and not args.no_major_loop and not args.no_secondary:
        del slice_ops, chunk_ops, chunks
        result['major_cycle_loop'] = major_cycle_loop(args, ctx, q, obs, extras=args.extras)
        if args.variant != 'generic' and K <= 64:
            # the same loop with the opt-in fp16 hi/lo form in the gridder and the degridder
            result['major_cycle_loop_split_fp16'] = major_cycle_loop(
                args, ctx, q, obs, arith='split_fp16', add_sources=False)
    barrier()
    if rank == 0:
        print(json.dumps(result))
    if distributed:
        dist.destroy_process_group()


def load_traffic(args, roofline):
    """roofline.traffic: HBM bytes per launch (FETCH_SIZE + WRITE_SIZE, corrected as
    MI355X_MICROARCH.md prescribes) from a rocprofv3 --pmc pass of THIS command, handed over by
    tools/profile_round.sh.  Never a stale number: the file must describe the same launch."""
    if not args.traffic_json or not os.path.exists(args.traffic_json):
        return None
    try:
        t = json.load(open(args.traffic_json))
    except Exception:
        return None
    same = (t.get('kernel') == roofline['kernel'] and t.get('vis_per_launch') == roofline['vis_per_launch']
            and t.get('arith') == 'fp32' and t.get('kernel_width') == args.kernel_width
            and t.get('polarizations') == args.polarizations)
    return t.get('bytes_per_launch') if same else None


def secondary(args, ctx, q, obs, ip, gp, ap, gridder, grid_buf, chunks, whole=None, n_whole=0):
    """CLEAN minor-cycles/s (second half of BASELINE's metric), FFT + layer_to_image, degrid,
    DFT predict."""
    import torch
    from katsdpimager_amd import accel, grid, image, clean, parameters
    out = {}
    P, G = args.polarizations, args.pixels

    # grid -> image: at w = 0 (every slice of C2 / C5) the Hermitian part of the grid goes through a
    # complex-to-real transform of half the size -- for an even layer size without a prime factor
    # above 7 in two launches of the library's own (kimg_grid_to_image_real: only the columns the grid reaches,
    # fold / padding / image correction fused), else kimg_grid_to_half_layer + rocFFT C2R in place +
    # kimg_real_layer_to_image; at w != 0 pad/shift + rocFFT C2C + layer_to_image
    Gg = grid_buf.shape[1]
    routes = {'own': {}, 'library': {'own_transform': False}}
    times = {}
    for route, tuning in routes.items():
        template = image.GridImageTemplate(ctx, np.float32, tuning)
        g2i = template.instantiate_grid_to_image(
            q, grid_buf.shape, float(ip.pixel_size), -0.5 * G * float(ip.pixel_size),
            template.make_fft_plan((G, G)))
        g2i.bind(grid=grid_buf)
        g2i.ensure_all_bound()
        g2i.buffer('kernel1d').set(q, gridder.convolve_kernel.taper(G).astype(np.float32))
        g2i.buffer('image').zero(q)

        def g2i_time(w, overwrite=False):
            g2i.set_w(w)
            g2i()
            q.finish()
            t0 = time.perf_counter()
            for _ in range(5):
                g2i.overwrite_next = overwrite
                g2i()
            q.finish()
            return (time.perf_counter() - t0) / 5
        times[route] = g2i_time(0.0)
        if route == 'own':
            own = g2i._own_transform()
            dt_write = g2i_time(0.0, True) if own else None
            dt_c2c = g2i_time(1.0)
        else:
            dt_c2c_library = g2i_time(1.0)
        del g2i
    dt = times['own']
    out['grid_to_image_ms'] = round(dt * 1e3, 3)
    if own:
        out['grid_to_image_route'] = ('w = 0: own transforms, two launches (columns the grid reaches, '
                                      'then row pairs with the image correction)')
        out['grid_to_image_write_ms'] = round(dt_write * 1e3, 3)    # first slice: no read of the image
        out['grid_to_image_library_plan_ms'] = round(times['library'] * 1e3, 3)
        # HBM traffic of the route per polarization: the grid, the columns' transforms out and in,
        # the image read and written
        route_bytes = P * (8 * Gg * Gg + 2 * 8 * (Gg // 2 + 1) * G + 8 * G * G)
    else:
        out['grid_to_image_route'] = 'w = 0: Hermitian part + complex-to-real transform (in place, half layer)'
        # half layer written, transformed (2 passes, read + write each), read; image read + written
        route_bytes = P * (8 * Gg * Gg + 4 * G * G + 16 * G * G + 4 * G * G + 8 * G * G)
    out['grid_to_image_route_GBps'] = round(route_bytes / dt / 1e9, 1)
    out['grid_to_image_route_frac_of_8TBps'] = round(route_bytes / dt / 1e9 / HBM_PEAK_GBS, 4)
    # the same work in the reference's formulation (complex layer): read the grid, write + FFT the
    # layer (2 passes, read + write each), read it again, read-modify-write the image
    g2i_bytes = P * (8 * Gg * Gg + 8 * G * G + 32 * G * G + 8 * G * G + 8 * G * G)
    out['grid_to_image_reference_formulation_MB'] = round(g2i_bytes / 1e6, 1)
    out['grid_to_image_route_MB'] = round(route_bytes / 1e6, 1)
    # w != 0: complex layer (own transforms: the Gg columns the grid reaches, then every row)
    out['grid_to_image_c2c_ms'] = round(dt_c2c * 1e3, 3)
    out['grid_to_image_c2c_library_plan_ms'] = round(dt_c2c_library * 1e3, 3)
    out['grid_to_image_c2c_GBps'] = round(g2i_bytes / dt_c2c / 1e9, 1)

    # restoring-beam convolution of one polarization plane (R2C + Gaussian + C2R, beam.py:351-398)
    from katsdpimager_amd import beam
    conv = beam.ConvolveBeamTemplate(ctx, (G, G), np.float32).instantiate(q)
    conv.beam = beam.Beam(1.0, 2.5, 1.8, 0.4)
    conv.ensure_all_bound()
    conv.buffer('image').zero(q)
    conv()
    q.finish()
    t0 = time.perf_counter()
    for _ in range(5):
        conv()
    q.finish()
    out['convolve_beam_ms'] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    del conv

    # CLEAN: dirty = 200 point sources (x) PSF + noise, patch from psf_patch
    rs = np.random.RandomState(4)
    g1 = np.exp(-0.5 * ((np.arange(G) - G // 2) / 6.0) ** 2).astype(np.float32)
    psf = np.outer(g1, g1)[None].repeat(P, axis=0).astype(np.float32)
    psf += (0.002 * rs.standard_normal(psf.shape)).astype(np.float32)
    psf[:, G // 2, G // 2] = 1.0
    sky = (0.01 * rs.standard_normal((P, G, G))).astype(np.float32)
    for _ in range(200):
        y, x = rs.randint(100, G - 100, 2)
        amp = rs.uniform(0.5, 2.0)
        sky[:, y - 30:y + 31, x - 30:x + 31] += amp * psf[:, G // 2 - 30:G // 2 + 31,
                                                          G // 2 - 30:G // 2 + 31]
    cp = parameters.CleanParameters(args.clean_cycles, 0.1, 0.85, 5.0, 0, 0.01, 0.5, 0.02)
    cl = clean.CleanTemplate(ctx, cp, np.float32, P).instantiate(q, ip)
    cl.ensure_all_bound()
    cl.buffer('psf').set(q, psf)
    pp = clean.PsfPatchTemplate(ctx, np.float32, P).instantiate(q, (P, G, G))
    pp.bind(psf=cl.buffer('psf'))
    patch = pp(cp.psf_cutoff, cp.psf_limit)

    def clean_rate(patch_, per_cycle=False, op=None, image=None):
        op = op or cl
        op.buffer('dirty').set(q, sky if image is None else image)
        op.buffer('model').zero(q)
        op.reset()
        n = min(args.clean_cycles, 200) if per_cycle else args.clean_cycles
        q.finish()
        t0 = time.perf_counter()
        if not per_cycle:
            # (the loop, the host's share of it and the read-back of what it logged, as arrays:
            # the per-cycle tuples of the reference's API are made from them on demand)
            op.run_cycles(patch_, 0.0, n, collect=False)
            done = len(op._collect_cycle_arrays()[0])
        else:
            done = 0
            for _ in range(n):
                v, p_, m = op(patch_, 0.0)
                done += v is not None
        q.finish()
        return done / (time.perf_counter() - t0)
    clean_rate(patch)           # first call: hipGraph capture + instantiation
    small = (P, min(111, patch[1]), min(133, patch[2]))
    clean_rate(small)
    large = clean_rate(patch)
    out['clean'] = {
        # metric half 2: minor cycles per second, device-resident loop, incl. the final read-back
        'large_patch': list(patch), 'large_patch_cycles_per_s': round(large, 1),
        'large_patch_us_per_cycle': round(1e6 / large, 2),
        'small_patch': list(small),
        'small_patch_cycles_per_s': round(clean_rate(small), 1),
        'per_cycle_host_sync_cycles_per_s': round(clean_rate(patch, per_cycle=True), 1),
        # a launch is a latency chain: kernel boundary ~2 us + 6 us of dependent
        # steps, whatever it commits; the multi-component form commits up to 8 components with it
        'bound': 'latency',
    }
    out['clean']['small_patch_us_per_cycle'] = round(1e6 / out['clean']['small_patch_cycles_per_s'], 2)
    launches = cl.last_launches()
    if launches:
        out['clean']['small_patch_launches'] = launches
        out['clean']['small_patch_components_per_launch'] = round(args.clean_cycles / launches, 2)
        out['clean']['small_patch_us_per_launch'] = round(
            1e6 * args.clean_cycles / out['clean']['small_patch_cycles_per_s'] / launches, 2)
    # the same loop with one component per launch (the form of rounds 2-3), for comparison
    one = clean.CleanTemplate(ctx, cp, np.float32, P, {'form': 'one_launch'}).instantiate(q, ip)
    one.bind(**{name: cl.buffer(name) for name in ('dirty', 'model', 'psf', 'tile_max', 'tile_pos')})
    one.ensure_all_bound()
    clean_rate(small, op=one)
    out['clean']['small_patch_one_component_per_launch_cycles_per_s'] = round(clean_rate(small, op=one), 1)
    # a field with a few sources far above the rest (the first cycles of any real image): the loop
    # steps the same peaks several times per launch; next to it the same loop held
    # to single steps
    bright = sky.copy()
    for amp in (100.0, 40.0):
        y, x = rs.randint(100, G - 100, 2)
        bright[:, y - 30:y + 31, x - 30:x + 31] += amp * psf[:, G // 2 - 30:G // 2 + 31,
                                                             G // 2 - 30:G // 2 + 31]
    clean_rate(small, image=bright)
    out['clean']['dominated_field_cycles_per_s'] = round(clean_rate(small, image=bright), 1)
    launches = cl.last_launches()
    if launches:
        out['clean']['dominated_field_components_per_launch'] = round(args.clean_cycles / launches, 2)
    single = clean.CleanTemplate(ctx, cp, np.float32, P, {'form': 'multi', 'repeats': 1}).instantiate(q, ip)
    single.bind(**{name: cl.buffer(name) for name in ('dirty', 'model', 'psf', 'tile_max', 'tile_pos')})
    single.ensure_all_bound()
    clean_rate(small, op=single, image=bright)
    out['clean']['dominated_field_single_steps_cycles_per_s'] = round(clean_rate(small, op=single, image=bright), 1)
    launches = single.last_launches()
    if launches:
        out['clean']['dominated_field_single_steps_components_per_launch'] = round(args.clean_cycles / launches, 2)
    del one, single, bright
    # several channels per launch (kimg_clean_cycles_batch: cycle i of C channels in ONE launch, the
    # kernel boundary is paid once): aggregate minor cycles per second over C channels of a band,
    # each with its own dirty image (the same sky at another amplitude and noise), same PSF patch
    extra = []
    for c in range(7):
        e = clean.CleanTemplate(ctx, cp, np.float32, P).instantiate(q, ip)
        e.ensure_all_bound()
        e.buffer('psf').set(q, psf)
        extra.append(e)
    skies = [sky] + [(sky * (1.0 + 0.07 * (c + 1))
                      + (0.002 * np.random.RandomState(50 + c).standard_normal(sky.shape))).astype(np.float32)
                     for c in range(7)]

    def batch_rate(C):
        ops = [cl] + extra[:C - 1]
        for op, sk in zip(ops, skies):
            op.buffer('dirty').set(q, sk)
            op.buffer('model').zero(q)
            op.reset()
        q.finish()
        t0 = time.perf_counter()
        runs = clean.run_cycles_batch(ops, [small] * C, [0.0] * C, [args.clean_cycles] * C)
        q.finish()
        return sum(len(r) for r in runs) / (time.perf_counter() - t0)
    batch_rate(2)                # first call: graph capture
    for C in (2, 4, 8):
        batch_rate(C)
        out['clean']['batched_%d_channels_cycles_per_s' % C] = round(batch_rate(C), 1)
    del extra, skies
    # bytes a cycle moves: 12 P patch^2 + tile refresh; only meaningful for big patches
    cyc_bytes = 12 * P * patch[1] * patch[2] + 4 * P * 1024 * ((patch[1] + 31) // 32 + 1) * ((patch[2] + 31) // 32 + 1)
    out['clean']['large_patch_GBps'] = round(cyc_bytes * large / 1e9, 1)
    if args.cpu_sample > 0:
        out['_cpu_baseline'] = cpu_baseline_rest(args, obs, gridder, sky, psf, small, ip, cp)
    del cl, pp

    # degridder over every chunk of the channel (hot loop of the 2nd+ major cycles with --degrid,
    # frontend.py:128-139): vis -= weights * degrid(model grid)
    wts = accel.DeviceArray(ctx, (args.vis_block, P), np.float32,
                            tensor=torch.ones((args.vis_block, P), device=ctx.device))
    torch.cuda.synchronize()        # torch filled `wts` on its own stream
    total = sum(c[3] for c in chunks)
    flop_per_vis = 8.0 * args.kernel_width ** 2 * P
    for arith in ('fp32', 'split_fp16'):
        dg = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed, {'arith': arith}).instantiate(
            q, ap, ip, gp, args.vis_block)
        dg.bind(grid=grid_buf, weights=wts)
        dg.ensure_all_bound()
        # the order of the slice was measured once (as the resident store does) for the gridder;
        # its answer goes to the degridder with every chunk, without it `auto` would measure -- and
        # synchronise -- per call
        dg.locality_hint = gridder.locality_hint

        def degrid_all():
            for uv_c, wp_c, vis_c, n in chunks:
                dg.bind(uv=uv_c, w_plane=wp_c, vis=vis_c)
                dg.num_vis = n
                dg._run()
        degrid_all()
        q.finish()
        t0 = time.perf_counter()
        for _ in range(2):
            degrid_all()
        q.finish()
        rate = 2 * total / (time.perf_counter() - t0)
        peak = FP32_MFMA_PEAK_TFLOPS if arith == 'fp32' else F16_MFMA_PEAK_TFLOPS
        out['degrid_' + arith] = {'Mvis_per_s': round(rate / 1e6, 1),
                                  'frac': round(flop_per_vis * rate / 1e12 / peak, 4), 'peak': peak,
                                  'vis_block': args.vis_block}
        del dg
        if whole is not None:
            # ... and as the resident-store driver runs it: one launch over the whole slice
            n_pad = whole['uv'].shape[0]
            dgs = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed, {'arith': arith}).instantiate(
                q, ap, ip, gp, n_pad)
            wts_all = accel.DeviceArray(ctx, (n_pad, P), np.float32,
                                        tensor=torch.ones((n_pad, P), device=ctx.device))
            torch.cuda.synchronize()
            dgs.bind(grid=grid_buf, weights=wts_all, **whole)
            dgs.ensure_all_bound()
            dgs.num_vis = n_whole
            dgs.locality_hint = gridder.locality_hint
            dgs._run()
            q.finish()
            t0 = time.perf_counter()
            for _ in range(3):
                dgs._run()
            q.finish()
            rate = 3 * n_whole / (time.perf_counter() - t0)
            out['degrid_slice_' + arith] = {'Mvis_per_s': round(rate / 1e6, 1),
                                            'frac': round(flop_per_vis * rate / 1e12 / peak, 4),
                                            'peak': peak}
            del dgs, wts_all

    # direct (DFT) prediction of a 1000-component model: the reference's default predictor when
    # --degrid is not given (frontend.py:113-138, predict.py:419-438)
    from katsdpimager_amd import predict
    S = 1000
    pr = predict.PredictTemplate(ctx, np.float32, P).instantiate(q, ip, gp, args.vis_block, S)
    pr.bind(weights=wts)
    pr.ensure_all_bound()
    rs = np.random.RandomState(3)
    lm = rs.uniform(-0.4, 0.4, (S, 2)) * float(ip.image_size)
    lmn = np.concatenate([lm, np.sqrt(1 - np.sum(lm * lm, axis=1, keepdims=True)) - 1], axis=1)
    pr.set_sky_arrays(lmn.astype(np.float32), rs.uniform(0.1, 1, (S, P)).astype(np.float32))
    pr.set_w(0.0)

    def predict_some(nchunks):
        for uv_c, wp_c, vis_c, n in chunks[:nchunks]:
            pr.bind(uv=uv_c, w_plane=wp_c, vis=vis_c)
            pr.num_vis = n
            pr._run()
    nch = min(8, len(chunks))
    predict_some(1)
    q.finish()
    t0 = time.perf_counter()
    predict_some(nch)
    q.finish()
    dt = time.perf_counter() - t0
    pairs = sum(c[3] for c in chunks[:nch]) * S
    out['predict_Gpairs_per_s'] = round(pairs / dt / 1e9, 1)
    out['predict_sources'] = S
    return out


def geometry_sweep(args, ctx, q, dev):
    """Other geometries and input orders, float32 arithmetic, one launch per slice: the
    reference's default geometry (kernel width 60, frontend.py:325; hundreds of W planes from the
    default w-step, frontend.py:318) and the orders SURVEY 8d lists."""
    import torch
    import synth
    from katsdpimager_amd import accel, grid
    out = {}
    P, G, K = args.polarizations, args.pixels, args.kernel_width
    n2 = min(args.vis, 16 * args.vis_block)

    def rate(obs2, Kx, arith='fp32', store_order=False):
        """M records/s of one gridder launch over the whole stream.  The stream's order is measured
        once, outside the timing, as the resident store does per slice (VisibilityReaderDevice.
        _locality); `store_order`: first re-ordered the way the store does when it is closed
        (kimg_store_reorder, no merge), and the cost of that one pass is returned too."""
        ip2, gp2, ap2 = synth.make_parameters(obs2, P, Kx)
        n = obs2.n_vis
        uv, wp, vis = obs2.uv, obs2.w_plane, obs2.vis
        reorder_ms = None
        if store_order:
            from katsdpimager_amd import preprocess
            arrays = dict(uv=accel.DeviceArray(ctx, (n, 4), np.int16, tensor=uv),
                          w_plane=accel.DeviceArray(ctx, (n,), np.int16, tensor=wp),
                          weights=accel.DeviceArray(ctx, (n, P), np.float32, tensor=obs2.weights),
                          vis=accel.DeviceArray(ctx, (n, P), np.complex64, tensor=vis))
            torch.cuda.synchronize()
            for _ in range(2):          # (the first call loads the kernels)
                q.finish()
                t0 = time.perf_counter()
                out, kept = preprocess.reorder_device_arrays(q, P, n, arrays, Kx, obs2.oversample,
                                                             obs2.w_planes, False)
                q.finish()
                reorder_ms = (time.perf_counter() - t0) * 1e3
            uv, wp, vis = out['uv'].tensor, out['w_plane'].tensor, out['vis'].tensor
        op = grid.GridderTemplate(ctx, ip2.fixed, gp2.fixed, {'variant': args.variant, 'arith': arith}) \
            .instantiate(q, ap2, ip2, gp2, n)
        shape = op.slots['grid'].shape
        op.bind(grid=accel.DeviceArray(ctx, shape, np.complex64),
                weights_grid=accel.DeviceArray(ctx, shape, np.float32,
                                               tensor=torch.ones(shape, device=dev)),
                uv=accel.DeviceArray(ctx, (n, 4), np.int16, tensor=uv),
                w_plane=accel.DeviceArray(ctx, (n,), np.int16, tensor=wp),
                vis=accel.DeviceArray(ctx, (n, P), np.complex64, tensor=vis))
        op.ensure_all_bound()
        op.num_vis = n
        torch.cuda.synchronize()
        if store_order:
            op.locality_hint = True         # what the store hands on for a re-ordered slice
        else:
            op.measure_locality()
        op._run()
        q.finish()
        reps = 2
        t0 = time.perf_counter()
        for _ in range(reps):
            op._run()
        q.finish()
        taken.append(op.last_variant)
        r = round(reps * n / (time.perf_counter() - t0) / 1e6, 1)
        return (r, round(reorder_ms, 3)) if store_order else r

    taken = []
    obs2 = synth.make_observation(G, n2, 256, P, device=dev, seed=5)
    out['grid_256_planes_Mvis_per_s'] = rate(obs2, K)
    if K != 60:
        out['grid_k60_256_planes_Mvis_per_s'] = rate(obs2, 60)
        out['grid_k60_256_planes_split_fp16_Mvis_per_s'] = rate(obs2, 60, 'split_fp16')
    del obs2
    # input orders (same visibilities, W=32): as generated (baseline-major, every baseline's whole
    # track), loader-shaped (baseline-sorted runs per load block of `dumps` time samples, adjacent
    # equal cells merged as preprocess.cpp:334-397 does), time-major, shuffled
    obs3 = synth.make_observation(G, n2, args.w_planes, P, device=dev, seed=6)
    out['order_baseline_major_Mvis_per_s'] = rate(obs3, K)
    for name, fn_order in (('loader_blocks', synth.order_loader_blocks),
                           ('time_major', synth.order_time_major),
                           ('shuffled', synth.order_shuffled)):
        o = fn_order(obs3)
        out['order_%s_Mvis_per_s' % name] = rate(o['obs'], K)
        out['order_%s_note' % name] = o['note'] + '; variant taken by auto: ' + str(taken[-1])
        # steady state of a channel: the resident store re-orders the slice once (when it is
        # closed); every gridding / degridding pass after that runs the window kernel on it
        r, ms = rate(o['obs'], K, store_order=True)
        out['order_%s_store_order_Mvis_per_s' % name] = r
        out['order_%s_store_reorder_ms' % name] = ms
    return out


def production_order(args, ctx, q, obs, ip, gp, ap, wg, grid_buf):
    """The SAME channel as the reference's gridder receives it, and as this package's gridder
    receives it after the resident store has been closed.

    The headline stream is uncompressed (every dump of every baseline, whole tracks).  The
    reference's preprocessor never delivers that: its loaders hand over blocks of ~256 dumps sorted
    by baseline (loader_ms.py:465-468) and compress() merges neighbouring records that fall on the
    same sub-cell (preprocess.cpp:334-397).  `as_delivered` grids exactly that stream with the
    window kernel; `store_order` after the once-per-channel re-order of the resident store
    (kimg_store_reorder: strips of grid columns swept along v), `store_merged` with the
    whole-slice merge that goes with it by default.  Rates are in stored RECORDS per second and in
    INPUT visibilities per second (the channel's 50 M); each roofline block counts the algorithmic
    flops of the records actually gridded."""
    import torch
    import synth
    from katsdpimager_amd import accel, grid, preprocess
    P, K = args.polarizations, args.kernel_width
    o = synth.order_loader_blocks(obs)
    m = o['obs']
    out = {'stream': o['note'], 'input_visibilities': obs.n_vis}
    n0 = m.n_vis
    arrival = dict(uv=accel.DeviceArray(ctx, (n0, 4), np.int16, tensor=m.uv),
                   w_plane=accel.DeviceArray(ctx, (n0,), np.int16, tensor=m.w_plane),
                   weights=accel.DeviceArray(ctx, (n0, P), np.float32, tensor=m.weights),
                   vis=accel.DeviceArray(ctx, (n0, P), np.complex64, tensor=m.vis))
    tg = grid.GridderTemplate(ctx, ip.fixed, gp.fixed, {'variant': args.variant})
    td = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed, {'variant': args.variant})
    g = tg.instantiate(q, ap, ip, gp, n0)
    d = td.instantiate(q, ap, ip, gp, n0)
    resid = accel.DeviceArray(ctx, (n0, P), np.complex64)
    torch.cuda.synchronize()

    def timed(op, reps=3):
        op._run()
        q.finish()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(q.stream)
        for _ in range(reps):
            op._run()
        e1.record(q.stream)
        q.finish()
        return e0.elapsed_time(e1) / reps

    def measure(arrays, n, ordered):
        g.bind(grid=grid_buf, weights_grid=wg, uv=arrays['uv'], w_plane=arrays['w_plane'],
               vis=arrays['vis'])
        d.bind(grid=grid_buf, uv=arrays['uv'], w_plane=arrays['w_plane'], weights=arrays['weights'],
               vis=resid)
        for op in (g, d):
            op.ensure_all_bound()
            op.num_vis = n
        if ordered:
            g.locality_hint = d.locality_hint = True
            jumps = None
        else:
            jumps = g.measure_locality()
            d.locality_hint = g.locality_hint
        grid_buf.zero(q)
        ms = timed(g)
        roof = grid_roofline('fp32', K, P, n, 1, ms, args.variant)
        ms_d = timed(d)
        block = {
            'records': n, 'grid_ms': round(ms, 4),
            'records_per_s_M': round(n / ms / 1e3, 1),
            'input_vis_per_s_M': round(obs.n_vis / ms / 1e3, 1),
            'variant': g.last_variant,
            'roofline': {k: roof[k] for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac',
                                              'flop_per_vis', 'vis_per_launch', 'avg_launch_us')},
            'degrid_ms': round(ms_d, 4), 'degrid_records_per_s_M': round(n / ms_d / 1e3, 1),
            'degrid_frac': round(8.0 * K * K * P * n / (ms_d * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
        }
        if jumps is not None:
            block['window_jump_fraction'] = round(jumps, 5)
        return block
    out['as_delivered'] = measure(arrival, n0, False)
    for label, merge in (('store_order', False), ('store_merged', True)):
        for _ in range(2):              # (the first call loads the kernels)
            q.finish()
            t0 = time.perf_counter()
            arrays, kept = preprocess.reorder_device_arrays(q, P, n0, arrival, K, obs.oversample,
                                                            obs.w_planes, merge)
            q.finish()
            ms = (time.perf_counter() - t0) * 1e3
        block = measure(arrays, kept, True)
        # paid once per channel, when the store is closed (includes the 8-byte read-back of the count)
        block['reorder_ms_once_per_channel'] = round(ms, 3)
        out[label] = block
        del arrays
    out['speedup_store_order_over_as_delivered'] = round(
        out['as_delivered']['grid_ms'] / out['store_order']['grid_ms'], 2)
    return out


def other_configs(args, ctx, q, dev):
    """Two more whole-channel gridding measurements in the default run (float32, one launch per
    W-slice, same timing as the headline): BASELINE config 4 (8192^2, 64 W-planes, 4 polarizations)
    and the C2 geometry with the longest baseline at 0.49 of the image size, so that the grid is
    (nearly) the full 4096^2 the metric's wording names (the headline's `cover` = 0.30 gives a
    2486^2 grid: SURVEY 8d)."""
    import torch
    import synth
    from katsdpimager_amd import accel, grid
    out = {}

    def one(G, W, P, cover, seed):
        obs = synth.make_observation(G, args.vis, W, P, device=dev, cover=cover, seed=seed)
        ip, gp, ap = synth.make_parameters(obs, P, args.kernel_width)
        n = obs.n_vis
        op = grid.GridderTemplate(ctx, ip.fixed, gp.fixed, {'variant': args.variant}) \
            .instantiate(q, ap, ip, gp, n)
        shape = op.slots['grid'].shape
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        op.bind(grid=accel.DeviceArray(ctx, shape, np.complex64),
                weights_grid=accel.DeviceArray(ctx, shape, np.float32,
                                               tensor=torch.rand(shape, generator=gen, device=dev)),
                uv=accel.DeviceArray(ctx, (n, 4), np.int16, tensor=obs.uv),
                w_plane=accel.DeviceArray(ctx, (n,), np.int16, tensor=obs.w_plane),
                vis=accel.DeviceArray(ctx, (n, P), np.complex64, tensor=obs.vis))
        op.ensure_all_bound()
        op.num_vis = n
        torch.cuda.synchronize()
        op.measure_locality()
        op._run()
        q.finish()
        reps = 3
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(q.stream)
        for _ in range(reps):
            op._run()
        e1.record(q.stream)
        q.finish()
        ms = e0.elapsed_time(e1) / reps
        launches = (P + 1) // 2                 # two polarizations per launch
        roof = grid_roofline('fp32', args.kernel_width, P, n, launches, ms, args.variant)
        return {'Mvis_per_s': round(n / ms / 1e3, 1), 'ms_per_pass': round(ms, 3),
                'grid_size': shape[1], 'pixels': G, 'w_planes': W, 'polarizations': P,
                'visibilities': n, 'frac': roof['frac'], 'achieved_TFLOPs': roof['achieved'],
                'peak': roof['peak'], 'launches_per_pass': launches}
    out['full_cover_grid'] = one(args.pixels, args.w_planes, args.polarizations, 0.49, 11)
    out['full_cover_grid_Mvis_per_s'] = out['full_cover_grid']['Mvis_per_s']
    if (args.pixels, args.w_planes, args.polarizations) == (4096, 32, 1):
        out['c4'] = one(8192, 64, 4, 0.30, 12)
        out['c4']['workload'] = ('C4: 8192^2 image, 64 W-planes, 4 polarizations, K={}, {} vis, one '
                                 'pass = all polarizations').format(args.kernel_width, args.vis)
        out['c4'].update(c4_clean(args, ctx, q))
    return out


def c4_clean(args, ctx, q):
    """CLEAN at BASELINE config 4's image: 8192^2, four polarizations, sum-of-squares peak metric
    (clean.py:28-31), 200 sources (x) PSF + noise, 133 x 111 patch: minor cycles per second of the
    device-resident loop (read-back included) as `auto` runs it and with one component per launch."""
    from katsdpimager_amd import clean, parameters
    G, P = 8192, 4
    rs = np.random.RandomState(14)
    g1 = np.exp(-0.5 * ((np.arange(G) - G // 2) / 6.0) ** 2).astype(np.float32)
    psf = np.repeat(np.outer(g1, g1)[None].astype(np.float32), P, axis=0)
    sky = (0.01 * rs.standard_normal((P, G, G))).astype(np.float32)
    h = 30
    for _ in range(200):
        y, x = rs.randint(100, G - 100, 2)
        amp = (rs.uniform(0.5, 2.0) * np.array([1.0, 0.3, -0.2, 0.1], np.float32))[:, None, None]
        sky[:, y - h:y + h + 1, x - h:x + h + 1] += amp * psf[:, G // 2 - h:G // 2 + h + 1, G // 2 - h:G // 2 + h + 1]
    fixed = parameters.FixedImageParameters(list(range(P)), np.float32)
    ip = parameters.ImageParameters(fixed, 1.0, None, 0.2, None, pixel_size=1e-5, pixels=G)
    cp = parameters.CleanParameters(args.clean_cycles, 0.1, 0.85, 5.0, 1, 0.01, 0.5, 0.02)
    out = {}
    patch = (P, 111, 133)
    for name, tuning in (('clean_cycles_per_s', None), ('clean_one_component_per_launch_cycles_per_s',
                                                        {'form': 'one_launch'})):
        op = clean.CleanTemplate(ctx, cp, np.float32, P, tuning).instantiate(q, ip)
        op.ensure_all_bound()
        op.buffer('psf').set(q, psf)
        best = 0.0
        for rep in range(2):
            op.buffer('dirty').set(q, sky)
            op.buffer('model').zero(q)
            op.reset()
            q.finish()
            t0 = time.perf_counter()
            op.run_cycles(patch, 0.0, args.clean_cycles, collect=False)
            done = len(op._collect_cycle_arrays()[0])
            q.finish()
            best = max(best, done / (time.perf_counter() - t0))
        out[name] = round(best, 1)
        if tuning is None and op.last_launches():
            out['clean_components_per_launch'] = round(args.clean_cycles / op.last_launches(), 2)
        del op
    return out


def major_cycle_loop(args, ctx, q, obs, extras=False, arith='fp32', add_sources=True):
    """BASELINE config 5: the per-channel loop of frontend.process_channel (frontend.py:465-585)
    on the Imaging facade with the channel resident in HBM and one launch per W-slice: robust
    weights -> PSF -> 2 major cycles of { grid -> FFT -> noise estimate -> CLEAN minor cycles ->
    degrid + regrid }.  Wall-clock milliseconds per stage (queue drained after each stage)."""
    import torch
    from katsdpimager_amd import accel, imaging, parameters, weight
    from katsdpimager_amd import preprocess as _pp
    P, G = args.polarizations, args.pixels
    import synth
    ipd, gpd, apd = synth.make_parameters(obs, P, args.kernel_width, degrid=True)
    cp = parameters.CleanParameters(args.clean_cycles, 0.1, 0.85, 5.0, 0, 0.01, 0.5, 0.02)
    wparm = parameters.WeightParameters(weight.WeightType.ROBUST, 0.0)
    tuning = {'gridder': {'arith': arith}, 'degridder': {'arith': arith}}
    template = imaging.ImagingTemplate(ctx, apd, ipd.fixed, wparm, gpd.fixed, cp, tuning)
    n = obs.n_vis
    im = template.instantiate(q, ipd, gpd, n, 0, 2)
    im.ensure_all_bound()
    # a sky with something to CLEAN: 200 point sources + noise (the gridder measurements above
    # used uniform random visibilities, as the reference's own tests do)
    if add_sources:
        synth.add_point_sources(obs, 200, seed=4, noise=0.01)
    vis = obs.vis
    chunk = _pp.DeviceChunk(
        n, accel.DeviceArray(ctx, (n, 4), np.int16, tensor=obs.uv),
        accel.DeviceArray(ctx, (n,), np.int16, tensor=obs.w_plane),
        accel.DeviceArray(ctx, (n, P), np.float32, tensor=obs.weights),
        accel.DeviceArray(ctx, (n, P), np.complex64, tensor=vis))
    torch.cuda.synchronize()
    # the resident store measures the order of a stored slice once (VisibilityReaderDevice._locality)
    # and hands the answer on with every chunk; without it the `auto` variant of gridder and
    # degridder would measure -- and synchronise -- in every pass of the loop
    im.set_chunk_device(chunk, 'vis')
    im._gridder.measure_locality()
    chunk.locality = im._gridder.locality_hint
    times = {}

    from katsdpimager_amd import trace
    if getattr(args, 'roctx', False):
        trace.enable()

    def timed(name, fn):
        q.finish()
        t0 = time.perf_counter()
        with trace.range(name):
            out = fn()
            q.finish()
        times[name] = times.get(name, 0.0) + time.perf_counter() - t0
        return out

    def make_weights():
        im.clear_weights()
        im.grid_weights_device(chunk)
        return im.finalize_weights()

    def grid_pass(field, predict):
        im.clear_grid()
        # zero-copy coordinates / weights, device-to-device copy of the visibilities
        im.set_chunk_device(chunk, field)
        if predict:
            im.predict(0.0)
        im.grid()

    # first use of a kernel loads its code object (milliseconds): not part of a channel's cost
    make_weights()
    grid_pass('weights', False)
    q.finish()
    timed('weights', make_weights)
    im.clear_dirty()
    timed('grid_psf', lambda: grid_pass('weights', False))
    timed('fft', lambda: im.grid_to_image(0.0))
    dirty = im.get_buffer('dirty')
    scale = np.reciprocal(dirty[:, G // 2, G // 2])
    im.scale_dirty(scale)
    im.dirty_to_psf()
    patch = timed('psf_patch', im.psf_patch)
    im.clear_model()
    minor = 0
    for major in range(2):
        im.clear_dirty()
        if major:
            timed('model_to_grid', lambda: im.model_to_grid(0.0))
        timed('degrid_grid' if major else 'grid', lambda: grid_pass('vis', major > 0))
        timed('fft', lambda: im.grid_to_image(0.0))
        im.scale_dirty(scale)
        noise = timed('noise_est', im.noise_est)
        im.clean_reset()
        peak = im.clean_cycle(patch)
        thr = max(float(noise) * 5.0, 0.15 * float(peak))
        vals = timed('clean', lambda: im.clean_cycles(patch, thr, args.clean_cycles - 1))
        minor += 1 + len(vals)
    total = sum(times.values())
    out = {k + '_ms': round(v * 1e3, 3) for k, v in times.items()}
    out['total_ms'] = round(total * 1e3, 3)
    out['minor_cycles'] = minor
    out['clean_cycles_per_s'] = round((minor - 2) / times['clean'], 1)
    out['psf_patch'] = list(patch)
    out['visibilities'] = n
    out['arith'] = arith
    del im
    if arith == 'fp32':
        keep = {}
        out['store_driven'] = store_driven(args, ctx, q, obs, template, ipd, gpd, cp, wparm, keep)
        if extras:
            out['extras'] = major_loop_extras(args, ctx, q, obs, template, ipd, gpd, cp, wparm,
                                              keep['reader'])
    return out


def store_driven(args, ctx, q, obs, template, ipd, gpd, cp, wparm, keep=None):
    """BASELINE config 5 the way the product runs it: raw visibilities -> device preprocessing
     -> HBM-resident store (8f-2; re-ordered and merged once when it is closed) ->
    frontend.process_channel (weights, PSF, 2 major cycles of up to 1000 minor cycles, degridding).
    The same with the store left in arrival order is timed next to it (`arrival_order_*`: what
    round 2 ran)."""
    import torch
    from katsdpimager_amd import accel, frontend, preprocess, trace
    P = args.polarizations
    out = {}
    n = obs.n_vis
    import synth
    raw_vis = torch.where((obs.uvw[:, 2] < 0)[:, None], torch.conj(obs.vis), obs.vis)
    raw_vis = torch.where(torch.isfinite(raw_vis.real), raw_vis, torch.zeros_like(raw_vis))
    # in the order the reference's loaders deliver: blocks of 256 dumps, each sorted by baseline
    # (loader_ms.py:465-468; tools/synth.order_loader_blocks)
    b, t, nb, T = synth._track_indices(obs)
    dumps = 256
    order = torch.argsort((t // dumps) * (nb * dumps) + b * dumps + (t % dumps))
    del b, t
    d_uvw = accel.DeviceArray(ctx, (n, 3), np.float32, tensor=obs.uvw[order].contiguous())
    d_wts = accel.DeviceArray(ctx, (1, n, P), np.float32, tensor=obs.weights[order][None].contiguous())
    d_vis = accel.DeviceArray(ctx, (1, n, P), np.complex64, tensor=raw_vis[order][None].contiguous())
    del order, raw_vis
    out['input_order'] = 'loader blocks of {} dumps, baseline-sorted'.format(dumps)
    ident = np.identity(P, np.complex64)
    torch.cuda.synchronize()
    sizes = [('preprocess_Mvis_per_s', args.vis_block)]
    if args.extras:
        sizes.insert(0, ('preprocess_16x_buffer_Mvis_per_s', 16 * args.vis_block))
    for label, bs in sizes:
        for rep in range(2):            # the first pass warms up the kernels and the allocator
            coll = preprocess.VisibilityCollectorDevice(q, [ipd], [gpd], bs)
            q.finish()
            t0 = time.perf_counter()
            coll.add(d_uvw, d_wts, d_vis, None, None, ident, None)
            q.finish()
            dt = time.perf_counter() - t0
        out[label] = round(n / dt / 1e6, 1)
    plain = preprocess.VisibilityCollectorDevice(q, [ipd], [gpd], args.vis_block, reorder=False)
    plain.add(d_uvw, d_wts, d_vis, None, None, ident, None)
    plain.close()
    q.finish()
    t0 = time.perf_counter()
    coll.close()                        # the once-per-channel re-order + whole-slice merge
    q.finish()
    out['store_reorder_ms_once_per_channel'] = round((time.perf_counter() - t0) * 1e3, 3)
    out['input_visibilities'] = n
    out['records_after_compress'] = int(coll.num_output)
    out['records_stored'] = int(coll.num_stored)
    out['store_MB'] = round(coll.nbytes() / 1e6, 1)
    for label, c in (('arrival_order', plain), ('store_order', coll)):
        reader = c.reader()
        block = max(reader.len(0, s) for s in range(reader.num_w_slices(0)))
        im = template.instantiate(q, ipd, gpd, block, 0, 2)
        im.ensure_all_bound()
        for rep in range(2):
            entries = []
            trace.record_timeline(entries if rep else None)
            q.finish()
            t0 = time.perf_counter()
            stats = frontend.process_channel(reader, 0, im, ipd, gpd, cp, wparm.weight_type,
                                             block, 2, True)
            q.finish()
            dt = time.perf_counter() - t0
        trace.record_timeline(None)
        stages = {}
        for _, name, a, b in entries:
            key = name.split('[')[0]
            stages[key] = stages.get(key, 0.0) + (b - a) * 1e3
        out[label + '_total_ms'] = round(dt * 1e3, 3)
        out[label + '_minor_cycles'] = int(stats['minor']) if stats else None
        # host-side time of the driver's stages (a stage that ends in a read-back also waits for
        # the device work enqueued before it)
        out[label + '_stage_ms'] = {k: round(v, 3) for k, v in stages.items()}
        del im
    if keep is not None:
        keep['reader'] = coll.reader()
    return out


def major_loop_extras(args, ctx, q, obs, template, ipd, gpd, cp, wparm, reader):
    """Four channels with 1-4 of them in flight (frontend.process_channels: one host thread and one
    stream per channel, the CLEAN cycles of the channels in flight sharing their launches)."""
    import torch
    from katsdpimager_amd import frontend, imaging, parameters
    out = {}
    block = max(reader.len(0, s) for s in range(reader.num_w_slices(0)))
    # Four channels (here: the same stored channel imaged four times) with 1, 2, 3 and 4 of them in
    # flight on their own streams; CLEAN thresholds forced low so that every major cycle runs its
    # full 1000 minor cycles, as in the staged loop above.
    cp2 = parameters.CleanParameters(args.clean_cycles, 0.1, 1.0, 0.0, 0, 0.01, 0.5, 0.02)
    template2 = imaging.ImagingTemplate(ctx, template.array_parameters, ipd.fixed, wparm, gpd.fixed, cp2)
    jobs = []
    for _ in range(4):
        qi = ctx.create_command_queue()
        imi = template2.instantiate(qi, ipd, gpd, block, 0, 2)
        imi.ensure_all_bound()
        jobs.append(dict(reader=reader, rel_channel=0, imager=imi, image_p=ipd, grid_p=gpd,
                         clean_p=cp2, weight_type=wparm.weight_type, vis_block=block,
                         major=2, degrid=True))
    frontend.process_channels(jobs, workers=4)          # warm-up (graph capture per imager)
    for workers in (1, 2, 3, 4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = frontend.process_channels(jobs, workers=workers, stagger=False)
        torch.cuda.synchronize()
        out['four_channels_%d_in_flight_ms' % workers] = round((time.perf_counter() - t0) * 1e3, 2)
        if workers > 1:
            # the channels taking turns at gridding (round 3's default; see the 12-channel stream below)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            frontend.process_channels(jobs, workers=workers, stagger=True)
            torch.cuda.synchronize()
            out['four_channels_%d_in_flight_turns_ms' % workers] = round(
                (time.perf_counter() - t0) * 1e3, 2)
        out['four_channels_%d_in_flight_clean_batches' % workers] = [
            list(b) for b in frontend.process_channel_stream.last_batches]
    out['four_channels_minor_cycles'] = [int(r['minor']) for r in res]
    # A longer stream (12 channels, one imager per worker thread), channels in step against taking
    # turns at the throughput-bound stages (CleanBatcher phased): wall time per channel.
    for workers in (2, 4):
        for stagger in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            frontend.process_channel_stream(lambda channel, worker: jobs[worker], range(12),
                                            workers=workers, stagger=stagger)
            torch.cuda.synchronize()
            out['stream_12_channels_%d_in_flight_%s_ms_per_channel' % (
                workers, 'turns' if stagger else 'in_step')] = round(
                    (time.perf_counter() - t0) * 1e3 / 12, 2)
    return out


def extras(args, ctx, q, obs, ip, gp, ap, template, grid_buf, wg, uv_all, wp_all, vis_all):
    """PCIe-inclusive gridding (never `value`): the reference-style host path, every chunk copied
    from host memory (uv, w_plane, vis: 18 B per visibility at P=1) before it is gridded."""
    vb = args.vis_block
    out = {}
    host_fn = template.instantiate(q, ap, ip, gp, vb)
    host_fn.bind(grid=grid_buf, weights_grid=wg)
    host_fn.ensure_all_bound()
    h_uv = uv_all[:vb].cpu().numpy()
    h_wp = wp_all[:vb].cpu().numpy()
    h_vis = vis_all[:vb].cpu().numpy()
    host_fn.num_vis = vb

    def host_chunk():
        host_fn.buffer('uv').set(q, h_uv)
        host_fn.buffer('w_plane').set(q, h_wp)
        host_fn.buffer('vis').set(q, h_vis)
        host_fn._run()
    host_chunk()
    q.finish()
    t0 = time.perf_counter()
    for _ in range(8):
        host_chunk()
    q.finish()
    out['grid_host_chunks_Mvis_per_s'] = round(8 * vb / (time.perf_counter() - t0) / 1e6, 1)
    return out


def cpu_baseline(args, obs, gridder, wg, Gg):
    """The oracle's C restatement of the reference's numba `_grid` loop (grid.py:1032-1052),
    single thread (the reference CPU path is single-threaded), on a bounded sample: the median of
    `--cpu-passes` timed passes over the same `--cpu-sample` visibilities after one warm-up."""
    from oracle import kimg_oracle as orc
    S = min(args.cpu_sample, obs.n_vis)
    # a sample spread over the whole track set: 64 runs of consecutive visibilities
    runs = 64
    run_len = max(S // runs, 1)
    starts = np.linspace(0, obs.n_vis - run_len, runs).astype(np.int64)
    idx = (starts[:, None] + np.arange(run_len)[None, :]).reshape(-1)
    import torch
    sel = torch.from_numpy(idx).to(obs.uv.device)
    uv = obs.uv[sel].cpu().numpy()
    wp = obs.w_plane[sel].cpu().numpy()
    vis = obs.vis[sel].cpu().numpy()
    kernel = gridder.convolve_kernel.data
    P = vis.shape[1]
    g = np.zeros((P, Gg, Gg), np.complex64)
    wgrid = wg.tensor.cpu().numpy()
    uv01 = np.ascontiguousarray(uv[:, :2])
    uv23 = np.ascontiguousarray(uv[:, 2:])
    orc.grid(kernel, g, wgrid, uv01[:20000], uv23[:20000], wp[:20000], vis[:20000])     # warm-up
    times = []
    for _ in range(max(args.cpu_passes, 1)):
        t0 = time.perf_counter()
        orc.grid(kernel, g, wgrid, uv01, uv23, wp, vis)
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    out = {'value': round(len(idx) / dt / 1e6, 4), 'unit': 'Mvis/s', 'cores': 1, 'kind': 'port',
           'sample': 'median of {} passes over {} visibilities ({} runs of {} consecutive samples '
                     'spread over the channel), same grid / kernel table, {:.1f} s per pass'.format(
                         len(times), len(idx), runs, run_len, dt)}
    # A stricter comparator than the reference itself (which is single-threaded): the same loop
    # on every host core, one private grid per thread (ctypes releases the GIL), same sample.
    import concurrent.futures
    threads = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
    threads = max(1, min(threads, runs, 16))        # the host share that goes with one GPU
    bounds = np.linspace(0, runs, threads + 1).astype(np.int64) * run_len
    grids = [np.zeros((P, Gg, Gg), np.complex64) for _ in range(threads)]

    def work(t):
        sl = slice(int(bounds[t]), int(bounds[t + 1]))
        orc.grid(kernel, grids[t], wgrid, uv01[sl], uv23[sl], wp[sl], vis[sl])
    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(threads) as pool:
        list(pool.map(work, range(threads)))
    dt_all = time.perf_counter() - t0
    out['all_cores'] = {'value': round(len(idx) / dt_all / 1e6, 3), 'cores': threads,
                        'note': 'not the reference: its CPU path is single-threaded numba'}
    return out


def cpu_baseline_rest(args, obs, gridder, sky, psf, patch, ip, cp):
    """The other halves of the metric on the host, 1 thread, median of 5 (3 for the FFT), same
    inputs as the GPU measurements next to them:
      clean_cycles_per_s  the oracle's restatement of CleanHost (clean.py:1060-1075, _tile_peak
                          :946-968 in C, the PSF subtraction in numpy like the reference) on the
                          bench's CLEAN image, same patch as `secondary.clean.small_patch`;
      grid_to_image_ms    GridToImageHost (image.py:781-799): numpy ifft2 (pocketfft) of the padded
                          4096^2 layer + taper / n correction, one polarization, w = 0;
      degrid_Mvis_per_s   the C restatement of `_degrid` (grid.py:1138-1154)."""
    from oracle import kimg_oracle as orc
    out = {}
    G = args.pixels
    # CLEAN
    cycles = 200
    rates = []
    for _ in range(5):
        dirty = sky.copy()
        model = np.zeros_like(dirty)
        ref = orc.Clean(G, cp.border, cp.loop_gain, cp.mode, dirty, psf, model)
        ref.reset()
        t0 = time.perf_counter()
        for _ in range(cycles):
            ref(patch, 0.0)
        rates.append(cycles / (time.perf_counter() - t0))
    out['clean_cycles_per_s'] = round(float(np.median(rates)), 1)
    out['clean_sample'] = ('median of 5 runs of {} cycles, {}^2 image, patch {}x{}, CleanHost restated '
                           '(tile scan in C, subtraction in numpy)').format(cycles, G, patch[1], patch[2])
    # grid -> image
    Gg = gridder.slots['grid'].shape[1]
    full = np.zeros((1, G, G), np.complex64)
    lo = (G - Gg) // 2
    rs = np.random.RandomState(1)
    full[0, lo:lo + Gg, lo:lo + Gg] = (rs.standard_normal((Gg, Gg))
                                       + 1j * rs.standard_normal((Gg, Gg))).astype(np.complex64)
    k1d = gridder.convolve_kernel.taper(G).astype(np.float32)
    image = np.zeros((1, G, G), np.float32)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        orc.grid_to_image(full, image, k1d, float(ip.pixel_size), -0.5 * G * float(ip.pixel_size), 0.0)
        times.append(time.perf_counter() - t0)
    out['grid_to_image_ms'] = round(float(np.median(times)) * 1e3, 1)
    out['grid_to_image_sample'] = 'median of 3, one polarization, {}^2, numpy pocketfft, 1 thread'.format(G)
    del full, image
    # degrid
    S = min(max(args.cpu_sample // 4, 1), obs.n_vis)
    runs = 64
    run_len = max(S // runs, 1)
    starts = np.linspace(0, obs.n_vis - run_len, runs).astype(np.int64)
    idx = (starts[:, None] + np.arange(run_len)[None, :]).reshape(-1)
    import torch
    sel = torch.from_numpy(idx).to(obs.uv.device)
    uv = obs.uv[sel].cpu().numpy()
    wp = obs.w_plane[sel].cpu().numpy()
    vis = obs.vis[sel].cpu().numpy()
    wts = obs.weights[sel].cpu().numpy()
    P = vis.shape[1]
    model = (rs.standard_normal((P, Gg, Gg)) + 1j * rs.standard_normal((P, Gg, Gg))).astype(np.complex64)
    uv01 = np.ascontiguousarray(uv[:, :2])
    uv23 = np.ascontiguousarray(uv[:, 2:])
    kernel = gridder.convolve_kernel.data
    times = []
    for _ in range(5):
        v = vis.copy()
        t0 = time.perf_counter()
        orc.degrid(kernel, model, uv01, uv23, wp, wts, v)
        times.append(time.perf_counter() - t0)
    out['degrid_Mvis_per_s'] = round(len(idx) / float(np.median(times)) / 1e6, 4)
    out['degrid_sample'] = 'median of 5 passes over {} visibilities ({} runs of {})'.format(len(idx), runs, run_len)
    return out


def rehearse(args):
    """The multi-process skeleton of main() on CPU tensors (gloo): same rendezvous, broadcast,
    channel assignment, barrier-bracketed timing, max-over-ranks and one JSON line from rank 0 --
    with a sleep where the device work would be.  `value` is meaningless; `rehearsal` says so."""
    import torch
    import torch.distributed as dist
    import synth
    from katsdpimager_amd import parallel
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        dist.init_process_group('gloo')
    if args.gpus != world:
        raise SystemExit('--gpus {} but WORLD_SIZE is {}: launch with torchrun'.format(
            args.gpus, world))
    shared = {'baselines': torch.from_numpy(synth.baselines_equatorial()) if rank == 0
              else torch.empty((2016, 3), dtype=torch.float64)}
    parallel.broadcast_shared(shared, src=0)
    assert torch.equal(shared['baselines'], torch.from_numpy(synth.baselines_equatorial()))
    channel = rank_channel(rank, world)
    n_vis = min(args.vis, 20000)
    obs = synth.make_observation(256, n_vis, 8, 1, device='cpu', cover=0.30,
                                 channel_scale=channel_scale(channel), seed=2 + rank)
    checksum = float(obs.uv.to(torch.float64).abs().sum())

    def barrier():
        if world > 1:
            dist.barrier()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    local_ms = (time.perf_counter() - t0) / args.steps * 1e3      # this rank's own clock
    barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0)
    stats = parallel.gather_stats([float(channel), checksum, local_ms])
    devices = parallel.check_rank_devices(rank)        # (stands in for the GPU index of the rank)
    if rank == 0:
        print(json.dumps({
            'metric': 'Mvis/s gridded (4096^2 grid, 32 W-planes)',
            'value': round(world * n_vis / (elapsed / args.steps) / 1e6, 3), 'unit': 'Mvis/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'rehearsal': 'no device work: plumbing only',
            'config': {'workload': 'rehearsal', 'channels': world,
                       'band_channels': [int(c) for c in stats[:, 0].tolist()]},
            'per_rank': [{'rank': r, 'band_channel': int(row[0]), 'ms_per_step': round(row[2], 3),
                          'device': devices[r]} for r, row in enumerate(stats.tolist())]}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
