#!/usr/bin/env python3
"""Benchmark of the imaging hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): one spectral channel, 4096^2 image, 32 W-planes,
~50 M synthetic visibilities (tools/synth.py), float32, Stokes I, kernel width 28,
oversample 8, streamed through Gridder in --vis-block chunks (reference default 1 048 576,
frontend.py:357) with every input resident in HBM.  One "step" = clear the grid + grid all
chunks of the channel (the hot loop #1 of frontend.make_dirty, frontend.py:126-139).
With N > 1 (one process per GPU, torchrun) every rank images its own channel (frequency
spread +-3 %): the path shards by channel with no data-path collective (weak scaling);
rank 0 broadcasts the channel-independent tables over RCCL once at start-up.

Prints ONE JSON line on rank 0: metric "Mvis/s gridded", plus `roofline` for the dominant
kernel (grid_mfma_kernel), `cpu_baseline` (the oracle's single-thread C restatement timed on
this host) and secondary numbers (CLEAN minor-cycles/s, degrid, FFT).
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
    p.add_argument('--vis-block', type=int, default=1048576)
    p.add_argument('--variant', default='auto', choices=['auto', 'generic', 'mfma'])
    p.add_argument('--clean-cycles', type=int, default=1000)
    p.add_argument('--cpu-sample', type=int, default=16_000_000,
                   help='visibilities gridded by the CPU baseline (0 disables it)')
    p.add_argument('--streams', type=int, default=2, choices=[1, 2],
                   help='HIP streams per channel in the major-loop measurements')
    p.add_argument('--major-loop', action='store_true',
                   help='also time the full major-cycle loop (BASELINE config 5)')
    p.add_argument('--no-secondary', action='store_true',
                   help='skip the CLEAN / degrid / FFT secondary measurements')
    return p.parse_args()


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    import synth
    from katsdpimager_amd import accel, grid, image, clean, parameters, parallel, _lib

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # KIMG_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    # (ranks share devices); the driver's runs use the default, RCCL with one GPU per rank.
    backend = os.environ.get('KIMG_DIST_BACKEND', 'nccl')
    device_index = local_rank if backend == 'nccl' else local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(device_index)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', device_index))
        else:
            dist.init_process_group(backend)
    if args.gpus != world:
        raise SystemExit('--gpus {} but WORLD_SIZE is {}: launch with torchrun'.format(
            args.gpus, world))
    _lib.lib()      # fail loudly if the HIP extension is missing
    ctx = accel.Context(device_index)
    q = ctx.create_command_queue()
    dev = ctx.device
    P, K, W, G = args.polarizations, args.kernel_width, args.w_planes, args.pixels

    # ---- shared tables: computed on rank 0, broadcast over RCCL/xGMI -------
    t_bcast = 0.0
    shared = {'baselines': torch.from_numpy(synth.baselines_equatorial()).to(dev) if rank == 0
              else torch.empty((2016, 3), dtype=torch.float64, device=dev)}
    if world > 1:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        parallel.broadcast_shared(shared, src=0)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
        assert torch.equal(shared['baselines'].cpu(), torch.from_numpy(synth.baselines_equatorial()))

    # ---- this rank's channel (channel c -> rank c mod world; one channel per GPU here) ----
    channel = parallel.assign_channels(world, world, rank)[0]
    chan_scale = parallel.channel_frequency_scale(channel, world)
    # uv coordinates scale with frequency: keep the footprint inside the grid for every channel
    cover = 0.30 / (1.03 if world > 1 else 1.0)
    obs = synth.make_observation(G, args.vis, W, P, device=dev, cover=cover,
                                 channel_scale=chan_scale, seed=2 + rank)
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

    template = grid.GridderTemplate(ctx, ip.fixed, gp.fixed, {'variant': args.variant})
    fn = template.instantiate(q, ap, ip, gp, vb)
    Gg = fn.slots['grid'].shape[1]
    gen = torch.Generator(device=dev)
    gen.manual_seed(2)
    wg = accel.DeviceArray(ctx, (P, Gg, Gg), np.float32,
                           tensor=torch.rand((P, Gg, Gg), generator=gen, device=dev))
    fn.bind(weights_grid=wg)
    fn.ensure_all_bound()
    chunks = []
    for i in range(n_chunks):
        s = slice(i * vb, (i + 1) * vb)
        chunks.append((accel.DeviceArray(ctx, (vb, 4), np.int16, tensor=uv_all[s]),
                       accel.DeviceArray(ctx, (vb,), np.int16, tensor=wp_all[s]),
                       accel.DeviceArray(ctx, (vb, P), np.complex64, tensor=vis_all[s]),
                       min(vb, n_vis - i * vb)))
    grid_buf = fn.buffer('grid')
    ev_start = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev_stop = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(k=None):
        grid_buf.zero(q)                               # imager.clear_grid()
        if k is not None:
            ev_start[k].record(q.stream)
        for uv_c, wp_c, vis_c, n in chunks:
            fn.bind(uv=uv_c, w_plane=wp_c, vis=vis_c)
            fn.num_vis = n
            fn._run()                                  # imager.grid()
        if k is not None:
            ev_stop[k].record(q.stream)

    def barrier():
        if world > 1:
            if backend == 'nccl':
                dist.barrier(device_ids=[device_index])
            else:
                dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = parallel.max_over_ranks(t1 - t0, dev)
    ms_per_step = elapsed / args.steps * 1e3
    mvis = world * n_vis / (elapsed / args.steps) / 1e6

    # ---- roofline of the dominant kernel (gridder), HIP events on the kernel's stream ----
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev_start, ev_stop)]))
    launch_us = kern_ms * 1e3 / n_chunks
    flop_per_vis = 8.0 * K * K * P               # one complex MAC per tap and polarization
    achieved_tflops = flop_per_vis * n_vis / (kern_ms * 1e-3) / 1e12
    bytes_per_vis = 8 + 2 + 8 * P + 4 * P        # uv + w_plane + vis + weight gather
    roofline = {
        'kernel': 'grid_mfma_kernel' if args.variant != 'generic' else 'grid_generic_kernel',
        'bound': 'mfma', 'achieved': round(achieved_tflops, 3), 'peak': FP32_MFMA_PEAK_TFLOPS,
        'unit': 'TFLOP/s', 'frac': round(achieved_tflops / FP32_MFMA_PEAK_TFLOPS, 4),
        'traffic': None,
        'flop_per_vis': flop_per_vis, 'avg_launch_us': round(launch_us, 2),
        'vis_per_launch': vb,
        'hbm_algorithmic_GBps': round(bytes_per_vis * n_vis / (kern_ms * 1e-3) / 1e9, 1),
        'hbm_frac_of_8TBps': round(bytes_per_vis * n_vis / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        # the reference bench's own figure of merit: grid-point additions per second
        # (tests/imager_bench.py:204-208), N K^2 P / t
        'GGAPS': round(n_vis * K * K * P / (kern_ms * 1e-3) / 1e9, 1),
    }
    # Which matrix instruction carried the work.  The gridder's default form splits every fp32
    # operand into an fp16 hi/lo pair and puts two visibilities into one v_mfma_f32_32x32x16_f16
    # (fp32 accumulation, results within 2e-6 of the exact form); KIMG_GRID_F16=0 selects the exact
    # v_mfma_f32_32x32x2_f32.  `achieved` / `peak` / `frac` stay what they were -- algorithmic fp32
    # work against the fp32 matrix peak, i.e. against what the exact instruction could ever reach --
    # and `pipe_*` give the executed flops against the peak of the pipe actually used.
    f16_form = args.variant != 'generic' and K <= 64 and os.environ.get('KIMG_GRID_F16', '1') != '0'
    blocks = 1 if K <= 32 else 4                 # 2 x 2 tap blocks for wide kernels
    executed_per_vis = blocks * P * (2 * 32 * 32 * 16 * 2 / 2 if f16_form else 2 * 32 * 32 * 2 * 2)
    pipe_peak = F16_MFMA_PEAK_TFLOPS if f16_form else FP32_MFMA_PEAK_TFLOPS
    executed_tflops = executed_per_vis * n_vis / (kern_ms * 1e-3) / 1e12
    roofline.update({
        'form': ('fp16 hi/lo pairs, 2 visibilities per v_mfma_f32_32x32x16_f16' if f16_form
                 else 'v_mfma_f32_32x32x2_f32' if args.variant != 'generic' else 'per-tap atomics'),
        'pipe_peak': pipe_peak, 'pipe_executed': round(executed_tflops, 1),
        'pipe_frac': round(executed_tflops / pipe_peak, 4),
    })
    traffic_file = os.path.join(ROOT, 'profiles', 'gridder_traffic.json')
    if os.path.exists(traffic_file):
        try:
            roofline['traffic'] = json.load(open(traffic_file)).get('bytes_per_launch')
        except Exception:
            pass

    result = {
        'metric': 'Mvis/s gridded (4096^2 grid, 32 W-planes)', 'value': round(mvis, 2),
        'unit': 'Mvis/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32 (operands as fp16 hi/lo pairs, fp32 accumulation)' if f16_form else 'f32',
        'data': 'synthetic',
        'config': {'workload': 'C2: 1 channel per GPU, {0}^2 image, {1} W-planes, {2} vis, '
                               'K={3}, P={4}, vis_block={5}'.format(G, W, n_vis, K, P, vb),
                   'grid_size': Gg, 'channels': world, 'parallelism': 'channel-sharded',
                   'broadcast_ms': round(t_bcast * 1e3, 3)},
        'roofline': roofline,
    }

    if rank == 0 and world == 1 and args.cpu_sample > 0:
        result['cpu_baseline'] = cpu_baseline(args, obs, fn, wg, Gg)
    if rank == 0 and not args.no_secondary:
        sec = secondary(args, ctx, q, obs, ip, gp, ap, fn, chunks)
        # all chunks of the channel resident and gridded by ONE launch (what a resident
        # visibility store, SURVEY 8f-2, buys over vis_block-sized launches)
        big = template.instantiate(q, ap, ip, gp, n_chunks * vb)
        big.bind(grid=grid_buf, weights_grid=wg,
                 uv=accel.DeviceArray(ctx, (n_chunks * vb, 4), np.int16, tensor=uv_all),
                 w_plane=accel.DeviceArray(ctx, (n_chunks * vb,), np.int16, tensor=wp_all),
                 vis=accel.DeviceArray(ctx, (n_chunks * vb, P), np.complex64, tensor=vis_all))
        big.num_vis = n_vis
        big._run()
        q.finish()
        t0 = time.perf_counter()
        for _ in range(3):
            big._run()
        q.finish()
        sec['grid_single_launch_Mvis_per_s'] = round(3 * n_vis / (time.perf_counter() - t0) / 1e6, 1)
        # Chunk launches alternated over two HIP streams (both accumulate into the same grid with
        # atomics): the tail of one launch overlaps the head of the next, which recovers the
        # single-launch rate while keeping vis_block-sized launches.  Not used for `value`: the
        # per-launch durations of overlapping kernels no longer add up to the wall time, so the
        # roofline accounting above would not apply.
        q2 = ctx.create_command_queue()
        fn_b = template.instantiate(q2, ap, ip, gp, vb)
        fn_b.bind(grid=grid_buf, weights_grid=wg)
        fn_b.ensure_all_bound()
        pair = (fn, fn_b)

        def grid_two_streams():
            for i, (uv_c, wp_c, vis_c, n) in enumerate(chunks):
                g_ = pair[i & 1]
                g_.bind(uv=uv_c, w_plane=wp_c, vis=vis_c)
                g_.num_vis = n
                g_._run()
            q.finish()
            q2.finish()
        grid_two_streams()
        t0 = time.perf_counter()
        for _ in range(3):
            grid_two_streams()
        sec['grid_two_streams_Mvis_per_s'] = round(3 * n_vis / (time.perf_counter() - t0) / 1e6, 1)
        # the same per-chunk launches as `value`, with the exact v_mfma_f32_32x32x2_f32 instruction
        # instead of the default fp16 hi/lo form (the library reads the variable per call)
        if f16_form:
            def grid_chunks_once():
                for uv_c, wp_c, vis_c, n in chunks:
                    fn.bind(uv=uv_c, w_plane=wp_c, vis=vis_c)
                    fn.num_vis = n
                    fn._run()
                q.finish()
            os.environ['KIMG_GRID_F16'] = '0'
            try:
                grid_chunks_once()
                t0 = time.perf_counter()
                for _ in range(3):
                    grid_chunks_once()
                sec['grid_exact_fp32_Mvis_per_s'] = round(3 * n_vis / (time.perf_counter() - t0) / 1e6, 1)
            finally:
                os.environ.pop('KIMG_GRID_F16', None)
        del fn_b
        # PCIe-inclusive: the reference-style host path, every chunk copied from host memory
        # (uv, w_plane, vis: 18 B per visibility at P=1) before it is gridded
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
        sec['grid_host_chunks_Mvis_per_s'] = round(8 * vb / (time.perf_counter() - t0) / 1e6, 1)
        # many W planes (the reference's default w-step gives hundreds per slice): the kernel
        # table no longer fits LDS and is read from a padded copy in HBM
        del host_fn, big
        n2 = min(n_vis, 16 * vb)
        obs2 = synth.make_observation(G, n2, 256, P, device=dev, seed=5)
        ip2, gp2, ap2 = synth.make_parameters(obs2, P, K)
        fn2 = template.instantiate(q, ap2, ip2, gp2, vb)
        grid2 = accel.DeviceArray(ctx, fn2.slots['grid'].shape, np.complex64)
        wg2 = accel.DeviceArray(ctx, fn2.slots['grid'].shape, np.float32,
                                tensor=torch.ones(fn2.slots['grid'].shape, device=dev))
        fn2.bind(grid=grid2, weights_grid=wg2)
        fn2.ensure_all_bound()
        torch.cuda.synchronize()

        def grid_obs2():
            for start in range(0, n2 - vb + 1, vb):
                sl = slice(start, start + vb)
                fn2.bind(uv=accel.DeviceArray(ctx, (vb, 4), np.int16, tensor=obs2.uv[sl]),
                         w_plane=accel.DeviceArray(ctx, (vb,), np.int16, tensor=obs2.w_plane[sl]),
                         vis=accel.DeviceArray(ctx, (vb, P), np.complex64, tensor=obs2.vis[sl]))
                fn2.num_vis = vb
                fn2._run()
        grid_obs2()
        q.finish()
        t0 = time.perf_counter()
        grid_obs2()
        q.finish()
        sec['grid_256_planes_Mvis_per_s'] = round((n2 // vb) * vb / (time.perf_counter() - t0) / 1e6, 1)
        # the reference's default geometry: kernel width 60 (frontend.py:325) on the same 256 planes
        if K != 60:
            del fn2
            ip3, gp3, ap3 = synth.make_parameters(obs2, P, 60)
            fn2 = grid.GridderTemplate(ctx, ip3.fixed, gp3.fixed, {'variant': args.variant}) \
                .instantiate(q, ap3, ip3, gp3, vb)
            del grid2, wg2
            shape3 = fn2.slots['grid'].shape
            fn2.bind(grid=accel.DeviceArray(ctx, shape3, np.complex64),
                     weights_grid=accel.DeviceArray(ctx, shape3, np.float32,
                                                    tensor=torch.ones(shape3, device=dev)))
            fn2.ensure_all_bound()
            torch.cuda.synchronize()
            grid_obs2()
            q.finish()
            t0 = time.perf_counter()
            grid_obs2()
            q.finish()
            sec['grid_k60_256_planes_Mvis_per_s'] = round(
                (n2 // vb) * vb / (time.perf_counter() - t0) / 1e6, 1)
        if args.major_loop:
            # PSF pass grids the weights as visibilities (frontend.py:511)
            wt_all = padded(obs.weights)
            psf_all = torch.complex(wt_all, torch.zeros_like(wt_all))
            obs.vis.copy_(torch.where(torch.isfinite(obs.vis.real), obs.vis, torch.zeros_like(obs.vis)))
            chunks_dev = []
            for i in range(n_chunks):
                sl = slice(i * vb, (i + 1) * vb)
                wt_c = accel.DeviceArray(ctx, (vb, P), np.float32, tensor=wt_all[sl])
                wt_c.psf_vis = accel.DeviceArray(ctx, (vb, P), np.complex64, tensor=psf_all[sl])
                chunks_dev.append((chunks[i][0], chunks[i][1], chunks[i][2], wt_c, chunks[i][3]))
            sec['major_cycle_loop'] = major_cycle_loop(args, ctx, q, obs, ip, gp, ap, chunks_dev)
        result['secondary'] = sec
    barrier()
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def secondary(args, ctx, q, obs, ip, gp, ap, gridder, chunks):
    """CLEAN minor-cycles/s (second half of BASELINE's metric), FFT + layer_to_image, degrid."""
    import torch
    from katsdpimager_amd import accel, grid, image, clean, parameters
    out = {}
    P, G = args.polarizations, args.pixels

    # grid -> image (pad/shift + rocFFT + layer_to_image)
    template = image.GridImageTemplate(ctx, np.float32)
    g2i = template.instantiate_grid_to_image(
        q, gridder.buffer('grid').shape, float(ip.pixel_size), -0.5 * G * float(ip.pixel_size),
        template.make_fft_plan((G, G)))
    g2i.bind(grid=gridder.buffer('grid'))
    g2i.ensure_all_bound()
    g2i.buffer('kernel1d').set(q, gridder.convolve_kernel.taper(G).astype(np.float32))
    g2i.buffer('image').zero(q)
    g2i()
    q.finish()
    t0 = time.perf_counter()
    for _ in range(5):
        g2i()
    q.finish()
    out['grid_to_image_ms'] = round((time.perf_counter() - t0) / 5 * 1e3, 3)

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
    dirty = g2i.buffer('image')
    img = dirty.get(q)
    peak = img[:, G // 2, G // 2].copy()
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
    for label in ('batched_first_call', 'batched', 'per_cycle'):
        # 'batched_first_call' includes the one-off capture + instantiation of the hipGraph
        cl.buffer('dirty').set(q, sky)
        cl.buffer('model').zero(q)
        cl.reset()
        n = min(args.clean_cycles, 200) if label == 'per_cycle' else args.clean_cycles
        q.finish()
        t0 = time.perf_counter()
        if label != 'per_cycle':
            done = len(cl.run_cycles(patch, 0.0, n))
        else:
            done = 0
            for _ in range(n):
                v, p_, m = cl(patch, 0.0)
                done += v is not None
        q.finish()
        out['clean_%s_cycles_per_s' % label] = round(done / (time.perf_counter() - t0), 1)
    out['clean_psf_patch'] = list(patch)
    # the same with the patch of a measured PSF (111 x 133, the --major-loop case): few lattice
    # blocks, so the cycle is ONE launch (every workgroup repeats the peak search)
    small = (P, min(111, patch[1]), min(133, patch[2]))
    for label in ('warm', 'timed'):
        cl.buffer('dirty').set(q, sky)
        cl.buffer('model').zero(q)
        cl.reset()
        q.finish()
        t0 = time.perf_counter()
        done = len(cl.run_cycles(small, 0.0, args.clean_cycles))
        q.finish()
        out['clean_small_patch_cycles_per_s'] = round(done / (time.perf_counter() - t0), 1)
    out['clean_small_patch'] = list(small)

    # degridder over every chunk of the channel (hot loop of the 2nd+ major cycles with --degrid,
    # frontend.py:128-139): vis -= weights * degrid(model grid)
    template_d = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed)
    dg = template_d.instantiate(q, ap, ip, gp, args.vis_block)
    dg.bind(grid=gridder.buffer('grid'))
    wts = accel.DeviceArray(ctx, (args.vis_block, P), np.float32,
                            tensor=torch.ones((args.vis_block, P), device=ctx.device))
    dg.bind(weights=wts)
    dg.ensure_all_bound()
    torch.cuda.synchronize()        # torch filled `wts` on its own stream

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
    total = sum(c[3] for c in chunks)
    out['degrid_Mvis_per_s'] = round(2 * total / (time.perf_counter() - t0) / 1e6, 2)

    # direct (DFT) prediction of a 1000-component model over every chunk: the reference's default
    # predictor when --degrid is not given (frontend.py:113-138, predict.py:419-438)
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


def major_cycle_loop(args, ctx, q, obs, ip, gp, ap, chunks_dev):
    """BASELINE config 5: the per-channel loop of frontend.process_channel (frontend.py:465-585)
    on the Imaging facade with every chunk resident in HBM: robust weights -> PSF -> 2 major
    cycles of { grid -> FFT -> noise estimate -> CLEAN minor cycles -> degrid + regrid }.
    Returns wall-clock seconds per stage (queue drained after each stage)."""
    import torch
    from katsdpimager_amd import accel, imaging, parameters, weight
    P, G = args.polarizations, args.pixels
    import synth
    ipd, gpd, apd = synth.make_parameters(obs, P, args.kernel_width, degrid=True)
    cp = parameters.CleanParameters(args.clean_cycles, 0.1, 0.85, 5.0, 0, 0.01, 0.5, 0.02)
    wparm = parameters.WeightParameters(weight.WeightType.ROBUST, 0.0)
    template = imaging.ImagingTemplate(ctx, apd, ipd.fixed, wparm, gpd.fixed, cp)
    im = template.instantiate(q, ipd, gpd, args.vis_block, 0, 2, streams=args.streams)
    im.ensure_all_bound()
    times = {}

    def timed(name, fn):
        q.finish()
        t0 = time.perf_counter()
        out = fn()
        q.finish()
        times[name] = times.get(name, 0.0) + time.perf_counter() - t0
        return out

    def make_weights():
        im.clear_weights()
        for uv_c, wp_c, vis_c, wt_c, n in chunks_dev:
            im.bind(uv=uv_c, weights=wt_c)
            im._weights.grid(n)
        return im.finalize_weights()

    from katsdpimager_amd import preprocess as _pp

    def grid_pass(field, predict):
        im.clear_grid()
        for uv_c, wp_c, vis_c, wt_c, n in chunks_dev:
            # zero-copy coordinates / weights, device-to-device copy of the visibilities; with
            # streams=2 consecutive chunks alternate between two HIP streams
            im.set_chunk_device(_pp.DeviceChunk(n, uv_c, wp_c, wt_c, vis_c), field)
            if predict:
                im.predict(0.0)
            im.grid()

    # first use of a kernel loads its code object (milliseconds): not part of a channel's cost
    uv_c, wp_c, vis_c, wt_c, n = chunks_dev[0]
    im.set_chunk_device(_pp.DeviceChunk(n, uv_c, wp_c, wt_c, vis_c), 'weights')
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
    out = {k + '_s': round(v, 4) for k, v in times.items()}
    out['total_s'] = round(total, 4)
    out['minor_cycles'] = minor
    out['psf_patch'] = list(patch)
    del im

    # The same channel from raw inputs: device preprocessing into the HBM-resident
    # store (8f-2), then the store-driven driver katsdpimager_amd.frontend.process_channel.
    from katsdpimager_amd import frontend, preprocess
    n = obs.n_vis
    raw_vis = torch.where((obs.uvw[:, 2] < 0)[:, None], torch.conj(obs.vis), obs.vis)
    raw_vis = torch.where(torch.isfinite(raw_vis.real), raw_vis, torch.zeros_like(raw_vis))
    d_uvw = accel.DeviceArray(ctx, (n, 3), np.float32, tensor=obs.uvw)
    d_wts = accel.DeviceArray(ctx, (1, n, P), np.float32, tensor=obs.weights[None].contiguous())
    d_vis = accel.DeviceArray(ctx, (1, n, P), np.complex64, tensor=raw_vis[None].contiguous())
    ident = np.identity(P, np.complex64)
    torch.cuda.synchronize()
    # the collector's buffer size is the user's choice (the reference passes --vis-block); each
    # buffer is a chain of ~13 small dependent launches, so larger buffers amortise it
    for rep in range(2):
        big = preprocess.VisibilityCollectorDevice(q, [ipd], [gpd], 16 * args.vis_block)
        q.finish()
        t0 = time.perf_counter()
        big.add(d_uvw, d_wts, d_vis, None, None, ident, None)
        q.finish()
        dt_big = time.perf_counter() - t0
        del big
    out['preprocess_16x_buffer_Mvis_per_s'] = round(n / dt_big / 1e6, 1)
    for rep in range(2):            # the first pass warms up the kernels and the allocator
        coll = preprocess.VisibilityCollectorDevice(q, [ipd], [gpd], args.vis_block)
        q.finish()
        t0 = time.perf_counter()
        coll.add(d_uvw, d_wts, d_vis, None, None, ident, None)
        q.finish()
        dt = time.perf_counter() - t0
        if rep == 0:
            del coll
    coll.close()
    reader = coll.reader()
    out['preprocess_Mvis_per_s'] = round(n / dt / 1e6, 1)
    out['preprocess_kept_fraction'] = round(coll.num_output / coll.num_input, 4)
    out['store_MB'] = round(coll.nbytes() / 1e6, 1)
    im = template.instantiate(q, ipd, gpd, args.vis_block, 0, 2, streams=args.streams)
    im.ensure_all_bound()
    for rep in range(2):
        q.finish()
        t0 = time.perf_counter()
        stats = frontend.process_channel(reader, 0, im, ipd, gpd, cp, wparm.weight_type,
                                         args.vis_block, 2, True)
        q.finish()
        dt = time.perf_counter() - t0
    out['store_driver_total_s'] = round(dt, 4)
    out['store_driver_minor_cycles'] = int(stats['minor']) if stats else None

    # Four channels (here: the same stored channel imaged four times) with 1, 2, 3 and 4 of them in
    # flight on their own streams; CLEAN thresholds forced low so that every major
    # cycle runs its full 1000 minor cycles, as in the staged loop above.
    cp2 = parameters.CleanParameters(args.clean_cycles, 0.1, 1.0, 0.0, 0, 0.01, 0.5, 0.02)
    template2 = imaging.ImagingTemplate(ctx, apd, ipd.fixed, wparm, gpd.fixed, cp2)
    jobs = []
    for _ in range(4):
        qi = ctx.create_command_queue()
        # one stream per channel: the device maps streams onto four hardware queues, and a
        # channel whose chain of CLEAN launches shares a queue with another one waits for it
        imi = template2.instantiate(qi, ipd, gpd, args.vis_block, 0, 2, streams=1)
        imi.ensure_all_bound()
        jobs.append(dict(reader=reader, rel_channel=0, imager=imi, image_p=ipd, grid_p=gpd,
                         clean_p=cp2, weight_type=wparm.weight_type, vis_block=args.vis_block,
                         major=2, degrid=True))
    frontend.process_channels(jobs, workers=4)          # warm-up (graph capture per imager)
    for workers in (1, 2, 3, 4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = frontend.process_channels(jobs, workers=workers)
        torch.cuda.synchronize()
        out['four_channels_%d_in_flight_s' % workers] = round(time.perf_counter() - t0, 4)
    out['four_channels_minor_cycles'] = [int(r['minor']) for r in res]
    return out


def cpu_baseline(args, obs, gridder, wg, Gg):
    """The oracle's C restatement of the reference's numba `_grid` loop (grid.py:1032-1052),
    single thread (the reference CPU path is single-threaded), on a bounded sample."""
    from oracle import kimg_oracle as orc
    S = min(args.cpu_sample, obs.n_vis)
    # a sample spread over the whole track set: every (n_vis // S)-block contributes a run
    runs = 64
    run_len = S // runs
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
    orc.grid(kernel, g, wgrid, uv[:1000, :2].copy(), uv[:1000, 2:].copy(), wp[:1000], vis[:1000])
    t0 = time.perf_counter()
    orc.grid(kernel, g, wgrid, np.ascontiguousarray(uv[:, :2]), np.ascontiguousarray(uv[:, 2:]),
             wp, vis)
    dt = time.perf_counter() - t0
    out = {'value': round(len(idx) / dt / 1e6, 4), 'unit': 'Mvis/s', 'cores': 1, 'kind': 'port',
           'sample': '{} visibilities ({} runs of {} consecutive samples spread over the '
                     'channel), same grid / kernel table, {:.1f} s'.format(len(idx), runs,
                                                                           run_len, dt)}
    # A stricter comparator than the reference itself (which is single-threaded): the same loop
    # on every host core, one private grid per thread (ctypes releases the GIL), same sample.
    import concurrent.futures
    threads = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
    threads = max(1, min(threads, runs, 16))        # the host share that goes with one GPU
    uv01 = np.ascontiguousarray(uv[:, :2])
    uv23 = np.ascontiguousarray(uv[:, 2:])
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


if __name__ == '__main__':
    main()
