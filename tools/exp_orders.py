#!/usr/bin/env python3
"""Experiment: which order of the STORED records suits the window gridder / degridder best?

The reference's gridder receives baseline-sorted load blocks, adjacent-merged
(loader_ms.py:465-468, preprocess.cpp:334-397).  The resident store may re-order a W-slice once
per channel; every later pass (weights, PSF, image, degrid, regrid) then runs on that order.  This
script builds the candidate orders with torch (no product code involved in the re-ordering) and
times the window kernels (variant 'mfma', float32) on each.

    python tools/exp_orders.py [--vis 16777216] [--dumps 256]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import synth                                                    # noqa: E402
from katsdpimager_amd import _lib as _kl
if os.environ.get('KIMG_VARIANT_LIB'):
    _kl.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build_variants',
                                'libkimg_%s.so' % os.environ['KIMG_VARIANT_LIB'])
from katsdpimager_amd import accel, grid                        # noqa: E402


def merged_loader_blocks(obs, dumps):
    """synth.order_loader_blocks, also returning the baseline of every merged record."""
    b, t, nb, T = synth._track_indices(obs)
    order = torch.argsort((t // dumps) * (nb * dumps) + b * dumps + (t % dumps))
    o = synth._reordered(obs, order)
    bl = b[order]
    return remerge(o, bl)


def remerge(o, bl):
    uv = o.uv.to(torch.int64)
    key = ((uv[:, 0] + 32768) << 48) | ((uv[:, 1] + 32768) << 32) | (uv[:, 2] << 24) \
        | (uv[:, 3] << 16) | o.w_plane.to(torch.int64)
    head = torch.ones(o.n_vis, dtype=torch.bool, device=key.device)
    head[1:] = (key[1:] != key[:-1]) | (bl[1:] != bl[:-1])
    seg = torch.cumsum(head.to(torch.int64), 0) - 1
    m = int(seg[-1]) + 1
    P = o.vis.shape[1]
    vis = torch.zeros((m, P, 2), dtype=torch.float32, device=key.device)
    vis.index_add_(0, seg, torch.view_as_real(o.vis))
    weights = torch.zeros((m, P), dtype=torch.float32, device=key.device)
    weights.index_add_(0, seg, o.weights)
    heads = torch.nonzero(head)[:, 0]
    return synth._copy_with(o, o.uv[heads], o.w_plane[heads], torch.view_as_complex(vis), weights), \
        bl[heads]


def tile_key(o, Gg, binsz, strips=False, transpose=False):
    half = Gg // 2
    u = o.uv[:, 0].to(torch.int64) + half
    v = o.uv[:, 1].to(torch.int64) + half
    if transpose:
        u, v = v, u
    bv = v // binsz
    if strips:
        nbu = Gg
        bu = u
    else:
        nbu = (Gg + binsz - 1) // binsz
        bu = u // binsz
    bu = torch.where((bv & 1) == 1, nbu - 1 - bu, bu)
    return bv * nbu + bu


def pad_bins(o, key, group=8):
    """Sorted-by-key copy in which every bin's run is padded to a multiple of `group` records with
    dead ones (coordinates of the bin's last record, zero visibility and weight)."""
    order = torch.argsort(key, stable=True)
    ks = key[order]
    uniq, counts = torch.unique_consecutive(ks, return_counts=True)
    padded = (counts + group - 1) // group * group
    starts_p = torch.cumsum(padded, 0) - padded
    starts = torch.cumsum(counts, 0) - counts
    total = int(padded.sum())
    # place i of sorted record -> starts_p[bin] + (i - starts[bin])
    binidx = torch.repeat_interleave(torch.arange(len(uniq), device=key.device), counts)
    pos = starts_p[binidx] + (torch.arange(len(ks), device=key.device) - starts[binidx])
    # source for every padded place: by default the last record of its bin
    last_src = order[starts + counts - 1]
    binidx_p = torch.repeat_interleave(torch.arange(len(uniq), device=key.device), padded)
    src = last_src[binidx_p]
    live = torch.zeros(total, dtype=torch.bool, device=key.device)
    src[pos] = order
    live[pos] = True
    uv = o.uv[src]
    wp = o.w_plane[src]
    vis = torch.where(live[:, None], o.vis[src], torch.zeros_like(o.vis[src]))
    w = torch.where(live[:, None], o.weights[src], torch.zeros_like(o.weights[src]))
    return synth._copy_with(o, uv, wp, vis, w)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--vis', type=int, default=16 * 1048576)
    ap.add_argument('--dumps', type=int, default=256)
    ap.add_argument('--pixels', type=int, default=4096)
    ap.add_argument('--kernel-width', type=int, default=28)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--raw-only', action='store_true')
    args = ap.parse_args()
    ctx = accel.Context(0)
    q = ctx.create_command_queue()
    dev = ctx.device
    P, K, G = 1, args.kernel_width, args.pixels
    obs = synth.make_observation(G, args.vis, 32, P, device=dev, seed=6)
    ip, gp, ap_ = synth.make_parameters(obs, P, K)
    tg = grid.GridderTemplate(ctx, ip.fixed, gp.fixed, {'variant': 'mfma'})
    td = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed, {'variant': 'mfma'})
    probe = tg.instantiate(q, ap_, ip, gp, 1024)
    shape = probe.slots['grid'].shape
    Gg = shape[1]
    del probe
    gbuf = accel.DeviceArray(ctx, shape, np.complex64)
    wgrid = accel.DeviceArray(ctx, shape, np.float32, tensor=torch.ones(shape, device=dev))
    ref_grid = [None]

    def run(name, o, check=True):
        n = o.n_vis
        g = tg.instantiate(q, ap_, ip, gp, n)
        d = td.instantiate(q, ap_, ip, gp, n)
        uv = accel.DeviceArray(ctx, (n, 4), np.int16, tensor=o.uv)
        wp = accel.DeviceArray(ctx, (n,), np.int16, tensor=o.w_plane)
        vis = accel.DeviceArray(ctx, (n, P), np.complex64, tensor=o.vis.clone())
        wts = accel.DeviceArray(ctx, (n, P), np.float32, tensor=o.weights)
        g.bind(grid=gbuf, weights_grid=wgrid, uv=uv, w_plane=wp, vis=vis)
        d.bind(grid=gbuf, uv=uv, w_plane=wp, vis=vis, weights=wts)
        g.ensure_all_bound()
        d.ensure_all_bound()
        g.num_vis = d.num_vis = n
        torch.cuda.synchronize()
        jf = g.jump_fraction()
        gbuf.zero(q)
        g._run()
        q.finish()
        err = None
        if check:
            if ref_grid[0] is None:
                ref_grid[0] = gbuf.tensor.clone()
            else:
                err = float((gbuf.tensor - ref_grid[0]).abs().max() / ref_grid[0].abs().max())
        res = {}
        for label, op in (('grid', g), ('degrid', d)):
            op._run()
            q.finish()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(q.stream)
            for _ in range(args.reps):
                op._run()
            e1.record(q.stream)
            q.finish()
            res[label] = e0.elapsed_time(e1) / args.reps
        print('%-44s n=%9d jumps=%.4f grid %7.3f ms %7.1f Mrec/s | degrid %7.3f ms %7.1f Mrec/s%s' % (
            name, n, jf, res['grid'], n / res['grid'] / 1e3, res['degrid'], n / res['degrid'] / 1e3,
            '' if err is None else ' | grid vs first %.2e' % err), flush=True)
        return res

    run('raw baseline-major (unmerged)', obs, check=False)
    if args.raw_only:
        k = tile_key(obs, Gg, 5, strips=True, transpose=True)
        run('raw -> strips of 5 columns by v', synth._reordered(obs, torch.argsort(k, stable=True)))
        k = tile_key(obs, Gg, 5, strips=True)
        run('raw -> strips of 5 rows by u', synth._reordered(obs, torch.argsort(k, stable=True)))
        m, bl = merged_loader_blocks(obs, args.dumps)
        print('merged: %d of %d records' % (m.n_vis, obs.n_vis))
        run('loader blocks, merged (as delivered)', m)
        k = tile_key(m, Gg, 5, strips=True, transpose=True)
        run('  -> strips of 5 columns by v', synth._reordered(m, torch.argsort(k, stable=True)))
        return
    m, bl = merged_loader_blocks(obs, args.dumps)
    print('merged: %d of %d records' % (m.n_vis, obs.n_vis))
    run('loader blocks, merged (as delivered)', m)
    order = torch.argsort(bl, stable=True)
    t = synth._reordered(m, order)
    run('  -> stable sort by baseline', t)
    t2, _ = remerge(t, bl[order])
    run('  -> ... + re-merge across block seams', t2)
    for binsz in (5, 3, 8, 16):
        k = tile_key(m, Gg, binsz)
        run('  -> tile sort, bins %d x %d serpentine' % (binsz, binsz),
            synth._reordered(m, torch.argsort(k, stable=True)))
    k = tile_key(m, Gg, 5, strips=True)
    run('  -> strips of 5 rows, by u (serpentine)', synth._reordered(m, torch.argsort(k, stable=True)))
    for width in (5, 4, 3):
        k = tile_key(m, Gg, width, strips=True, transpose=True)
        run('  -> strips of %d COLUMNS, by v (serpentine)' % width,
            synth._reordered(m, torch.argsort(k, stable=True)))
    k = tile_key(m, Gg, 5)
    run('  -> tile sort 5 x 5, bins padded to 8', pad_bins(m, k), check=False)
    run('  -> tile sort 5 x 5, bins padded to 4', pad_bins(m, k, 4), check=False)
    # big tiles, track order inside
    for big in (64, 128):
        kb = tile_key(m, Gg, big) * 4096 + bl
        run('  -> tiles %d^2, by baseline inside' % big, synth._reordered(m, torch.argsort(kb, stable=True)))
    del m, t, t2
    ref_grid[0] = None
    for name, fn in (('time-major', synth.order_time_major), ('shuffled', synth.order_shuffled)):
        o = fn(obs)['obs']
        k = tile_key(o, Gg, 5)
        run('%s raw -> tile sort 5 x 5 (once)' % name, synth._reordered(o, torch.argsort(k, stable=True)))
        run('%s raw -> tile sort 5 x 5, padded to 8' % name, pad_bins(o, k), check=False)
        k = tile_key(o, Gg, 5, strips=True, transpose=True)
        run('%s raw -> strips of 5 columns by v' % name, synth._reordered(o, torch.argsort(k, stable=True)))
        del o


if __name__ == '__main__':
    main()
