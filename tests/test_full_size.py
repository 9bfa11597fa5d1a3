"""Full-size (BASELINE.json configs 2 and 4) checks of the HIP gridder through
size-independent properties, plus an oracle comparison on a 1 M-visibility chunk.
Run with ``-m gpu``; needs ~3 GB of HBM."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

from helpers import context_queue, relerr          # noqa: E402
from oracle import kimg_oracle as orc               # noqa: E402

pytestmark = pytest.mark.gpu


def _setup(pixels, n_vis, w_planes, P, K=28, variant='auto', vis_block=1048576, arith='fp32'):
    import torch
    import synth
    from katsdpimager_amd import accel, grid
    ctx, q = context_queue()
    obs = synth.make_observation(pixels, n_vis, w_planes, P, device=ctx.device)
    ip, gp, ap = synth.make_parameters(obs, P, K)
    fn = grid.GridderTemplate(ctx, ip.fixed, gp.fixed, {'variant': variant, 'arith': arith}) \
        .instantiate(q, ap, ip, gp, vis_block)
    Gg = fn.slots['grid'].shape[1]
    gen = torch.Generator(device=ctx.device)
    gen.manual_seed(2)
    wg = accel.DeviceArray(ctx, (P, Gg, Gg), np.float32,
                           tensor=torch.rand((P, Gg, Gg), generator=gen, device=ctx.device))
    fn.bind(weights_grid=wg)
    fn.ensure_all_bound()
    torch.cuda.synchronize()        # torch generated the inputs on its own stream
    return ctx, q, obs, fn, wg


def _grid_all(ctx, q, obs, fn, vis=None, zero=True):
    """Grid every chunk of the observation (optionally with substitute visibilities)."""
    import torch
    from katsdpimager_amd import accel
    vb = fn.max_vis
    P = obs.vis.shape[1]
    vis = obs.vis if vis is None else vis
    torch.cuda.synchronize()        # inputs may have been produced on torch's stream
    if zero:
        fn.buffer('grid').zero(q)
    for start in range(0, obs.n_vis, vb):
        n = min(vb, obs.n_vis - start)
        if n == vb:
            fn.bind(uv=accel.DeviceArray(ctx, (vb, 4), np.int16, tensor=obs.uv[start:start + vb]),
                    w_plane=accel.DeviceArray(ctx, (vb,), np.int16,
                                              tensor=obs.w_plane[start:start + vb]),
                    vis=accel.DeviceArray(ctx, (vb, P), np.complex64, tensor=vis[start:start + vb]))
        else:       # ragged last chunk: copy into a full-size staging buffer
            import torch
            uv = torch.zeros((vb, 4), dtype=torch.int16, device=ctx.device)
            wp = torch.zeros((vb,), dtype=torch.int16, device=ctx.device)
            vs = torch.zeros((vb, P), dtype=torch.complex64, device=ctx.device)
            uv[:n], wp[:n], vs[:n] = obs.uv[start:], obs.w_plane[start:], vis[start:]
            fn.bind(uv=accel.DeviceArray(ctx, (vb, 4), np.int16, tensor=uv),
                    w_plane=accel.DeviceArray(ctx, (vb,), np.int16, tensor=wp),
                    vis=accel.DeviceArray(ctx, (vb, P), np.complex64, tensor=vs))
        fn.num_vis = n
        fn()
        if n != vb:
            q.finish()      # the staging tensors above must outlive the launch
    # the operator runs on its own stream; torch ops below use the default stream
    q.finish()
    return fn.buffer('grid').tensor


def _expected_checksum(obs, fn, wg, vis=None):
    """sum over all grid cells = sum_r s_r * conj(sum_j kv[j]) * conj(sum_k ku[k]), in float64
    on the host (linear in N, independent of the grid size)."""
    kernel = fn.convolve_kernel.data.astype(np.complex128)
    rowsum = kernel.sum(axis=2)                                     # [W][OV]
    uv = obs.uv.cpu().numpy().astype(np.int64)
    wp = obs.w_plane.cpu().numpy().astype(np.int64)
    vis = (obs.vis if vis is None else vis).cpu().numpy().astype(np.complex128)
    wgrid = wg.tensor.cpu().numpy()
    Gg = wgrid.shape[-1]
    out = []
    for p in range(vis.shape[1]):
        w = wgrid[p][uv[:, 1] + Gg // 2, uv[:, 0] + Gg // 2].astype(np.float64)
        s = vis[:, p] * w
        out.append(np.sum(s * np.conj(rowsum[wp, uv[:, 3]]) * np.conj(rowsum[wp, uv[:, 2]])))
    return np.array(out)


def test_c2_checksum_and_linearity():
    """Config 2 geometry (4096^2, 32 planes, K=28, P=1), 6.3 M visibilities with a ragged last
    chunk: checksum of the whole grid vs float64; linearity grid(a x + b y) = a grid(x) + b grid(y)."""
    import torch
    ctx, q, obs, fn, wg = _setup(4096, 6_300_000, 32, 1)
    g1 = _grid_all(ctx, q, obs, fn).clone()
    got = np.array([complex(g1[p].sum(dtype=torch.complex128)) for p in range(1)])
    want = _expected_checksum(obs, fn, wg)
    assert np.all(np.abs(got - want) <= 2e-5 * np.abs(want).max())
    gen = torch.Generator(device=ctx.device)
    gen.manual_seed(9)
    y = torch.complex(torch.rand(obs.vis.shape, generator=gen, device=ctx.device) - 0.5,
                      torch.rand(obs.vis.shape, generator=gen, device=ctx.device) - 0.5)
    g2 = _grid_all(ctx, q, obs, fn, vis=y).clone()
    a, b = 0.75 - 0.5j, -1.25 + 2.0j
    g3 = _grid_all(ctx, q, obs, fn, vis=a * obs.vis + b * y)
    err = float((g3 - (a * g1 + b * g2)).abs().max() / g3.abs().max())
    assert err < 1e-5
    # the grid stays inside its allocation: nothing outside the footprint envelope
    assert float(g1.abs().max()) > 0


def test_c2_chunk_vs_oracle():
    """One full 1 048 576-visibility chunk of the config-2 workload against the oracle."""
    ctx, q, obs, fn, wg = _setup(4096, 1_048_576 * 3, 32, 1)
    import torch
    from katsdpimager_amd import accel
    vb = fn.max_vis
    # take the chunk in the middle of the stream (long and short baselines)
    s = slice(vb, 2 * vb)
    fn.buffer('grid').zero(q)
    fn.bind(uv=accel.DeviceArray(ctx, (vb, 4), np.int16, tensor=obs.uv[s]),
            w_plane=accel.DeviceArray(ctx, (vb,), np.int16, tensor=obs.w_plane[s]),
            vis=accel.DeviceArray(ctx, (vb, 1), np.complex64, tensor=obs.vis[s]))
    fn.num_vis = vb
    fn()
    got = fn.buffer('grid').get(q)
    want = np.zeros_like(got)
    uv = obs.uv[s].cpu().numpy()
    orc.grid(fn.convolve_kernel.data, want, wg.tensor.cpu().numpy(),
             np.ascontiguousarray(uv[:, :2]), np.ascontiguousarray(uv[:, 2:]),
             obs.w_plane[s].cpu().numpy(), obs.vis[s].cpu().numpy())
    assert relerr(got, want) < 1e-5


def test_c4_full_stokes_checksum_and_chunk_vs_oracle():
    """Config 4 geometry (8192^2, 64 planes, 4 polarizations): the single-row LDS table variant,
    two polarizations per launch.  Checksum of the whole grid vs float64, and ONE 524 288-visibility
    chunk of the workload against the oracle (all four polarizations, every cell) -- for both
    arithmetic forms of the window kernel."""
    import torch
    from katsdpimager_amd import accel
    vb = 524288
    ctx, q, obs, fn, wg = _setup(8192, 3 * vb, 64, 4, vis_block=vb)
    g = _grid_all(ctx, q, obs, fn).clone()
    got = np.array([complex(g[p].sum(dtype=torch.complex128)) for p in range(4)])
    want = _expected_checksum(obs, fn, wg)
    assert np.all(np.abs(got - want) <= 2e-5 * np.abs(want).max())
    del g
    s = slice(vb, 2 * vb)           # the chunk in the middle of the stream
    uv = obs.uv[s].cpu().numpy()
    expected = np.zeros(fn.buffer('grid').shape, np.complex64)
    orc.grid(fn.convolve_kernel.data, expected, wg.tensor.cpu().numpy(),
             np.ascontiguousarray(uv[:, :2]), np.ascontiguousarray(uv[:, 2:]),
             obs.w_plane[s].cpu().numpy(), obs.vis[s].cpu().numpy())
    _, _, _, fn_split, _ = _setup(8192, 3 * vb, 64, 4, vis_block=vb, arith='split_fp16')
    fn_split.bind(weights_grid=wg, grid=fn.buffer('grid'))
    for op in (fn, fn_split):
        op.buffer('grid').zero(q)
        op.bind(uv=accel.DeviceArray(ctx, (vb, 4), np.int16, tensor=obs.uv[s]),
                w_plane=accel.DeviceArray(ctx, (vb,), np.int16, tensor=obs.w_plane[s]),
                vis=accel.DeviceArray(ctx, (vb, 4), np.complex64, tensor=obs.vis[s]))
        op.num_vis = vb
        op()
        assert relerr(op.buffer('grid').get(q), expected) < 1e-5


def _degrid_all(ctx, q, obs, template_args, model_grid, vis_block):
    """vis = 0 - 1 * degrid(model_grid) over every chunk; returns the predicted visibilities."""
    import torch
    from katsdpimager_amd import accel, grid
    ap, ip, gp = template_args
    P = obs.vis.shape[1]
    dg = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed).instantiate(q, ap, ip, gp, vis_block)
    dg.bind(grid=model_grid)
    ones = accel.DeviceArray(ctx, (vis_block, P), np.float32,
                             tensor=torch.ones((vis_block, P), device=ctx.device))
    dg.bind(weights=ones)
    dg.ensure_all_bound()
    n_full = obs.n_vis // vis_block * vis_block          # whole chunks only
    out = torch.zeros((n_full, P), dtype=torch.complex64, device=ctx.device)
    # torch fills on its own stream; the operators run on the queue's stream
    torch.cuda.synchronize()
    for start in range(0, n_full, vis_block):
        sl = slice(start, start + vis_block)
        dg.bind(uv=accel.DeviceArray(ctx, (vis_block, 4), np.int16, tensor=obs.uv[sl]),
                w_plane=accel.DeviceArray(ctx, (vis_block,), np.int16, tensor=obs.w_plane[sl]),
                vis=accel.DeviceArray(ctx, (vis_block, P), np.complex64, tensor=out[sl]))
        dg.num_vis = vis_block
        dg()
    q.finish()
    return -out, n_full


@pytest.mark.parametrize('pixels,w_planes,P,n_vis,vis_block', [
    (4096, 32, 1, 4 * 1048576, 1048576),          # BASELINE config 2 geometry
    (8192, 64, 4, 2 * 524288, 524288)])           # BASELINE config 4 geometry
def test_grid_degrid_adjoint(pixels, w_planes, P, n_vis, vis_block):
    """Gridding (conjugated kernel, grid.py:1049-1050) and degridding (plain kernel,
    grid.py:1150) are adjoint: <degrid(G), v> = <G, grid(v)> with <a, b> = sum conj(a) b, for any
    model grid G and visibilities v.  Both sides at full size, inner products in float64."""
    import torch
    import synth
    from katsdpimager_amd import accel
    ctx, q, obs, fn, wg = _setup(pixels, n_vis, w_planes, P, vis_block=vis_block)
    wg.tensor.fill_(1.0)                                   # no density weights in the identity
    ip, gp, ap = synth.make_parameters(obs, P, 28, degrid=True)
    Gg = fn.slots['grid'].shape[1]
    gen = torch.Generator(device=ctx.device)
    gen.manual_seed(31)
    G = torch.complex(torch.rand((P, Gg, Gg), generator=gen, device=ctx.device) - 0.5,
                      torch.rand((P, Gg, Gg), generator=gen, device=ctx.device) - 0.5)
    model_grid = accel.DeviceArray(ctx, (P, Gg, Gg), np.complex64, tensor=G)
    torch.cuda.synchronize()
    pred, n_full = _degrid_all(ctx, q, obs, (ap, ip, gp), model_grid, vis_block)
    assert n_full == n_vis
    v = obs.vis[:n_full]
    lhs = torch.sum(torch.conj(pred).to(torch.complex128) * v.to(torch.complex128), dim=0)
    gridded = _grid_all(ctx, q, obs, fn)
    rhs = torch.stack([torch.sum(torch.conj(G[p]).to(torch.complex128)
                                 * gridded[p].to(torch.complex128)) for p in range(P)])
    scale = float(torch.sum(pred.abs().to(torch.float64) * v.abs().to(torch.float64)))
    err = float((lhs - rhs).abs().max()) / scale * P
    assert float(pred.abs().max()) > 0 and err < 1e-6


@pytest.mark.parametrize('G,P', [(4096, 1), (2048, 4)])
def test_full_size_noise_estimate_and_clean_rate_invariants(G, P):
    """At BASELINE image sizes: the device noise estimate is numpy's median exactly, and a run of
    minor cycles conserves flux: sum(model) == sum of the logged loop_gain * pixel values == what the
    residual image lost (divided by the PSF volume)."""
    from katsdpimager_amd import clean, parameters
    ctx, q = context_queue()
    rs = np.random.RandomState(G + P)
    img = (0.05 * rs.standard_normal((P, G, G))).astype(np.float32)
    for _ in range(50):
        y, x = rs.randint(200, G - 200, 2)
        img[:, y, x] += rs.uniform(1.0, 3.0)
    fixed = parameters.FixedImageParameters(list(range(P)), np.float32)
    ip = parameters.ImageParameters(fixed, 1.0, None, 0.2, None, pixel_size=1e-5, pixels=G)
    ne = clean.NoiseEstTemplate(ctx, np.float32, P).instantiate(q, (P, G, G), 0.02)
    ne.ensure_all_bound()
    ne.buffer('dirty').set(q, img)
    bp = ne.border_pixels
    want = np.median(np.abs(img[:, bp:G - bp, bp:G - bp])) * np.float32(clean._MEDIAN_TO_RMS)
    assert ne() == np.float32(want)

    cp = parameters.CleanParameters(1000, 0.1, 0.85, 5.0, 1 if P > 1 else 0, 0.01, 0.5, 0.02)
    fn = clean.CleanTemplate(ctx, cp, np.float32, P).instantiate(q, ip)
    fn.ensure_all_bound()
    g1 = np.exp(-0.5 * ((np.arange(G) - G // 2) / 3.0) ** 2).astype(np.float32)
    psf = np.repeat(np.outer(g1, g1)[None], P, axis=0)
    fn.buffer('dirty').set(q, img)
    fn.buffer('psf').set(q, psf)
    fn.buffer('model').zero(q)
    fn.reset()
    log = fn.run_cycles((P, 65, 65), 0.0, 1000)
    assert len(log) == 1000
    model = fn.buffer('model').get(q)
    total = np.sum([e[2] for e in log], axis=0, dtype=np.float64)
    np.testing.assert_allclose(model.sum(axis=(1, 2), dtype=np.float64), total, rtol=1e-5)
    # the residual really lost what the model gained (PSF of unit peak, everything inside the patch)
    resid = fn.buffer('dirty').get(q)
    np.testing.assert_allclose((img.sum(axis=(1, 2), dtype=np.float64)
                                - resid.sum(axis=(1, 2), dtype=np.float64)),
                               total * psf[0, G // 2 - 32:G // 2 + 33, G // 2 - 32:G // 2 + 33].sum(dtype=np.float64),
                               rtol=1e-3)


def test_c2_forms_repeatable_at_scale():
    """A/B harness at config-2 scale (the round-1 packed-math variant differed by 7e-3 of
    the peak between two runs at this scale while every small test passed): 8.4 M visibilities,
    each arithmetic form gridded three times.  The only run-to-run difference allowed is the order
    of the float atomics (5e-7 of the peak); the split form stays within 2e-6 of the float32 form;
    both touch exactly the same cells."""
    n = 8 << 20
    ctx, q, obs, fn, wg = _setup(4096, n, 32, 1)
    _, _, _, fn_split, _ = _setup(4096, n, 32, 1, arith='split_fp16')
    fn_split.bind(weights_grid=wg)
    exact = [_grid_all(ctx, q, obs, fn).clone() for _ in range(3)]
    split = [_grid_all(ctx, q, obs, fn_split).clone() for _ in range(3)]
    peak = float(exact[0].abs().max())
    for runs in (exact, split):
        for other in runs[1:]:
            assert float((runs[0] - other).abs().max()) <= 5e-7 * peak
    assert float((split[0] - exact[0]).abs().max()) <= 2e-6 * peak
    assert int((split[0] != 0).sum()) == int((exact[0] != 0).sum())


def test_c2_true_length_one_launch():
    """BASELINE config 2 at its TRUE length -- the launch bench.py times: 50 M visibilities of one
    W-slice in ONE gridder launch (4096^2, 32 planes, K = 28, one polarization).  (i) The float64
    checksum of the whole grid (linear in N; evaluated with torch in float64 on the device,
    independently of the kernel); (ii) the same cells, values within 4e-6 of the peak, as
    `vis_block` launches of 1 M; (iii) the 1 M chunk in the MIDDLE of the 50 M against the oracle."""
    import torch
    from katsdpimager_amd import accel
    n = 50_000_000
    ctx, q, obs, fn_chunked, wg = _setup(4096, n, 32, 1)
    _, _, _, fn, _ = _setup(4096, n, 32, 1, vis_block=n, variant='mfma')
    fn.bind(weights_grid=wg)
    got = _grid_all(ctx, q, obs, fn).clone()                    # one launch
    assert fn.last_variant == 'mfma'
    # (i)
    dev = ctx.device
    rowsum = torch.from_numpy(fn.convolve_kernel.data.astype(np.complex128).sum(axis=2)).to(dev)
    Gg = wg.shape[-1]
    uv, wp = obs.uv.long(), obs.w_plane.long()
    w = wg.tensor[0][uv[:, 1] + Gg // 2, uv[:, 0] + Gg // 2].double()
    want_sum = complex((obs.vis[:, 0].to(torch.complex128) * w * torch.conj(rowsum[wp, uv[:, 3]])
                        * torch.conj(rowsum[wp, uv[:, 2]])).sum())
    got_sum = complex(got[0].sum(dtype=torch.complex128))
    assert abs(got_sum - want_sum) <= 2e-5 * abs(want_sum), (got_sum, want_sum)
    del uv, wp, w
    # (ii)
    want = _grid_all(ctx, q, obs, fn_chunked)
    peak = float(want.abs().max())
    assert float((got - want).abs().max()) <= 4e-6 * peak
    assert int((got != 0).sum()) == int((want != 0).sum())
    del got, want
    # (iii)
    vb = fn_chunked.max_vis
    s = slice(24 * vb, 25 * vb)
    fn_chunked.buffer('grid').zero(q)
    fn_chunked.bind(uv=accel.DeviceArray(ctx, (vb, 4), np.int16, tensor=obs.uv[s]),
                    w_plane=accel.DeviceArray(ctx, (vb,), np.int16, tensor=obs.w_plane[s]),
                    vis=accel.DeviceArray(ctx, (vb, 1), np.complex64, tensor=obs.vis[s]))
    fn_chunked.num_vis = vb
    fn_chunked()
    mid = fn_chunked.buffer('grid').get(q)
    ref = np.zeros_like(mid)
    uvh = obs.uv[s].cpu().numpy()
    orc.grid(fn_chunked.convolve_kernel.data, ref, wg.tensor.cpu().numpy(),
             np.ascontiguousarray(uvh[:, :2]), np.ascontiguousarray(uvh[:, 2:]),
             obs.w_plane[s].cpu().numpy(), obs.vis[s].cpu().numpy())
    assert relerr(mid, ref) < 1e-5


@pytest.mark.parametrize('K,W', [(28, 32), (60, 256)])
@pytest.mark.parametrize('arith', ['fp32', 'split_fp16'])
def test_long_launch_work_by_the_chunk(arith, K, W):
    """A launch long enough for the window kernels to hand out their work by the chunk (>= 2 x 1024
    visibilities per wave: 8.4 M here): (a) one launch with the chunk counter in the workspace, (b)
    the same launch without a workspace (chunks taken round-robin; the C ABI accepts NULL when the
    table fits LDS), (c) 1 M-visibility launches (one contiguous range per wave), gridder and
    degridder.  Same cells, values within float32 summation-order rounding."""
    import torch
    from katsdpimager_amd import accel, grid
    from katsdpimager_amd._lib import lib, check
    n = 8 << 20
    # (K = 60 with 256 W planes: four tap-block launches, tables in HBM -- the other kernel forms)
    ctx, q, obs, fn_chunked, wg = _setup(4096, n, W, 1, K=K, arith=arith)       # (c) vis_block 1 M
    _, _, _, fn, _ = _setup(4096, n, W, 1, K=K, vis_block=n, arith=arith, variant='mfma')
    fn.bind(weights_grid=wg)
    want = _grid_all(ctx, q, obs, fn_chunked).clone()
    got_queue = _grid_all(ctx, q, obs, fn).clone()                              # (a)
    peak = float(want.abs().max())
    # (another split of the stream over the waves is another order of summation per cell: float32
    # rounding of sums of ~10^3 terms, not just the order of the atomics)
    tol = (4e-6 if arith == 'fp32' else 6e-6) * peak
    assert float((got_queue - want).abs().max()) <= tol
    assert int((got_queue != 0).sum()) == int((want != 0).sum())
    # (b) straight through the C ABI, no workspace
    g = fn.buffer('grid')
    if fn._workspace_bytes == 256:
        g.zero(q)
        P, G = g.shape[0], g.shape[1]
        table, t_w, t_ov, t_k = fn._kernel_args()
        rc = lib().kimg_grid(g.ptr, G, G * G, G, P, wg.ptr, G, G * G, fn.buffer('uv').ptr,
                             fn.buffer('w_plane').ptr, fn.buffer('vis').ptr, n, table, t_w, t_ov, t_k,
                             None, 0, grid.GRID_VARIANTS['mfma'], grid.GRID_ARITH[arith], q.handle)
        check(rc, 'kimg_grid')
        q.finish()
        got_static = g.tensor.clone()
        assert float((got_static - want).abs().max()) <= tol
        assert int((got_static != 0).sum()) == int((want != 0).sum())
    # degridder: one long launch against 1 M-visibility launches (the order of a visibility's sum
    # follows the window's position, so the two differ by float32 rounding)
    import synth
    ip, gp, ap = synth.make_parameters(obs, 1, K)
    model = accel.DeviceArray(ctx, g.shape, np.complex64, tensor=(want / peak).contiguous())
    torch.cuda.synchronize()
    res = {}
    for vb in (n, 1 << 20):
        dg = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed, {'arith': arith, 'variant': 'mfma'}) \
            .instantiate(q, ap, ip, gp, vb)
        out = torch.zeros((n, 1), dtype=torch.complex64, device=ctx.device)
        ones = accel.DeviceArray(ctx, (vb, 1), np.float32, tensor=torch.ones((vb, 1), device=ctx.device))
        dg.bind(grid=model, weights=ones)
        dg.ensure_all_bound()
        torch.cuda.synchronize()
        for start in range(0, n, vb):
            sl = slice(start, start + vb)
            dg.bind(uv=accel.DeviceArray(ctx, (vb, 4), np.int16, tensor=obs.uv[sl]),
                    w_plane=accel.DeviceArray(ctx, (vb,), np.int16, tensor=obs.w_plane[sl]),
                    vis=accel.DeviceArray(ctx, (vb, 1), np.complex64, tensor=out[sl]))
            dg.num_vis = vb
            dg()
        q.finish()
        res[vb] = out
    scale = float(res[n].abs().max())
    assert scale > 0
    assert float((res[n] - res[1 << 20]).abs().max()) <= (2e-6 if arith == 'fp32' else 4e-6) * scale


def test_long_launch_ragged_length():
    """The chunked long launch with a length that is no multiple of 64, of the chunk or of anything
    else (last chunk and last batch partial), against 1 M-visibility launches."""
    n = 7_000_003
    ctx, q, obs, fn_chunked, wg = _setup(4096, n, 32, 1)
    _, _, _, fn, _ = _setup(4096, n, 32, 1, vis_block=n, variant='mfma')
    fn.bind(weights_grid=wg)
    want = _grid_all(ctx, q, obs, fn_chunked).clone()
    got = _grid_all(ctx, q, obs, fn).clone()
    peak = float(want.abs().max())
    assert float((got - want).abs().max()) <= 4e-6 * peak
    assert int((got != 0).sum()) == int((want != 0).sum())
    # every visibility was gridded exactly once: the sums of all cells agree
    a, b = got.to(__import__('torch').complex128).sum(), want.to(__import__('torch').complex128).sum()
    assert abs(a - b) <= 1e-6 * abs(b)


# ---- BASELINE config 5: the major-cycle loop at full size -------------------------------------
def _c5_inputs(n_in=12_000_000, n_sources=150):
    import torch
    import synth
    from katsdpimager_amd import accel, preprocess
    ctx, q = context_queue()
    obs = synth.make_observation(4096, n_in, 32, 1, device=ctx.device)
    pos, flux = synth.add_point_sources(obs, n_sources, seed=4, noise=0.02)
    ipd, gpd, apd = synth.make_parameters(obs, 1, 28, degrid=True)
    d_uvw = accel.DeviceArray(ctx, (n_in, 3), np.float32, tensor=obs.uvw)
    d_wts = accel.DeviceArray(ctx, (1, n_in, 1), np.float32, tensor=obs.weights[None].contiguous())
    d_vis = accel.DeviceArray(ctx, (1, n_in, 1), np.complex64, tensor=obs.raw_vis[None].contiguous())
    torch.cuda.synchronize()
    coll = preprocess.VisibilityCollectorDevice(q, [ipd], [gpd], 1 << 20)
    coll.add(d_uvw, d_wts, d_vis, None, None, np.identity(1, np.complex64), None)
    coll.close()
    return ctx, q, obs, (ipd, gpd, apd), coll.reader(), pos, flux


def test_c5_major_cycle_loop_full_size():
    """BASELINE config 5 through the product's driver: device preprocessing -> HBM-resident store
    (>= 5 M stored visibilities) -> frontend.process_channel at 4096^2, 32 planes, K = 28, robust
    weights, degridding, 2 major cycles x 1000 minor cycles (reference loop: frontend.py:543-585).
    * the same components whether a W-slice is gridded in one launch or in vis_block chunks on one
      or two streams, and whether the minor cycles run batched on the device or one host round trip
      per cycle;
    * flux conservation: the model image holds exactly the logged components;
    * the residual visibilities left in the chunk buffer equal stored - weights * degrid(model
      grid) evaluated by the oracle on a 100 K sample;
    * the brightest components sit on the simulated sources."""
    import torch
    from katsdpimager_amd import frontend, imaging, parameters, weight
    ctx, q, obs, (ipd, gpd, apd), reader, src_pos, src_flux = _c5_inputs()
    stored = reader.len(0, 0)
    assert reader.num_w_slices(0) == 1 and stored >= 5_000_000
    cp = parameters.CleanParameters(1000, 0.1, 0.85, 5.0, 0, 0.01, 0.5, 0.02)
    wparm = parameters.WeightParameters(weight.WeightType.ROBUST, 0.0)
    template = imaging.ImagingTemplate(ctx, apd, ipd.fixed, wparm, gpd.fixed, cp)

    def run(block, streams, batched):
        im = template.instantiate(q, ipd, gpd, block, 0, 2, streams=streams)
        im.ensure_all_bound()
        stats = frontend.process_channel(reader, 0, im, ipd, gpd, cp, wparm.weight_type, block, 2,
                                         True, batched_clean=batched)
        q.finish()
        return im, stats

    im_a, st_a = run(stored, 1, True)           # one launch per slice (the driver's default)
    assert st_a['major'] == 2 and st_a['minor'] == 2 * 999     # counted as frontend.py:577-582 does
    comps_a = {k: float(v[0]) for k, v in im_a._model_components.items()}
    model = im_a.get_buffer('model')
    # flux conservation: model image == logged components, nothing else
    assert int(np.count_nonzero(model)) == len(comps_a)
    total = sum(comps_a.values())
    assert abs(float(model.sum(dtype=np.float64)) - total) <= 1e-5 * abs(total)
    for (y, x), f in list(comps_a.items())[:50]:
        assert abs(model[0, y, x] - f) <= 1e-5 * abs(f) + 1e-7
    # the strongest components are the simulated sources (pixel-centred, so exactly there)
    src = {(int(y), int(x)) for y, x in src_pos}
    brightest = sorted(comps_a, key=lambda k: -abs(comps_a[k]))[:20]
    assert sum(k in src for k in brightest) >= 18

    # residual visibilities vs the oracle's degridder (a 100 K sample of the slice)
    rs = np.random.RandomState(5)
    starts = np.sort(rs.randint(0, stored - 1000, 100))
    rows = (starts[:, None] + np.arange(1000)[None, :]).reshape(-1)
    sel = torch.from_numpy(rows).to(ctx.device)
    chunk = next(iter(reader.iter_slice_device(0, 0, stored)))
    uv = chunk.uv.tensor[sel].cpu().numpy()
    want = chunk.vis.tensor[sel].cpu().numpy().copy()
    orc.degrid(im_a._predict.convolve_kernel.data, im_a.get_buffer('degrid'),
               np.ascontiguousarray(uv[:, :2]), np.ascontiguousarray(uv[:, 2:]),
               chunk.w_plane.tensor[sel].cpu().numpy(), chunk.weights.tensor[sel].cpu().numpy(), want)
    got = im_a.buffer('vis').tensor[sel].cpu().numpy()
    scale = np.abs(chunk.vis.tensor[sel].cpu().numpy()).max()
    assert np.abs(got - want).max() <= 2e-5 * scale
    # ... and the model explains part of the data (1000 cycles at loop gain 0.1 over 150 sources
    # remove about half of the flux): the residuals are smaller than the visibilities
    assert np.abs(got).mean() < 0.8 * np.abs(chunk.vis.tensor[sel].cpu().numpy()).mean()

    resid_a = im_a.get_buffer('dirty')
    del im_a
    for block, streams, batched in ((1 << 20, 2, True), (1 << 20, 1, False)):
        im_b, st_b = run(block, streams, batched)
        assert st_b['minor'] == st_a['minor'] and st_b['psf_patch'] == st_a['psf_patch']
        np.testing.assert_allclose(st_b['peaks'], st_a['peaks'], rtol=1e-5)
        comps_b = {k: float(v[0]) for k, v in im_b._model_components.items()}
        # the gridders' float atomics land in a different order, so the images differ in the last
        # bits and a near-tie between faint components may resolve differently; the bright ones
        # may not move
        common = set(comps_a) & set(comps_b)
        assert len(common) >= 0.98 * max(len(comps_a), len(comps_b))
        big = max(abs(v) for v in comps_a.values())
        for k in common:
            assert abs(comps_a[k] - comps_b[k]) <= 2e-4 * big
        assert abs(sum(comps_b.values()) - total) <= 2e-4 * abs(total)
        # residual images: against the dirty image's peak, away from the corners where the division
        # by the taper amplifies the last-bit differences of the grids (helpers.tapered_relerr)
        inner = np.s_[:, 512:-512, 512:-512]
        assert np.abs(im_b.get_buffer('dirty')[inner] - resid_a[inner]).max() <= 2e-4 * st_a['peaks'][0]
        del im_b


def test_c5_dirty_image_of_a_sub_band_vs_oracle():
    """frontend.make_weights + make_dirty at 4096^2 / 32 planes on a thinned copy of the config-5
    visibilities (~150 K stored records, natural weights) against the oracle's gridder + inverse
    FFT + taper on the same stored records: the full-size dirty image itself, not a property."""
    import torch
    from katsdpimager_amd import accel, frontend, imaging, parameters, preprocess, weight
    from helpers import tapered_relerr
    import synth
    ctx, q = context_queue()
    obs = synth.make_observation(4096, 12_000_000, 32, 1, device=ctx.device)
    synth.add_point_sources(obs, 60, seed=4, noise=0.02)
    ipd, gpd, apd = synth.make_parameters(obs, 1, 28, degrid=True)
    pick = torch.arange(0, obs.n_vis, 80, device=ctx.device)
    n = int(pick.shape[0])
    d_uvw = accel.DeviceArray(ctx, (n, 3), np.float32, tensor=obs.uvw[pick].contiguous())
    d_wts = accel.DeviceArray(ctx, (1, n, 1), np.float32, tensor=obs.weights[pick][None].contiguous())
    d_vis = accel.DeviceArray(ctx, (1, n, 1), np.complex64, tensor=obs.raw_vis[pick][None].contiguous())
    torch.cuda.synchronize()
    coll = preprocess.VisibilityCollectorDevice(q, [ipd], [gpd], 1 << 18)
    coll.add(d_uvw, d_wts, d_vis, None, None, np.identity(1, np.complex64), None)
    coll.close()
    reader = coll.reader()
    cp = parameters.CleanParameters(100, 0.1, 0.85, 5.0, 0, 0.01, 0.5, 0.02)
    wparm = parameters.WeightParameters(weight.WeightType.NATURAL, 0.0)
    template = imaging.ImagingTemplate(ctx, apd, ipd.fixed, wparm, gpd.fixed, cp)
    block = reader.len(0, 0)
    im = template.instantiate(q, ipd, gpd, block, 0, 1)
    im.ensure_all_bound()
    frontend.make_weights(reader, 0, im, wparm.weight_type, block)
    mid_w = frontend.slice_mid_w(ipd, gpd)
    frontend.make_dirty(reader, 0, 'vis', im, mid_w, block, True)
    got = im.get_buffer('dirty')
    rec = next(iter(reader.iter_slice(0, 0, block)))
    Gg = im.buffer('grid').shape[1]
    grid_ = np.zeros((1, Gg, Gg), np.complex64)
    kernel = im._gridder.convolve_kernel
    orc.grid(kernel.data, grid_, np.ones((1, Gg, Gg), np.float32), np.ascontiguousarray(rec.uv),
             np.ascontiguousarray(rec.sub_uv), np.ascontiguousarray(rec.w_plane),
             np.ascontiguousarray(rec.vis))
    assert relerr(im.get_buffer('grid'), grid_) < 1e-5
    G = 4096
    full = np.zeros((1, G, G), np.complex64)
    lo = (G - Gg) // 2
    full[:, lo:lo + Gg, lo:lo + Gg] = grid_
    want = np.zeros((1, G, G), np.float32)
    k1d = kernel.taper(G).astype(np.float32)
    orc.grid_to_image(full, want, k1d, float(ipd.pixel_size), -0.5 * G * float(ipd.pixel_size),
                      float(mid_w[0]))
    assert tapered_relerr(got, want, k1d) < 1e-5
    inner = np.s_[:, G // 8:-G // 8, G // 8:-G // 8]
    assert relerr(got[inner], want[inner]) < 1e-4


def test_c1_true_size_two_w_slices_vs_oracle():
    """BASELINE config 1 at its TRUE size -- 1024^2 image, 8 W-planes, Stokes I, here with two W
    slices so that the second one is imaged at w != 0 -- one 1 M-visibility block of a simulated
    64-antenna track set: device preprocessing -> resident store (re-ordered, unmerged) ->
    make_weights + make_dirty over both slices, against the oracle on the same stored records:
    `orc.grid` per slice (1e-5), then `orc.grid_to_image` of both slices accumulated (1e-5
    taper-weighted, 1e-4 on the inner image)."""
    import torch
    from katsdpimager_amd import accel, frontend, imaging, parameters, preprocess, weight
    from helpers import tapered_relerr
    import synth
    ctx, q = context_queue()
    G = 1024
    obs = synth.make_observation(G, 1_048_576, 8, 1, device=ctx.device, w_slices=2, seed=9)
    synth.add_point_sources(obs, 25, seed=3, noise=0.05)
    ipd, gpd, apd = synth.make_parameters(obs, 1, 28, degrid=True)
    assert gpd.w_slices == 2 and gpd.w_planes == 8
    n = obs.n_vis
    d_uvw = accel.DeviceArray(ctx, (n, 3), np.float32, tensor=obs.uvw)
    d_wts = accel.DeviceArray(ctx, (1, n, 1), np.float32, tensor=obs.weights[None].contiguous())
    d_vis = accel.DeviceArray(ctx, (1, n, 1), np.complex64, tensor=obs.raw_vis[None].contiguous())
    torch.cuda.synchronize()
    coll = preprocess.VisibilityCollectorDevice(q, [ipd], [gpd], 1 << 20, merge=False)
    coll.add(d_uvw, d_wts, d_vis, None, None, np.identity(1, np.complex64), None)
    coll.close()
    reader = coll.reader()
    lens = [reader.len(0, s_) for s_ in range(2)]
    assert min(lens) > 5000 and sum(lens) == coll.num_output       # both slices populated
    cp = parameters.CleanParameters(100, 0.1, 0.85, 5.0, 0, 0.01, 0.5, 0.02)
    wparm = parameters.WeightParameters(weight.WeightType.NATURAL, 0.0)
    template = imaging.ImagingTemplate(ctx, apd, ipd.fixed, wparm, gpd.fixed, cp)
    block = max(lens)
    im = template.instantiate(q, ipd, gpd, block, 0, 1)
    im.ensure_all_bound()
    frontend.make_weights(reader, 0, im, wparm.weight_type, block)
    mid_w = frontend.slice_mid_w(ipd, gpd)
    assert mid_w[0] == 0.0 and mid_w[1] > 0.0
    frontend.make_dirty(reader, 0, 'vis', im, mid_w, block, True)
    got = im.get_buffer('dirty')
    Gg = im.buffer('grid').shape[1]
    kernel = im._gridder.convolve_kernel
    k1d = kernel.taper(G).astype(np.float32)
    want = np.zeros((1, G, G), np.float32)
    lo = (G - Gg) // 2
    for s_ in range(2):
        rec = next(iter(reader.iter_slice(0, s_, block)))
        grid_ = np.zeros((1, Gg, Gg), np.complex64)
        orc.grid(kernel.data, grid_, np.ones((1, Gg, Gg), np.float32), np.ascontiguousarray(rec.uv),
                 np.ascontiguousarray(rec.sub_uv), np.ascontiguousarray(rec.w_plane),
                 np.ascontiguousarray(rec.vis))
        if s_ == 1:
            # (the imager's grid buffer holds the last slice gridded)
            assert relerr(im.get_buffer('grid'), grid_) < 1e-5
        full = np.zeros((1, G, G), np.complex64)
        full[:, lo:lo + Gg, lo:lo + Gg] = grid_
        orc.grid_to_image(full, want, k1d, float(ipd.pixel_size), -0.5 * G * float(ipd.pixel_size),
                          float(mid_w[s_]))
    assert tapered_relerr(got, want, k1d) < 1e-5
    inner = np.s_[:, G // 8:-G // 8, G // 8:-G // 8]
    assert relerr(got[inner], want[inner]) < 1e-4


def test_c2_orders_binned_vs_track_order():
    """Config 2 geometry, 4 M visibilities: the same visibilities in track order (window kernel as
    is), time-major and shuffled (both binned by `auto`) give the same grid; the loader-shaped,
    adjacent-merged stream (tools/synth.order_loader_blocks) stays on the window kernel and gives
    the same grid too (merging only reorders the float sums)."""
    import torch
    import synth
    from katsdpimager_amd import accel
    n = 4 << 20
    ctx, q, obs, fn, wg = _setup(4096, n, 32, 1, vis_block=n)

    def run(o):
        m = o.n_vis
        pad = n - m
        z = lambda t: torch.cat([t, torch.zeros((pad,) + tuple(t.shape[1:]), dtype=t.dtype,
                                                device=t.device)]) if pad else t
        fn.bind(uv=accel.DeviceArray(ctx, (n, 4), np.int16, tensor=z(o.uv)),
                w_plane=accel.DeviceArray(ctx, (n,), np.int16, tensor=z(o.w_plane)),
                vis=accel.DeviceArray(ctx, (n, 1), np.complex64, tensor=z(o.vis)))
        fn.num_vis = m
        torch.cuda.synchronize()
        fn.buffer('grid').zero(q)
        fn()
        q.finish()
        return fn.buffer('grid').tensor.clone(), fn.last_variant

    ref, v0 = run(obs)
    assert v0 == 'mfma'
    peak = float(ref.abs().max())
    from katsdpimager_amd import preprocess
    for order, want_variant in ((synth.order_time_major, 'binned'), (synth.order_shuffled, 'binned'),
                                (synth.order_loader_blocks, 'mfma')):
        o = order(obs)['obs']
        fn.locality_hint = None
        got, variant = run(o)
        assert variant == want_variant, order.__name__
        assert float((got - ref).abs().max()) <= 1e-5 * peak, order.__name__
        # ... and after the resident store's once-per-channel re-order (with and without the
        # whole-slice merge): the window kernel as it is, the same grid
        m = o.n_vis
        arrays = dict(uv=accel.DeviceArray(ctx, (m, 4), np.int16, tensor=o.uv),
                      w_plane=accel.DeviceArray(ctx, (m,), np.int16, tensor=o.w_plane),
                      weights=accel.DeviceArray(ctx, (m, 1), np.float32, tensor=o.weights),
                      vis=accel.DeviceArray(ctx, (m, 1), np.complex64, tensor=o.vis))
        torch.cuda.synchronize()
        for merge in (False, True):
            out, kept = preprocess.reorder_device_arrays(q, 1, m, arrays, 28, obs.oversample,
                                                         obs.w_planes, merge)
            assert (kept == m) if not merge else (0 < kept <= m)
            q.finish()
            so = synth._copy_with(o, out['uv'].tensor[:kept], out['w_plane'].tensor[:kept],
                                  out['vis'].tensor[:kept], out['weights'].tensor[:kept])
            fn.locality_hint = True
            got, variant = run(so)
            assert variant == 'mfma'
            assert float((got - ref).abs().max()) <= 1e-5 * peak, (order.__name__, merge)
    fn.locality_hint = None


# ---- BASELINE config 3: eight spectral-line channels, one per GPU -- as far as one GPU goes --------
def test_c3_eight_channels_through_the_sharded_driver():
    """Config 3 (8 channels, 4096^2, 32 W-planes, K = 28) through `parallel.image_assigned_channels`, the
    driver a rank of the 8-GPU job runs (channel c on rank c mod world size; frontend.py:749-767 loops
    channels serially).  With one GPU the world has one rank and it owns all eight channels: they
    are imaged with four in flight (one host thread and one HIP stream per worker, jobs and imagers
    made lazily) and every channel's result is held against the same channel imaged on its own, one
    after the other -- components at the same pixels, fluxes and residual images to the float32
    tolerance that the gridders' atomics leave.  The channels differ (frequency scaling of uvw,
    their own sources), so a mix-up between streams would show."""
    import torch
    import synth
    from katsdpimager_amd import accel, frontend, imaging, parallel, parameters, preprocess, weight
    ctx, q = context_queue()
    channels, n_in = 8, 1_500_000
    cp = parameters.CleanParameters(200, 0.1, 0.85, 5.0, 0, 0.01, 0.5, 0.02)
    wparm = parameters.WeightParameters(weight.WeightType.ROBUST, 0.0)
    stores, params = [], []
    for c in range(channels):
        obs = synth.make_observation(4096, n_in, 32, 1, device=ctx.device, seed=20 + c,
                                     channel_scale=parallel.channel_frequency_scale(c, channels))
        synth.add_point_sources(obs, 40, seed=100 + c, noise=0.02)
        ipd, gpd, apd = synth.make_parameters(obs, 1, 28, degrid=True)
        # (the inputs stay alive until the collector has read them: its kernels run on the queue's
        # stream, and memory handed back to torch's allocator before they are done could be given
        # out again under them)
        d_uvw = accel.DeviceArray(ctx, (n_in, 3), np.float32, tensor=obs.uvw)
        d_wts = accel.DeviceArray(ctx, (1, n_in, 1), np.float32, tensor=obs.weights[None].contiguous())
        d_vis = accel.DeviceArray(ctx, (1, n_in, 1), np.complex64, tensor=obs.raw_vis[None].contiguous())
        torch.cuda.synchronize()
        coll = preprocess.VisibilityCollectorDevice(q, [ipd], [gpd], 1 << 20)
        coll.add(d_uvw, d_wts, d_vis, None, None, np.identity(1, np.complex64), None)
        coll.close()
        q.finish()
        torch.cuda.synchronize()
        stores.append(coll.reader())
        params.append((ipd, gpd, apd))
        del obs, d_uvw, d_wts, d_vis, coll
    torch.cuda.synchronize()
    block = max(r.len(0, 0) for r in stores)
    imagers = {}

    def make_job(channel, worker=0, own_queue=True):
        # (an imager per channel: the channels' parameters differ -- cell size and wavelength follow
        # the frequency -- and frontend.process_channel refuses an imager made for others)
        ipd, gpd, apd = params[channel]
        key = (channel, own_queue)
        if key not in imagers:
            queue = ctx.create_command_queue() if own_queue else q
            template = imaging.ImagingTemplate(ctx, apd, ipd.fixed, wparm, gpd.fixed, cp)
            im = template.instantiate(queue, ipd, gpd, block, 0, 2)
            im.ensure_all_bound()
            imagers[key] = im
        return dict(reader=stores[channel], rel_channel=0, imager=imagers[key], image_p=ipd, grid_p=gpd,
                    clean_p=cp, weight_type=wparm.weight_type, vis_block=block, major=2, degrid=True)

    def summary(job, stats):
        im = job['imager']
        comps = {k: float(v[0]) for k, v in im._model_components.items()}
        return stats, comps, im.get_buffer('dirty')

    # the sharded driver: this rank's share of eight channels (all of them), four in flight
    got = {}

    # (what a channel's imager holds is read when its worker is done with the channel)
    orig = frontend.process_channel

    def process_and_read(**kwargs):
        stats = orig(**kwargs)
        channel = stores.index(kwargs['reader'])
        got[channel] = summary(kwargs, stats)
        return stats
    frontend.process_channel = process_and_read
    try:
        out = parallel.image_assigned_channels(lambda ch, worker: make_job(ch, worker), channels, workers=4)
    finally:
        frontend.process_channel = orig
    assert sorted(out) == list(range(channels)) and sorted(got) == list(range(channels))
    # every channel on its own, one after the other
    for c in range(channels):
        job = make_job(c, 0, own_queue=False)
        stats = frontend.process_channel(**job)
        q.finish()
        want_stats, want_comps, want_dirty = summary(job, stats)
        got_stats, got_comps, got_dirty = got[c]
        assert got_stats['psf_patch'] == want_stats['psf_patch'] and got_stats['major'] == want_stats['major'] == 2
        assert got_stats['minor'] == want_stats['minor']
        # (1.5 M float32 terms summed in two different orders: 1e-4; the second major cycle starts from
        # what 200 components of the first left, where the faint ones may differ)
        np.testing.assert_allclose(got_stats['peaks'][0], want_stats['peaks'][0], rtol=1e-4)
        np.testing.assert_allclose(got_stats['peaks'], want_stats['peaks'], rtol=1e-2)
        common = set(got_comps) & set(want_comps)
        assert len(common) >= 0.9 * max(len(got_comps), len(want_comps)), (c, len(common), len(want_comps))
        # (the bright ones may not move)
        brightest = sorted(want_comps, key=lambda k: -abs(want_comps[k]))[:20]
        assert all(k in got_comps for k in brightest)
        # (two runs whose peaks differ in the last bits may give the last cycles before the limit to
        # different sources: a component's flux may differ by a cycle's worth or two -- the loop gain
        # times the level the loop has come down to -- not by more)
        level = min(abs(p) for p in want_stats['peaks'])
        for k in common:
            assert abs(got_comps[k] - want_comps[k]) <= 0.25 * level + 3e-4 * abs(want_comps[k]), (c, k)
        total_got, total_want = sum(got_comps.values()), sum(want_comps.values())
        assert abs(total_got - total_want) <= 0.01 * abs(total_want)
        # (the residual is what is left after the components: a few faint ones that resolve
        # differently between two runs of float atomics are a larger share of it than of the image)
        assert relerr(got_dirty, want_dirty) < 2e-3
    # an imager made for another channel's parameters is refused
    wrong = make_job(1, 0, own_queue=False)
    wrong['imager'] = imagers[(0, False)]
    with pytest.raises(ValueError):
        frontend.process_channel(**wrong)
    # the channels are not each other's: a different one's components do not fit
    assert len(set(got[0][1]) & set(got[1][1])) < 0.2 * len(got[0][1])
