"""GPU parity tests: the HIP path (through the C ABI and the operator classes)
against the oracle and the committed golden vectors.  Run with ``-m gpu``.

Tolerances (BASELINE.json north_star): bit-exact for index / peak selection;
norm-wise max|a-b|/max|b| <= 1e-5 for float32 grids and images; the
reference's own 5e-4-class tolerance for DFT prediction (test_predict.py:92).
"""
import numpy as np
import pytest

import golden_inputs as gi
from helpers import context_queue, kernel_taper, make_params, relerr, tapered_relerr
from oracle import kimg_oracle as orc

pytestmark = pytest.mark.gpu

GRID_TOL = 1e-5


ARITHS = ['fp32', 'split_fp16']         # KIMG_ARITH_*: both forms of the window kernels


def _gridder(c, variant, max_vis=1280):
    """`variant` = 'generic' | 'mfma' | 'auto', optionally followed by ':' and the arithmetic."""
    from katsdpimager_amd import grid
    ctx, q = context_queue()
    ip, gp, ap = make_params(c)
    variant, _, arith = variant.partition(':')
    template = grid.GridderTemplate(ctx, ip.fixed, gp.fixed,
                                    {'variant': variant, 'arith': arith or 'fp32'})
    fn = template.instantiate(q, ap, ip, gp, max_vis)
    fn.ensure_all_bound()
    return fn, q


def _run_gridder(fn, q, t):
    n = len(t['uv'])
    fn.buffer('grid').zero(q)
    wg = np.zeros(fn.buffer('weights_grid').shape, np.float32)
    gi.middle(wg, t['weights_grid'].shape)[:] = t['weights_grid']
    fn.buffer('weights_grid').set(q, wg)
    fn.num_vis = n
    fn.buffer('uv').set_region(q, np.concatenate((t['uv'], t['sub_uv']), axis=1), np.s_[:n], np.s_[:])
    fn.buffer('w_plane').set_region(q, t['w_plane'], np.s_[:n], np.s_[:])
    fn.buffer('vis').set_region(q, t['vis'], np.s_[:n], np.s_[:])
    fn()
    return fn.buffer('grid').get(q)


@pytest.mark.parametrize('variant', ['generic', 'mfma', 'mfma:split_fp16'])
@pytest.mark.parametrize('name', ['p4_f32', 'p1_k8', 'p2_k60'])
def test_gridder_vs_golden(golden, name, variant):
    """G2: reference GridderHost output on the test_grid.py track recipe."""
    c = gi.GRID_CONFIGS[name]
    t = gi.grid_track(c)
    fn, q = _gridder(c, variant, max_vis=2048)
    actual = _run_gridder(fn, q, t)
    expected = gi.middle(golden('g2_grid_' + name)['grid'], actual.shape)
    assert relerr(actual, expected) < GRID_TOL


@pytest.mark.gpu
@pytest.mark.parametrize('cus', [1, 7, 64, 192])
def test_window_kernels_on_part_of_the_device(cus):
    """kimg_set_window_cus: gridder and degridder on a part of the CUs (what a process with several
    channels in flight sets, so that the other channels' CLEAN launches find room) give the results
    of the whole device; the setting is process-wide and restored here."""
    from katsdpimager_amd import grid
    from katsdpimager_amd._lib import lib
    c = gi.make_config(512, 0.0001, 0.01, 2, 28, 16, grid_cover=300, n_vis=60000)
    t = gi.grid_track(c)
    fn, q = _gridder(c, 'mfma', max_vis=65536)
    whole = _run_gridder(fn, q, t)
    assert lib().kimg_get_window_cus() == 256
    try:
        assert lib().kimg_set_window_cus(cus) == 0 and lib().kimg_get_window_cus() == cus
        part = _run_gridder(fn, q, t)
        assert relerr(part, whole) < 2e-6
        ip, gp, ap = make_params(c)
        d = grid.DegridderTemplate(fn.template.context, ip.fixed, gp.fixed,
                                   {'variant': 'mfma'}).instantiate(q, ap, ip, gp, 65536)
        d.bind(grid=fn.buffer('grid'), uv=fn.buffer('uv'), w_plane=fn.buffer('w_plane'))
        d.ensure_all_bound()
        d.num_vis = len(t['uv'])
        res = {}
        for setting in (cus, 0):
            assert lib().kimg_set_window_cus(setting) == 0
            d.buffer('weights').set_region(q, np.ones(t['vis'].shape, np.float32),
                                           np.s_[:len(t['uv'])], np.s_[:])
            d.buffer('vis').set_region(q, t['vis'], np.s_[:len(t['uv'])], np.s_[:])
            d()
            res[setting] = d.buffer('vis').get(q)[:len(t['uv'])]
        assert lib().kimg_get_window_cus() == 256          # 0 = all
        np.testing.assert_allclose(res[cus], res[0], rtol=1e-5, atol=1e-5 * np.abs(res[0]).max())
    finally:
        lib().kimg_set_window_cus(0)


@pytest.mark.gpu
def test_window_cus_travel_with_the_call():
    """`Gridder.window_cus` / KIMG_WINDOW_CUS: the share of the CUs is an argument of every
    kimg_grid / kimg_degrid call, not process state -- two imagers of one process with different
    settings, gridding concurrently from two host threads on two streams, each get the results of
    the whole device, the process-wide default stays what it was, and a value outside 1 .. 256 is
    refused."""
    import threading
    from katsdpimager_amd._lib import lib
    c = gi.make_config(512, 0.0001, 0.01, 2, 28, 16, grid_cover=300, n_vis=60000)
    t = gi.grid_track(c)
    fn, q = _gridder(c, 'mfma', max_vis=65536)
    whole = _run_gridder(fn, q, t)
    ops = []
    ip, gp, ap = make_params(c)
    for cus in (5, 192):
        qq = fn.template.context.create_command_queue()        # a stream of its own
        op = fn.template.instantiate(qq, ap, ip, gp, 65536)
        op.ensure_all_bound()
        op.window_cus = cus
        ops.append((op, qq))
    out, errors = {}, []

    def work(k):
        try:
            import torch
            with torch.cuda.device(0):
                for _ in range(6):
                    out[k] = _run_gridder(ops[k][0], ops[k][1], t)
        except Exception as exc:        # noqa: B902
            errors.append(exc)
    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert relerr(out[0], whole) < 2e-6 and relerr(out[1], whole) < 2e-6
    assert lib().kimg_get_window_cus() == 256
    ops[0][0].window_cus = 257
    with pytest.raises(Exception):
        _run_gridder(ops[0][0], ops[0][1], t)


@pytest.mark.parametrize('variant', ['generic', 'mfma', 'mfma:split_fp16'])
@pytest.mark.parametrize('P', [1, 2, 3, 4])
def test_gridder_bruteforce(variant, P):
    """test_grid.py:91-112 (do_grid) at the reference test's own size (256^2, K=28, 32 planes,
    1000 vis): float64 brute force as the expectation, norm-wise 1e-5."""
    c = gi.make_config(256, 0.0001, 0.01, P, 28, 32, grid_cover=180, n_vis=1000)
    t = gi.grid_track(c)
    fn, q = _gridder(c, variant)
    actual = _run_gridder(fn, q, t)
    kernel = fn.convolve_kernel.data
    G = actual.shape[-1]
    expected = np.zeros(actual.shape, np.complex128)
    uv_bias = (28 - 1) // 2 - G // 2
    for i in range(c['n_vis']):
        k = np.conj(np.outer(kernel[t['w_plane'][i], t['sub_uv'][i, 1], :],
                             kernel[t['w_plane'][i], t['sub_uv'][i, 0], :]).astype(np.complex128))
        u = t['uv'][i, 0] - uv_bias
        v = t['uv'][i, 1] - uv_bias
        wu = t['uv'][i, 0] + t['weights_grid'].shape[2] // 2
        wv = t['uv'][i, 1] + t['weights_grid'].shape[1] // 2
        for j in range(P):
            expected[j, v:v + 28, u:u + 28] += (t['vis'][i, j].astype(np.complex128)
                                                * t['weights_grid'][j, wv, wu] * k)
    assert relerr(actual, expected) < GRID_TOL
    # and against the float32 oracle
    G_or = np.zeros(actual.shape, np.complex64)
    wg = np.zeros(actual.shape, np.float32)
    gi.middle(wg, t['weights_grid'].shape)[:] = t['weights_grid']
    orc.grid(kernel, G_or, wg, t['uv'], t['sub_uv'], t['w_plane'], t['vis'])
    assert relerr(actual, G_or) < GRID_TOL


@pytest.mark.parametrize('variant', ['generic', 'mfma', 'mfma:split_fp16'])
@pytest.mark.parametrize('K,P', [(33, 1), (45, 2), (60, 1), (60, 4), (64, 3)])
def test_gridder_wide_kernels(variant, K, P):
    """Kernel widths 33..64 (60 is the reference's CLI default, frontend.py:325): the MFMA
    gridder runs them as 2 x 2 tap blocks; odd widths split unevenly.  Against the oracle."""
    c = gi.make_config(512, 0.0001, 0.01, P, K, 16, grid_cover=300, n_vis=1500)
    t = gi.grid_track(c)
    fn, q = _gridder(c, variant, max_vis=2048)
    actual = _run_gridder(fn, q, t)
    expected = np.zeros(actual.shape, np.complex64)
    wg = np.zeros(actual.shape, np.float32)
    gi.middle(wg, t['weights_grid'].shape)[:] = t['weights_grid']
    orc.grid(fn.convolve_kernel.data, expected, wg, t['uv'], t['sub_uv'], t['w_plane'], t['vis'])
    assert relerr(actual, expected) < GRID_TOL
    # no locality at all
    rs = np.random.RandomState(K)
    n = 700
    half = (t['weights_grid'].shape[-1] // 2) - 1
    adv = dict(uv=rs.randint(-half, half, (n, 2)).astype(np.int16),
               sub_uv=rs.randint(0, 8, (n, 2)).astype(np.int16),
               w_plane=rs.randint(0, 16, n).astype(np.int16), weights_grid=t['weights_grid'],
               vis=(rs.standard_normal((n, P)) + 1j * rs.standard_normal((n, P))).astype(np.complex64))
    actual = _run_gridder(fn, q, adv)
    expected[:] = 0
    orc.grid(fn.convolve_kernel.data, expected, wg, adv['uv'], adv['sub_uv'], adv['w_plane'], adv['vis'])
    assert relerr(actual, expected) < GRID_TOL


@pytest.mark.parametrize('arith', ARITHS)
@pytest.mark.parametrize('K,OV,W,P', [(1, 8, 3, 1), (2, 4, 8, 2), (7, 16, 4, 1), (15, 2, 8, 3),
                                      (31, 8, 5, 1), (32, 8, 8, 2), (32, 4, 1, 4), (27, 2, 7, 1)])
def test_grid_degrid_odd_shapes(K, OV, W, P, arith):
    """Unusual kernel widths (1, odd, exactly the window width: no slack), oversampling factors and
    plane counts through the MFMA gridder and degridder, smooth and scattered positions."""
    from katsdpimager_amd import grid
    ctx, q = context_queue()
    c = gi.make_config(256, 0.0001, 0.01, P, K, W, oversample=OV, grid_cover=180, n_vis=1200)
    t = gi.grid_track(c)
    rs = gi.RandomState(K * 100 + OV)
    n2 = 800
    scattered = dict(
        uv=rs.randint(-80, 80, (n2, 2)).astype(np.int16),
        sub_uv=rs.randint(0, OV, (n2, 2)).astype(np.int16),
        w_plane=rs.randint(0, W, n2).astype(np.int16), weights_grid=t['weights_grid'],
        vis=rs.complex_uniform(-1, 1, size=(n2, P)).astype(np.complex64))
    fn, q = _gridder(c, 'mfma:' + arith, max_vis=2048)
    kernel = fn.convolve_kernel.data
    assert kernel.shape == (W, OV, K)
    ip, gp, ap = make_params(c)
    dg = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed, {'arith': arith}).instantiate(q, ap, ip, gp, 2048)
    dg.ensure_all_bound()
    G = dg.buffer('grid').shape[-1]
    gdata = rs.complex_uniform(-1, 1, size=(P, G, G)).astype(np.complex64)
    dg.buffer('grid').set(q, gdata)
    for data in (t, scattered):
        actual = _run_gridder(fn, q, data)
        expected = np.zeros(actual.shape, np.complex64)
        wg = np.zeros(actual.shape, np.float32)
        gi.middle(wg, data['weights_grid'].shape)[:] = data['weights_grid']
        orc.grid(kernel, expected, wg, data['uv'], data['sub_uv'], data['w_plane'], data['vis'])
        assert relerr(actual, expected) < GRID_TOL
        n = len(data['uv'])
        vis = rs.complex_uniform(-1, 1, size=(n, P)).astype(np.complex64)
        w = rs.uniform(0.5, 1.5, size=(n, P)).astype(np.float32)
        dg.num_vis = n
        dg.buffer('uv').set_region(q, np.concatenate((data['uv'], data['sub_uv']), axis=1),
                                   np.s_[:n], np.s_[:])
        dg.buffer('w_plane').set_region(q, data['w_plane'], np.s_[:n], np.s_[:])
        dg.buffer('vis').set_region(q, vis, np.s_[:n], np.s_[:])
        dg.buffer('weights').set_region(q, w, np.s_[:n], np.s_[:])
        dg()
        want = vis.copy()
        orc.degrid(kernel, gdata, data['uv'], data['sub_uv'], data['w_plane'], w, want)
        got = dg.buffer('vis').get(q)[:n]
        assert np.abs(got - want).max() <= 1e-5 * max(np.abs(want).max(), 1.0)


@pytest.mark.parametrize('arith', ARITHS)
@pytest.mark.parametrize('K,W,P', [(28, 128, 1), (28, 300, 2), (16, 96, 4), (60, 48, 1), (45, 64, 3)])
def test_gridder_many_w_planes(K, W, P, arith):
    """More W planes than an LDS-resident table allows (the reference's default w-step gives
    hundreds per slice): the MFMA gridder reads a padded copy of the table from HBM instead.
    Smooth track and scattered positions over all planes, against the oracle; the result must
    also agree with the per-tap kernel."""
    c = gi.make_config(512, 0.0001, 0.01, P, K, W, grid_cover=300, n_vis=1500)
    t = gi.grid_track(c)
    fn, q = _gridder(c, 'mfma:' + arith, max_vis=2048)
    assert fn._workspace_bytes > 0
    kernel = fn.convolve_kernel.data
    rs = gi.RandomState(K + W)
    n2 = 900
    half = t['weights_grid'].shape[-1] // 2 - 1
    scattered = dict(
        uv=rs.randint(-half, half, (n2, 2)).astype(np.int16),
        sub_uv=rs.randint(0, 8, (n2, 2)).astype(np.int16),
        w_plane=rs.randint(0, W, n2).astype(np.int16), weights_grid=t['weights_grid'],
        vis=rs.complex_uniform(-1, 1, size=(n2, P)).astype(np.complex64))
    fg, _ = _gridder(c, 'generic', max_vis=2048)
    for data in (t, scattered):
        actual = _run_gridder(fn, q, data)
        expected = np.zeros(actual.shape, np.complex64)
        wg = np.zeros(actual.shape, np.float32)
        gi.middle(wg, data['weights_grid'].shape)[:] = data['weights_grid']
        orc.grid(kernel, expected, wg, data['uv'], data['sub_uv'], data['w_plane'], data['vis'])
        assert relerr(actual, expected) < GRID_TOL
        assert relerr(_run_gridder(fg, q, data), actual) < GRID_TOL


@pytest.mark.parametrize('arith', ARITHS)
@pytest.mark.parametrize('K,W,P', [(28, 128, 1), (28, 300, 2), (16, 96, 4), (60, 48, 1), (45, 64, 3)])
def test_degridder_many_w_planes(K, W, P, arith):
    """The degridder with its table in HBM (more W planes than LDS holds), against the oracle."""
    from katsdpimager_amd import grid
    ctx, q = context_queue()
    c = gi.make_config(512, 0.0001, 0.01, P, K, W, grid_cover=300, n_vis=1500)
    t = gi.grid_track(c)
    ip, gp, ap = make_params(c)
    fn = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed, {'arith': arith}).instantiate(q, ap, ip, gp, 2048)
    fn.ensure_all_bound()
    assert fn._workspace_bytes > 0
    G = fn.buffer('grid').shape[-1]
    rs = gi.RandomState(K + W + 1)
    gdata = rs.complex_uniform(-1, 1, size=(P, G, G)).astype(np.complex64)
    fn.buffer('grid').set(q, gdata)
    kernel = fn.convolve_kernel.data
    half = t['weights_grid'].shape[-1] // 2 - 1
    n2 = 900
    cases = [(np.concatenate((t['uv'], t['sub_uv']), axis=1), t['w_plane']),
             (np.concatenate([rs.randint(-half, half, (n2, 2)), rs.randint(0, 8, (n2, 2))],
                             axis=1).astype(np.int16), rs.randint(0, W, n2).astype(np.int16))]
    for uv, wp in cases:
        n = len(uv)
        vis = rs.complex_uniform(-1, 1, size=(n, P)).astype(np.complex64)
        w = rs.uniform(0.5, 1.5, size=(n, P)).astype(np.float32)
        fn.num_vis = n
        fn.buffer('uv').set_region(q, uv, np.s_[:n], np.s_[:])
        fn.buffer('w_plane').set_region(q, wp, np.s_[:n], np.s_[:])
        fn.buffer('vis').set_region(q, vis, np.s_[:n], np.s_[:])
        fn.buffer('weights').set_region(q, w, np.s_[:n], np.s_[:])
        fn()
        expected = vis.copy()
        orc.degrid(kernel, gdata, np.ascontiguousarray(uv[:, :2]), np.ascontiguousarray(uv[:, 2:]),
                   wp, w, expected)
        actual = fn.buffer('vis').get(q)[:n]
        assert np.abs(actual - expected).max() <= 1e-5 * np.abs(expected).max()


def test_kernel_wider_than_mfma_window():
    """K = 70 exceeds the 2 x 2 tap-block range: the automatic variant falls back to the
    per-tap kernels, the explicit MFMA variant is refused (KIMG_EUNSUPPORTED)."""
    from katsdpimager_amd import grid, _lib
    ctx, q = context_queue()
    c = gi.make_config(512, 0.0001, 0.01, 1, 70, 4, grid_cover=300, n_vis=300)
    t = gi.grid_track(c)
    fn, q = _gridder(c, 'auto', max_vis=512)
    actual = _run_gridder(fn, q, t)
    expected = np.zeros(actual.shape, np.complex64)
    wg = np.zeros(actual.shape, np.float32)
    gi.middle(wg, t['weights_grid'].shape)[:] = t['weights_grid']
    orc.grid(fn.convolve_kernel.data, expected, wg, t['uv'], t['sub_uv'], t['w_plane'], t['vis'])
    assert relerr(actual, expected) < GRID_TOL
    fn2, q = _gridder(c, 'mfma', max_vis=512)
    with pytest.raises(_lib.KimgError):
        _run_gridder(fn2, q, t)
    ip, gp, ap = make_params(c)
    dg = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed).instantiate(q, ap, ip, gp, 512)
    dg.ensure_all_bound()
    G = dg.buffer('grid').shape[-1]
    rs = gi.RandomState(70)
    gdata = rs.complex_uniform(-1, 1, size=(1, G, G)).astype(np.complex64)
    dg.buffer('grid').set(q, gdata)
    n = len(t['uv'])
    vis = rs.complex_uniform(-1, 1, size=(n, 1)).astype(np.complex64)
    w = rs.uniform(0.5, 1.5, size=(n, 1)).astype(np.float32)
    dg.num_vis = n
    dg.buffer('uv').set_region(q, np.concatenate((t['uv'], t['sub_uv']), axis=1), np.s_[:n], np.s_[:])
    dg.buffer('w_plane').set_region(q, t['w_plane'], np.s_[:n], np.s_[:])
    dg.buffer('vis').set_region(q, vis, np.s_[:n], np.s_[:])
    dg.buffer('weights').set_region(q, w, np.s_[:n], np.s_[:])
    dg()
    want = vis.copy()
    orc.degrid(fn.convolve_kernel.data, gdata, t['uv'], t['sub_uv'], t['w_plane'], w, want)
    assert np.abs(dg.buffer('vis').get(q)[:n] - want).max() <= 1e-5 * np.abs(want).max()


@pytest.mark.parametrize('arith', ARITHS)
def test_gridder_64_planes(arith):
    """64 W-planes: the doubled LDS table does not fit, single-row variant (config 4)."""
    c = gi.make_config(256, 0.0001, 0.01, 2, 28, 64, grid_cover=180, n_vis=1000)
    t = gi.grid_track(c)
    fn, q = _gridder(c, 'mfma:' + arith)
    actual = _run_gridder(fn, q, t)
    expected = np.zeros(actual.shape, np.complex64)
    wg = np.zeros(actual.shape, np.float32)
    gi.middle(wg, t['weights_grid'].shape)[:] = t['weights_grid']
    orc.grid(fn.convolve_kernel.data, expected, wg, t['uv'], t['sub_uv'], t['w_plane'], t['vis'])
    assert relerr(actual, expected) < GRID_TOL


@pytest.mark.parametrize('variant', ['generic', 'mfma', 'mfma:split_fp16'])
def test_gridder_edge_cases(variant):
    """Empty input (grid.py:810-811), a single visibility, odd counts, batches that are not a
    multiple of 64, large jumps between consecutive visibilities, repeated positions."""
    c = gi.make_config(256, 0.0001, 0.01, 1, 28, 32, grid_cover=180, n_vis=1000)
    t = gi.grid_track(c)
    fn, q = _gridder(c, variant, max_vis=4096)
    kernel = fn.convolve_kernel.data
    wg = np.zeros(fn.buffer('grid').shape, np.float32)
    gi.middle(wg, t['weights_grid'].shape)[:] = t['weights_grid']
    rs = np.random.RandomState(5)
    for n in (0, 1, 2, 63, 65, 129, 777):
        sel = np.sort(rs.choice(1000, n, replace=False)) if n else np.zeros(0, int)
        sub = {k: (t[k][sel] if k != 'weights_grid' else t[k]) for k in t}
        actual = _run_gridder(fn, q, sub) if n else None
        if n == 0:
            fn.buffer('grid').zero(q)
            fn.num_vis = 0
            fn()
            assert not np.any(fn.buffer('grid').get(q))
            continue
        expected = np.zeros(actual.shape, np.complex64)
        orc.grid(kernel, expected, wg, sub['uv'], sub['sub_uv'], sub['w_plane'], sub['vis'])
        assert relerr(actual, expected) < GRID_TOL, n
    # adversarial: uniformly random positions (no locality) + all at one position
    n = 3000
    adv = dict(uv=rs.randint(-90, 90, (n, 2)).astype(np.int16),
               sub_uv=rs.randint(0, 8, (n, 2)).astype(np.int16),
               w_plane=rs.randint(0, 32, n).astype(np.int16), weights_grid=t['weights_grid'],
               vis=(rs.standard_normal((n, 1)) + 1j * rs.standard_normal((n, 1))).astype(np.complex64))
    adv['uv'][2000:] = adv['uv'][2000]
    actual = _run_gridder(fn, q, adv)
    expected = np.zeros(actual.shape, np.complex64)
    orc.grid(kernel, expected, wg, adv['uv'], adv['sub_uv'], adv['w_plane'], adv['vis'])
    assert relerr(actual, expected) < GRID_TOL
    with pytest.raises(ValueError):
        fn.num_vis = 5000          # grid.py:700-702


def test_gridder_too_small_image():
    """grid.py:759-761."""
    from katsdpimager_amd import grid
    ctx, q = context_queue()
    c = gi.make_config(256, 0.0001, 0.01, 1, 28, 32)
    ip, gp, ap = make_params(c, longest_baseline=0.390625 * 120)
    template = grid.GridderTemplate(ctx, ip.fixed, gp.fixed)
    with pytest.raises(ValueError):
        template.instantiate(q, ap, ip, gp, 100)


@pytest.mark.parametrize('arith', ARITHS)
@pytest.mark.parametrize('name', ['p4_f32', 'p1_k8', 'p2_k60'])
def test_degridder_vs_golden(golden, name, arith):
    """G3: reference DegridderHost residuals (rtol 1e-5 as test_grid.py:135, plus an absolute
    floor for values that cancel)."""
    from katsdpimager_amd import grid
    ctx, q = context_queue()
    c = gi.GRID_CONFIGS[name]
    t = gi.grid_track(c)
    dg = gi.degrid_inputs(c)
    ip, gp, ap = make_params(c)
    fn = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed, {'arith': arith}).instantiate(q, ap, ip, gp, 2048)
    fn.ensure_all_bound()
    n = c['n_vis']
    fn.buffer('grid').set(q, gi.middle(dg['grid'], fn.buffer('grid').shape))
    fn.num_vis = n
    fn.buffer('uv').set_region(q, np.concatenate((t['uv'], t['sub_uv']), axis=1), np.s_[:n], np.s_[:])
    fn.buffer('w_plane').set_region(q, t['w_plane'], np.s_[:n], np.s_[:])
    fn.buffer('vis').set_region(q, dg['vis'], np.s_[:n], np.s_[:])
    fn.buffer('weights').set_region(q, dg['weights'], np.s_[:n], np.s_[:])
    fn()
    actual = fn.buffer('vis').get(q)[:n]
    expected = golden('g3_degrid_' + name)['residual']
    np.testing.assert_allclose(actual, expected, rtol=1e-5, atol=1e-5)
    fn.num_vis = 0
    fn()                                      # grid.py:989-990


@pytest.mark.parametrize('arith', ARITHS)
@pytest.mark.parametrize('P,W', [(1, 32), (2, 32), (3, 32), (4, 32), (1, 64), (4, 64)])
def test_degridder_adversarial(P, W, arith):
    """Uniformly random positions (every 16-group needs several window passes), repeated
    positions, ragged counts and zero visibilities, against the oracle.  W = 64 planes does not
    fit the doubled LDS table and takes the single-row (wrapping) variant."""
    from katsdpimager_amd import grid
    ctx, q = context_queue()
    c = gi.make_config(256, 0.0001, 0.01, P, 28, W, grid_cover=180, n_vis=1000)
    ip, gp, ap = make_params(c)
    fn = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed, {'arith': arith}).instantiate(q, ap, ip, gp, 4096)
    fn.ensure_all_bound()
    G = fn.buffer('grid').shape[-1]
    rs = gi.RandomState(7)
    gdata = rs.complex_uniform(-1, 1, size=(P, G, G)).astype(np.complex64)
    fn.buffer('grid').set(q, gdata)
    kernel = fn.convolve_kernel.data
    for n in (0, 1, 15, 17, 64, 1000, 3001):
        uv = np.concatenate([rs.randint(-90, 90, (n, 2)), rs.randint(0, 8, (n, 2))],
                            axis=1).astype(np.int16)
        if n > 600:
            uv[500:600] = uv[500]
        wp = rs.randint(0, W, n).astype(np.int16)
        vis = rs.complex_uniform(-1, 1, size=(n, P)).astype(np.complex64)
        w = rs.uniform(0.5, 1.5, size=(n, P)).astype(np.float32)
        fn.num_vis = n
        if n:
            fn.buffer('uv').set_region(q, uv, np.s_[:n], np.s_[:])
            fn.buffer('w_plane').set_region(q, wp, np.s_[:n], np.s_[:])
            fn.buffer('vis').set_region(q, vis, np.s_[:n], np.s_[:])
            fn.buffer('weights').set_region(q, w, np.s_[:n], np.s_[:])
        fn()
        if n == 0:
            continue
        expected = vis.copy()
        orc.degrid(kernel, gdata, np.ascontiguousarray(uv[:, :2]), np.ascontiguousarray(uv[:, 2:]),
                   wp, w, expected)
        actual = fn.buffer('vis').get(q)[:n]
        np.testing.assert_allclose(actual, expected, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize('arith', ARITHS)
@pytest.mark.parametrize('K,P', [(33, 1), (45, 2), (60, 1), (60, 4), (64, 3)])
def test_degridder_wide_kernels(K, P, arith):
    """Kernel widths 33..64 through the MFMA degridder (2 x 2 tap blocks, each subtracting its
    partial sum) against the oracle: a smooth track and positions without any locality."""
    from katsdpimager_amd import grid
    ctx, q = context_queue()
    c = gi.make_config(512, 0.0001, 0.01, P, K, 16, grid_cover=300, n_vis=1500)
    t = gi.grid_track(c)
    ip, gp, ap = make_params(c)
    fn = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed, {'arith': arith}).instantiate(q, ap, ip, gp, 2048)
    fn.ensure_all_bound()
    G = fn.buffer('grid').shape[-1]
    rs = gi.RandomState(K)
    gdata = rs.complex_uniform(-1, 1, size=(P, G, G)).astype(np.complex64)
    fn.buffer('grid').set(q, gdata)
    kernel = fn.convolve_kernel.data
    half = t['weights_grid'].shape[-1] // 2 - 1
    n2 = 900
    cases = [(np.concatenate((t['uv'], t['sub_uv']), axis=1), t['w_plane']),
             (np.concatenate([rs.randint(-half, half, (n2, 2)), rs.randint(0, 8, (n2, 2))],
                             axis=1).astype(np.int16), rs.randint(0, 16, n2).astype(np.int16))]
    for uv, wp in cases:
        n = len(uv)
        vis = rs.complex_uniform(-1, 1, size=(n, P)).astype(np.complex64)
        w = rs.uniform(0.5, 1.5, size=(n, P)).astype(np.float32)
        fn.num_vis = n
        fn.buffer('uv').set_region(q, uv, np.s_[:n], np.s_[:])
        fn.buffer('w_plane').set_region(q, wp, np.s_[:n], np.s_[:])
        fn.buffer('vis').set_region(q, vis, np.s_[:n], np.s_[:])
        fn.buffer('weights').set_region(q, w, np.s_[:n], np.s_[:])
        fn()
        expected = vis.copy()
        orc.degrid(kernel, gdata, np.ascontiguousarray(uv[:, :2]), np.ascontiguousarray(uv[:, 2:]),
                   wp, w, expected)
        actual = fn.buffer('vis').get(q)[:n]
        scale = np.abs(expected).max()
        assert np.abs(actual - expected).max() <= 1e-5 * scale


def test_predict_vs_golden(golden):
    """G4: reference _predict_host (norm-wise: see tests/test_oracle_golden.py::test_g4)."""
    from katsdpimager_amd import predict
    ctx, q = context_queue()
    c = gi.PREDICT_CONFIG
    g = golden('g4_predict')
    pi = gi.predict_inputs(c)
    ip, gp, _ = make_params(c)
    n = c['n_vis']
    fn = predict.PredictTemplate(ctx, np.float32, c['P']).instantiate(q, ip, gp, n, 64)
    fn.ensure_all_bound()
    fn.num_vis = n
    fn.buffer('uv').set_region(q, np.concatenate((pi['uv'], pi['sub_uv']), axis=1), np.s_[:n], np.s_[:])
    fn.buffer('w_plane').set_region(q, pi['w_plane'], np.s_[:n], np.s_[:])
    fn.buffer('vis').set_region(q, pi['vis'], np.s_[:n], np.s_[:])
    fn.buffer('weights').set_region(q, pi['weights'], np.s_[:n], np.s_[:])
    fn.set_sky_image(pi['components'])
    assert fn.num_sources == len(pi['components'])
    fn.set_w(pi['w'])
    fn()
    actual = fn.buffer('vis').get(q)[:n]
    pred = pi['vis'] - actual
    pred_ref = pi['vis'] - g['residual']
    assert relerr(pred, pred_ref) < 2e-3
    uv_scale, w_scale, w_bias = g['uv_scale'], g['w_scale'], g['w_bias']
    u = (pi['uv'].astype(np.float64) * c['oversample'] + pi['sub_uv'] + 0.5) * uv_scale
    w = pi['w_plane'] * w_scale + w_bias + pi['w']
    lmn, flux = g['lmn'].astype(np.float64), g['flux'].astype(np.float64)
    phase = u[:, 0:1] * lmn[:, 0] + u[:, 1:2] * lmn[:, 1] + w[:, None] * lmn[:, 2]
    exact = (np.exp(-2j * np.pi * phase) @ flux) * pi['weights']
    assert relerr(pred, exact) < 2e-3
    with pytest.raises(ValueError):
        fn.set_sky_arrays(np.zeros((100, 3)), np.zeros((100, c['P'])))   # predict.py:337-338


def _run_predict(components, c, pi):
    from katsdpimager_amd import predict
    ctx, q = context_queue()
    ip, gp, _ = make_params(c)
    n = c['n_vis']
    fn = predict.PredictTemplate(ctx, np.float32, c['P']).instantiate(q, ip, gp, n, 64)
    fn.ensure_all_bound()
    fn.num_vis = n
    fn.buffer('uv').set_region(q, np.concatenate((pi['uv'], pi['sub_uv']), axis=1), np.s_[:n], np.s_[:])
    fn.buffer('w_plane').set_region(q, pi['w_plane'], np.s_[:n], np.s_[:])
    fn.buffer('vis').set_region(q, pi['vis'], np.s_[:n], np.s_[:])
    fn.buffer('weights').set_region(q, pi['weights'], np.s_[:n], np.s_[:])
    fn.set_sky_image(components)
    fn.set_w(pi['w'])
    fn()
    return fn.buffer('vis').get(q)[:n]


def _predict_exact(c, pi, g, golden_scales):
    u = (pi['uv'].astype(np.float64) * c['oversample'] + pi['sub_uv'] + 0.5) * golden_scales['uv_scale']
    w = pi['w_plane'] * golden_scales['w_scale'] + golden_scales['w_bias'] + pi['w']
    lmn, flux = g['lmn'].astype(np.float64), g['flux'].astype(np.float64)
    phase = u[:, 0:1] * lmn[:, 0] + u[:, 1:2] * lmn[:, 1] + w[:, None] * lmn[:, 2]
    return (np.exp(-2j * np.pi * phase) @ flux) * pi['weights']


def test_predict_mild_vs_golden_at_reference_tolerance(golden):
    """The reference's own predictor test (test_predict.py:54-92) compares device and host
    element-wise at rtol 5e-4 on sources within 2.5 arcmin of the phase centre; the same
    geometry here (golden_inputs.predict_mild_components), the same gate, against the
    reference's `_predict_host` (G4b)."""
    c = gi.PREDICT_CONFIG
    pi = gi.predict_inputs(c)
    g = golden('g4_predict_mild')
    actual = _run_predict(gi.predict_mild_components(), c, pi)
    pred, pred_ref = pi['vis'] - actual, pi['vis'] - g['residual']
    # The reference's gate is relative to each RESIDUAL (vis - prediction).  Where the two nearly
    # cancel no float32 evaluation can be held to 5e-4 of what is left (1 of these 903 residuals is
    # 7e-3 next to a prediction of 7: the two float32 paths differ there by 2e-5 = 3e-6 of the
    # prediction), so the relative gate gets an absolute floor of 4e-6 of the largest prediction:
    # the rounding of ONE float32 sum of that size.
    floor = 4e-6 * float(np.abs(pred_ref).max())
    np.testing.assert_allclose(actual, g['residual'], rtol=5e-4, atol=floor)
    assert np.count_nonzero(np.abs(actual - g['residual']) > 5e-4 * np.abs(g['residual'])) <= 2
    # ... and norm-wise on the predicted part, against the host and against float64
    exact = _predict_exact(c, pi, g, golden('g4_predict'))
    assert relerr(pred, pred_ref) < 5e-4
    assert relerr(pred, exact) < 5e-4


def test_predict_error_is_the_float32_phase_not_the_kernel(golden):
    """On G4's geometry (components out to the image corners: |phase| ~ 1e3 turns) both float32
    evaluations are ~1e-3 from the float64 truth.  The HIP kernel must be no farther from it than
    1.5 x the reference's host path -- i.e. the 2e-3 gate of test_predict_vs_golden is the float32
    phase, not slack in the kernel."""
    c = gi.PREDICT_CONFIG
    pi = gi.predict_inputs(c)
    g = golden('g4_predict')
    actual = _run_predict(pi['components'], c, pi)
    exact = _predict_exact(c, pi, g, g)
    err_hip = relerr(pi['vis'] - actual, exact)
    err_host = relerr(pi['vis'] - g['residual'], exact)
    print('predict vs float64: HIP %.3e, reference host %.3e' % (err_hip, err_host))
    assert err_hip <= 1.5 * err_host
    # element-wise too: the worst visibility of the HIP path is no worse than 1.5 x the host's worst
    d_hip = np.abs((pi['vis'] - actual) - exact).max()
    d_host = np.abs((pi['vis'] - g['residual']) - exact).max()
    assert d_hip <= 1.5 * d_host


@pytest.mark.parametrize('name', list(gi.IMAGE_CONFIGS))
def test_grid_image_vs_golden(golden, name):
    """G5: GridToImageHost / ImageToGridHost incl. accumulation, off-centre lm_bias, w != 0."""
    from katsdpimager_amd import image
    ctx, q = context_queue()
    c = gi.IMAGE_CONFIGS[name]
    g = golden('g5_image_' + name)
    ii = gi.image_inputs(c)
    shape = ii['image_shape']
    template = image.GridImageTemplate(ctx, np.float32)
    plan = template.make_fft_plan(shape[1:], shape[1:])
    g2i = template.instantiate_grid_to_image(q, shape, c['lm_scale'], c['lm_bias'], plan)
    i2g = template.instantiate_image_to_grid(q, shape, c['lm_scale'], c['lm_bias'], plan)
    g2i.ensure_all_bound()
    i2g.bind(layer=g2i.buffer('layer'), kernel1d=g2i.buffer('kernel1d'))
    i2g.ensure_all_bound()
    g2i.buffer('kernel1d').set(q, ii['kernel1d'])
    g2i.buffer('grid').set(q, ii['grid'])
    i2g.buffer('image').set(q, ii['model'])
    for wi, w in enumerate(c['ws']):
        g2i.buffer('image').zero(q)
        g2i.set_w(w)
        g2i()
        g2i()
        assert relerr(g2i.buffer('image').get(q), g['g2i_w%d' % wi]) < 1e-5
        i2g.set_w(w)
        i2g()
        assert relerr(i2g.buffer('grid').get(q), g['i2g_w%d' % wi]) < 1e-5
    with pytest.raises(ValueError):
        template.layer_to_image.instantiate(q, (1, 63, 63), 0.1, 0.0)     # image.py:127-128
    with pytest.raises(IndexError):
        g2i._layer_image.set_polarization(shape[0])                      # image.py:149-150


def test_grid_to_image_smaller_grid():
    """The grid may be smaller than the image (grid.py:753-767): zero padding by the
    quadrant copies (image.py:659-672) must equal the host result on the padded grid."""
    from katsdpimager_amd import image
    ctx, q = context_queue()
    rs = gi.RandomState(3)
    G, Gg, P = 96, 40, 2
    small = rs.complex_uniform(-1, 1, (P, Gg, Gg)).astype(np.complex64)
    k1d = rs.uniform(1.0, 2.0, G).astype(np.float32)
    lm_scale, lm_bias, w = 0.001, -0.5 * G * 0.001, 37.5
    template = image.GridImageTemplate(ctx, np.float32)
    plan = template.make_fft_plan((G, G))
    g2i = template.instantiate_grid_to_image(q, (P, Gg, Gg), lm_scale, lm_bias, plan)
    g2i.ensure_all_bound()
    g2i.buffer('kernel1d').set(q, k1d)
    g2i.buffer('grid').set(q, small)
    g2i.buffer('image').zero(q)
    g2i.set_w(w)
    g2i()
    full = np.zeros((P, G, G), np.complex64)
    gi.middle(full, small.shape)[:] = small
    expected = np.zeros((P, G, G), np.float32)
    orc.grid_to_image(full, expected, k1d, lm_scale, lm_bias, w)
    assert relerr(g2i.buffer('image').get(q), expected) < 1e-5
    # and back: image -> smaller grid = centre of the host's full grid
    i2g = template.instantiate_image_to_grid(q, (P, Gg, Gg), lm_scale, lm_bias, plan)
    i2g.bind(layer=g2i.buffer('layer'), kernel1d=g2i.buffer('kernel1d'))
    i2g.ensure_all_bound()
    model = rs.uniform(-1, 1, (P, G, G)).astype(np.float32)
    i2g.buffer('image').set(q, model)
    i2g.set_w(w)
    i2g()
    full_grid, _ = orc.image_to_grid(model, k1d, lm_scale, lm_bias, w)
    assert relerr(i2g.buffer('grid').get(q), gi.middle(full_grid, small.shape)) < 1e-5


#: tuning of image.GridImageTemplate for the three routes grid <-> image has at w = 0
_W0_ROUTES = {'own': {}, 'library': {'own_transform': False}, 'c2c': {'real_transform': False}}


@pytest.mark.gpu
@pytest.mark.parametrize('G,Gg,P', [(96, 40, 2), (64, 64, 1), (128, 126, 3), (4096, 2486, 1),
                                    (16, 2, 1), (16, 16, 2), (32, 6, 1), (512, 154, 2),
                                    (2048, 620, 1), (2048, 2048, 1), (8192, 2486, 1),
                                    (120, 36, 1), (70, 70, 2), (66, 20, 1), (1000, 300, 1),
                                    (4800, 1440, 1), (6720, 2016, 1), (5040, 5040, 1)])
def test_grid_to_image_real_transform_route(G, Gg, P):
    """w = 0: GridToImage takes the Hermitian part of the (zero-padded) grid through a
    complex-to-real transform of half the size -- with the library's own two-launch transforms
    where the layer size has no prime factor above 7 ('own': the sizes the reference picks,
    parameters.py:17-25), else on the FFT library's plan ('library').
    Both against the oracle (= the reference's host path on the padded grid) and against the
    complex-to-complex route of the same operator, with accumulation; grids as large as the image
    (the -G/2 row and column have no mirror) and smaller, even and odd log2 of the size."""
    from katsdpimager_amd import image
    from katsdpimager_amd._lib import lib
    ctx, q = context_queue()
    rs = gi.RandomState(G + Gg)
    small = rs.complex_uniform(-1, 1, (P, Gg, Gg)).astype(np.complex64)
    k1d = rs.uniform(1.0, 2.0, G).astype(np.float32)
    lm_scale = 0.3 / G
    lm_bias = -0.5 * G * lm_scale
    own = bool(lib().kimg_grid_image_real_supported(G, Gg))
    rest = G
    for prime in (2, 3, 5, 7):
        while rest % prime == 0:
            rest //= prime
    assert own == (rest == 1 and G >= 16)           # (66 = 2 * 3 * 11 stays with the FFT library)
    got = {}
    for route, tuning in _W0_ROUTES.items():
        template = image.GridImageTemplate(ctx, np.float32, tuning)
        g2i = template.instantiate_grid_to_image(q, (P, Gg, Gg), lm_scale, lm_bias,
                                                 template.make_fft_plan((G, G)))
        g2i.ensure_all_bound()
        g2i.buffer('kernel1d').set(q, k1d)
        g2i.buffer('grid').set(q, small)
        g2i.buffer('image').zero(q)
        g2i.set_w(0.0)
        g2i()
        g2i()                                   # accumulates
        got[route] = g2i.buffer('image').get(q)
        assert g2i._own_transform() == (own and route == 'own')
        assert (g2i._real_plan is not None) == (route == 'library' or (route == 'own' and not own))
        del g2i
    peak = np.abs(got['c2c']).max()
    for route in ('own', 'library'):
        assert np.abs(got[route] - got['c2c']).max() <= 2e-6 * peak, route
    if G <= 128:
        full = np.zeros((P, G, G), np.complex64)
        gi.middle(full, small.shape)[:] = small
        expected = np.zeros((P, G, G), np.float32)
        orc.grid_to_image(full, expected, k1d, lm_scale, lm_bias, 0.0)
        orc.grid_to_image(full, expected, k1d, lm_scale, lm_bias, 0.0)
        for route in ('own', 'library'):
            assert relerr(got[route], expected) < 1e-5, route
    with pytest.raises(ValueError):
        image.GridImageTemplate(ctx, np.float32, {'real': True})
    # and back (ImageToGrid at w = 0: real layer, real-to-complex transform, F(-k) = conj F(k))
    model = rs.uniform(-1, 1, (P, G, G)).astype(np.float32)
    back = {}
    for route, tuning in _W0_ROUTES.items():
        template = image.GridImageTemplate(ctx, np.float32, tuning)
        i2g = template.instantiate_image_to_grid(q, (P, Gg, Gg), lm_scale, lm_bias,
                                                 template.make_fft_plan((G, G)))
        i2g.ensure_all_bound()
        i2g.buffer('kernel1d').set(q, k1d)
        i2g.buffer('image').set(q, model)
        i2g.buffer('grid').zero(q)
        i2g.set_w(0.0)
        i2g()
        back[route] = i2g.buffer('grid').get(q)
        assert (i2g._real_plan is not None) == (route == 'library' or (route == 'own' and not own))
        del i2g
    peak = np.abs(back['c2c']).max()
    for route in ('own', 'library'):
        assert np.abs(back[route] - back['c2c']).max() <= 2e-6 * peak, route
    if G <= 128:
        full_grid, _ = orc.image_to_grid(model, k1d, lm_scale, lm_bias, 0.0)
        for route in ('own', 'library'):
            assert relerr(back[route], gi.middle(full_grid, small.shape)) < 1e-5, route


@pytest.mark.gpu
@pytest.mark.parametrize('G,Gg,P,w', [(64, 64, 1, 37.5), (128, 50, 2, -12.25), (16, 6, 1, 3.0),
                                      (512, 154, 1, 151.0), (2048, 620, 1, 40.0),
                                      (4096, 1244, 1, -75.5), (96, 40, 2, 21.0), (126, 126, 1, -9.5),
                                      (4800, 1440, 1, 33.0), (5040, 1512, 1, -18.0)])
def test_grid_image_own_transform_any_w(G, Gg, P, w):
    """w != 0 (a slice of the W stack away from the middle): the two-launch route on the library's
    own transforms (kimg_grid_to_image_w / kimg_image_to_grid_w) against the FFT library's
    complex-to-complex plan with the separate pad and correction kernels, and against the oracle."""
    from katsdpimager_amd import image
    ctx, q = context_queue()
    rs = gi.RandomState(G * 3 + Gg)
    small = rs.complex_uniform(-1, 1, (P, Gg, Gg)).astype(np.complex64)
    k1d = rs.uniform(1.0, 2.0, G).astype(np.float32)
    model = rs.uniform(-1, 1, (P, G, G)).astype(np.float32)
    lm_scale = 0.3 / G
    lm_bias = -0.5 * G * lm_scale
    got, back = {}, {}
    for route, tuning in (('own', {}), ('library', {'own_transform': False})):
        template = image.GridImageTemplate(ctx, np.float32, tuning)
        plan = template.make_fft_plan((G, G))
        g2i = template.instantiate_grid_to_image(q, (P, Gg, Gg), lm_scale, lm_bias, plan)
        g2i.ensure_all_bound()
        i2g = template.instantiate_image_to_grid(q, (P, Gg, Gg), lm_scale, lm_bias, plan)
        i2g.bind(layer=g2i.buffer('layer'), kernel1d=g2i.buffer('kernel1d'))
        i2g.ensure_all_bound()
        g2i.buffer('kernel1d').set(q, k1d)
        g2i.buffer('grid').set(q, small)
        g2i.buffer('image').set(q, np.full((P, G, G), 3.0, np.float32))
        g2i.set_w(w)
        assert g2i._own_transform() == (route == 'own')
        g2i.overwrite_next = g2i.can_overwrite()        # the library route accumulates onto ...
        if not g2i.can_overwrite():
            g2i.buffer('image').zero(q)                 # ... explicit zeros
        g2i()
        g2i()                                           # accumulates
        got[route] = g2i.buffer('image').get(q)
        i2g.buffer('image').set(q, model)
        i2g.buffer('grid').zero(q)
        i2g.set_w(w)
        i2g()
        back[route] = i2g.buffer('grid').get(q)
        del g2i, i2g
    assert np.abs(got['own'] - got['library']).max() <= 2e-6 * np.abs(got['library']).max()
    assert np.abs(back['own'] - back['library']).max() <= 2e-6 * np.abs(back['library']).max()
    if G <= 128:
        full = np.zeros((P, G, G), np.complex64)
        gi.middle(full, small.shape)[:] = small
        expected = np.zeros((P, G, G), np.float32)
        orc.grid_to_image(full, expected, k1d, lm_scale, lm_bias, w)
        orc.grid_to_image(full, expected, k1d, lm_scale, lm_bias, w)
        assert relerr(got['own'], expected) < 1e-5
        full_grid, _ = orc.image_to_grid(model, k1d, lm_scale, lm_bias, w)
        assert relerr(back['own'], gi.middle(full_grid, small.shape)) < 1e-5


def test_image_streams():
    """Scale / AddImage / ApplyPrimaryBeam known answers (test_image.py:91-162)."""
    from katsdpimager_amd import image
    ctx, q = context_queue()
    shape = (4, 123, 234)
    rs = np.random.RandomState(1)
    src = rs.uniform(size=shape).astype(np.float32)
    fn = image.ScaleTemplate(ctx, np.float32, 4).instantiate(q, shape)
    fn.ensure_all_bound()
    fn.buffer('data').set(q, src)
    sf = np.array([1.2, 2.3, 3.4, -4.5], np.float32)
    fn.set_scale_factor(sf)
    fn()
    np.testing.assert_array_equal(fn.buffer('data').get(q), src * sf[:, None, None])

    add = image.AddImageTemplate(ctx, np.float32, 4).instantiate(q, shape)
    add.ensure_all_bound()
    dest = rs.uniform(size=shape).astype(np.float32)
    add.buffer('src').set(q, src)
    add.buffer('dest').set(q, dest)
    add()
    np.testing.assert_array_equal(add.buffer('dest').get(q), src + dest)

    pb = image.ApplyPrimaryBeamTemplate(ctx, np.float32, 4).instantiate(q, shape, 0.2, 12345.0)
    pb.ensure_all_bound()
    beam = rs.uniform(size=shape[1:]).astype(np.float32)
    pb.buffer('data').set(q, src)
    pb.buffer('beam_power').set(q, beam)
    pb()
    np.testing.assert_array_equal(pb.buffer('data').get(q),
                                  np.where(beam < 0.2, np.float32(12345.0), src / beam))
    with pytest.raises(ValueError):
        image.ScaleTemplate(ctx, np.float32, 3).instantiate(q, shape)    # image.py:342-343


def test_grid_weights_known():
    """test_weight.py:10-57 (exact) incl. garbage beyond num_vis."""
    from katsdpimager_amd import weight
    ctx, q = context_queue()
    shape = (4, 100, 200)
    uv = np.array([[-10, 5, 0, 0], [23, 17, 0, 0], [-10, 5, 0, 0], [-10, 5, 0, 0],
                   [-10, 6, 0, 0], [-11, 5, 0, 0]], np.int16)
    w = np.array([[1.0, 10.0, 100.0, 1000.0], [2.0, 20.0, 200.0, 2000.0],
                  [4.0, 40.0, 400.0, 4000.0], [8.0, 80.0, 800.0, 8000.0],
                  [16.0, 160.0, 1600.0, 16000.0], [32.0, 320.0, 3200.0, 32000.0]], np.float32)
    fn = weight.GridWeightsTemplate(ctx, 4).instantiate(q, shape, 1000)
    fn.ensure_all_bound()
    rs = np.random.RandomState(1)
    uv_full = rs.randint(-40, 40, (1000, 4)).astype(np.int16)
    uv_full[:6] = uv
    w_full = rs.uniform(size=(1000, 4)).astype(np.float32)
    w_full[:6] = w
    fn.buffer('uv').set(q, uv_full)
    fn.buffer('weights').set(q, w_full)
    fn.buffer('grid').zero(q)
    fn.num_vis = 6
    fn()
    expected = np.zeros(shape, np.float32)
    for i in range(4):
        expected[i, 55, 90] = 13 * 10 ** i
        expected[i, 67, 123] = 2 * 10 ** i
        expected[i, 56, 90] = 16 * 10 ** i
        expected[i, 55, 89] = 32 * 10 ** i
    np.testing.assert_array_equal(fn.buffer('grid').get(q), expected)
    with pytest.raises(ValueError):
        weight.GridWeightsTemplate(ctx, 4).instantiate(q, (4, 101, 200), 10)   # weight.py:131


def test_density_and_mean_weight_known():
    """test_weight.py:60-118."""
    from katsdpimager_amd import weight
    ctx, q = context_queue()
    rs = np.random.RandomState(1)
    shape = (4, 50, 107)
    data = np.zeros(shape, np.float32)
    expected = np.zeros(shape, np.float32)
    sum_w = sum_dw = sum_d2w = 0.0
    for index in rs.choice(data.size, 100, replace=False):
        w = rs.uniform(low=0.1, high=2.0)
        d = 1.0 / (2.5 * w + 1.75)
        data.flat[index] = w
        expected.flat[index] = d
        if index < data[0].size:
            sum_w += w
            sum_dw += d * w
            sum_d2w += d ** 2 * w
    fn = weight.DensityWeightsTemplate(ctx, 4).instantiate(q, shape)
    fn.ensure_all_bound()
    fn.a, fn.b = 2.5, 1.75
    fn.buffer('grid').set(q, data)
    rms, nrms = fn()
    np.testing.assert_allclose(fn.buffer('grid').get(q), expected, 1e-5, 1e-5)
    np.testing.assert_allclose(rms, np.sqrt(sum_d2w) / sum_dw, 1e-6)
    np.testing.assert_allclose(nrms, np.sqrt(sum_d2w * sum_w) / sum_dw, 1e-6)
    data = rs.uniform(size=shape).astype(np.float32)
    mw = weight.MeanWeightTemplate(ctx).instantiate(q, shape)
    mw.ensure_all_bound()
    mw.buffer('grid').set(q, data)
    pol0 = data[0].astype(np.float64)
    np.testing.assert_allclose(mw(), np.sum(pol0 * pol0) / np.sum(pol0), rtol=1e-5)


def test_weights_vs_golden(golden):
    """G6: compound WeightsHost for natural / uniform / robust weighting."""
    from katsdpimager_amd import weight
    ctx, q = context_queue()
    g = golden('g6_weights')
    wi = gi.weights_inputs()
    n = len(wi['uv'])
    for name, wt in [('natural', weight.WeightType.NATURAL), ('uniform', weight.WeightType.UNIFORM),
                     ('robust', weight.WeightType.ROBUST)]:
        fn = weight.WeightsTemplate(ctx, wt, wi['shape'][0]).instantiate(q, wi['shape'], 2048)
        fn.ensure_all_bound()
        if wt == weight.WeightType.ROBUST:
            fn.robustness = wi['robustness']
        fn.clear()
        if wt != weight.WeightType.NATURAL:
            for start in range(0, n, 2048):
                stop = min(n, start + 2048)
                m = stop - start
                fn.buffer('uv').set_region(q, wi['uv'][start:stop], (np.s_[:m], np.s_[:2]), np.s_[:])
                fn.buffer('weights').set_region(q, wi['weights'][start:stop], np.s_[:m], np.s_[:])
                fn.grid(m)
        rms, nrms = fn.finalize()
        np.testing.assert_allclose(fn.buffer('grid').get(q), g[name + '_grid'], rtol=2e-6, atol=0)
        np.testing.assert_allclose(nrms, g[name + '_nrms'], rtol=1e-5)
        if rms is None:
            assert np.isnan(g[name + '_rms'])
        else:
            np.testing.assert_allclose(rms, g[name + '_rms'], rtol=1e-5)


def _clean_op(c, ci):
    from katsdpimager_amd import clean, parameters
    ctx, q = context_queue()
    fixed = parameters.FixedImageParameters(list(range(c['P'])), np.float32)
    ip = parameters.ImageParameters(fixed, 1.0, None, 0.2, None, pixel_size=1e-5,
                                    pixels=c['pixels'])
    cp = parameters.CleanParameters(1000, c['loop_gain'], 0.85, 5.0, c['mode'], 0.01, 0.5,
                                    c['border'])
    fn = clean.CleanTemplate(ctx, cp, np.float32, c['P']).instantiate(q, ip)
    fn.ensure_all_bound()
    fn.buffer('dirty').set(q, ci['dirty'])
    fn.buffer('psf').set(q, ci['psf'])
    fn.buffer('model').zero(q)
    return fn, q


@pytest.mark.parametrize('batched', [False, True])
@pytest.mark.parametrize('name', list(gi.CLEAN_CONFIGS))
def test_clean_vs_golden(golden, name, batched):
    """G7: CleanHost component lists -- positions, metric values, component fluxes and the
    residual image are BIT-EXACT, for the per-cycle API and the device-resident loop."""
    c = gi.CLEAN_CONFIGS[name]
    g = golden('g7_clean_' + name)
    ci = gi.clean_inputs(c)
    fn, q = _clean_op(c, ci)
    fn.reset()
    np.testing.assert_array_equal(fn.buffer('tile_max').get(q), g['tile_max0'])
    np.testing.assert_array_equal(fn.buffer('tile_pos').get(q), g['tile_pos0'])
    if batched:
        out = fn.run_cycles(ci['psf_patch'], c['threshold'], c['cycles'])
    else:
        out = []
        for i in range(c['cycles']):
            v, p, m = fn(ci['psf_patch'], c['threshold'])
            if v is None:
                break
            out.append((v, p, m))
    assert len(out) == len(g['values'])
    # 'true_pos' = pixels the reference actually subtracted at (its returned position is an
    # aliased view, see oracle Clean.__call__)
    np.testing.assert_array_equal(np.array([o[1] for o in out], np.int32), g['true_pos'])
    np.testing.assert_array_equal(np.array([o[0] for o in out], np.float32), g['values'])
    np.testing.assert_array_equal(np.array([o[2] for o in out], np.float32), g['pixels'])
    np.testing.assert_array_equal(fn.buffer('dirty').get(q), g['dirty_final'])
    np.testing.assert_array_equal(fn.buffer('model').get(q), g['model_final'])


def test_clean_steps_known():
    """test_clean.py:80-172: _UpdateTiles window semantics, _FindPeak, clipped _SubtractPsf."""
    from katsdpimager_amd import clean
    ctx, q = context_queue()
    image_shape = (4, 567, 456)
    border_pixels = 65
    border = border_pixels / image_shape[2]
    rs = np.random.RandomState(seed=1)
    fn = clean._UpdateTilesTemplate(ctx, np.float32, 4, clean.CLEAN_I).instantiate(
        q, image_shape, border)
    fn.ensure_all_bound()
    dirty = rs.standard_normal(image_shape).astype(np.float32)
    fn.buffer('dirty').set(q, dirty)
    fn.buffer('tile_max').zero(q)
    fn.buffer('tile_pos').zero(q)
    fn(135, 161, 385, 450)
    tile_max = fn.buffer('tile_max').get(q)
    tile_pos = fn.buffer('tile_pos').get(q)
    assert tile_max.shape == (14, 11)
    for y in range(14):
        for x in range(11):
            if x < 2 or x >= 10 or y < 3 or y >= 13:
                assert tile_max[y, x] == 0.0 and not tile_pos[y, x].any()
            else:
                y0, x0 = y * 32 + border_pixels, x * 32 + border_pixels
                y1 = min(y0 + 32, dirty.shape[1] - border_pixels)
                x1 = min(x0 + 32, dirty.shape[2] - border_pixels)
                tile = np.abs(dirty[0, y0:y1, x0:x1])
                pos = np.unravel_index(np.argmax(tile), tile.shape)
                assert tile[pos] == tile_max[y, x]
                assert (pos[0] + y0, pos[1] + x0) == tuple(tile_pos[y, x])

    image_shape, tile_shape = (4, 256, 256), (72, 67)
    fp = clean._FindPeakTemplate(ctx, np.float32, 4).instantiate(q, image_shape, tile_shape)
    fp.ensure_all_bound()
    dirty = rs.uniform(1.0, 2.0, image_shape).astype(np.float32)
    tmax = rs.uniform(1.0, 2.0, tile_shape).astype(np.float32)
    tmax[40, 3] = tmax[10, 50] = 2.5            # a tie: the first in row-major order wins
    tpos = np.array([[[y, x] for x in range(67)] for y in range(72)], np.int32)
    fp.buffer('tile_max').set(q, tmax)
    fp.buffer('tile_pos').set(q, tpos)
    fp.buffer('dirty').set(q, dirty)
    fp()
    assert fp.buffer('peak_value').get(q)[0] == np.float32(2.5)
    np.testing.assert_array_equal(fp.buffer('peak_pos').get(q), [10, 50])
    np.testing.assert_array_equal(fp.buffer('peak_pixel').get(q), dirty[:, 10, 50])

    loop_gain, image_shape, psf_patch, pos = 0.25, (4, 200, 344), (4, 72, 130), (170, 59)
    dirty = rs.standard_normal(image_shape).astype(np.float32)
    psf = rs.standard_normal(psf_patch).astype(np.float32)
    expected = dirty.copy()
    peak_pixel = dirty[:, pos[0], pos[1]].copy()
    expected[:, 134:200, 0:124] -= \
        (np.float32(loop_gain) * peak_pixel)[:, None, None] * psf[:, :66, 6:]
    psf_full = np.ones(image_shape, np.float32)
    psf_full[:, 64:136, 107:237] = psf
    sp = clean._SubtractPsfTemplate(ctx, np.float32, 4).instantiate(
        q, loop_gain, image_shape, image_shape)
    sp.ensure_all_bound()
    sp.buffer('dirty').set(q, dirty)
    sp.buffer('psf').set(q, psf_full)
    sp.buffer('peak_pixel').set(q, peak_pixel)
    sp.buffer('model').zero(q)
    sp(pos, psf_patch)
    np.testing.assert_array_equal(sp.buffer('dirty').get(q), expected)
    model = sp.buffer('model').get(q)
    np.testing.assert_array_equal(model[:, pos[0], pos[1]], np.float32(loop_gain) * peak_pixel)
    assert np.count_nonzero(model) == 4


def test_psf_patch_and_noise_vs_golden(golden):
    """G8 + test_clean.py:13-37, 174-200."""
    from katsdpimager_amd import clean
    ctx, q = context_queue()
    g = golden('g8_psfpatch_noise')
    fn = clean.PsfPatchTemplate(ctx, np.float32, 4).instantiate(q, (4, 206, 304))
    fn.ensure_all_bound()
    known = [(4, 1, 1), (4, 206, 304), (4, 205, 303), None, (4, 15, 5)]
    for i, (psf, thr, lim) in enumerate(gi.psf_patch_cases()):
        fn.buffer('psf').set(q, psf)
        box = fn(thr, lim)
        assert box == tuple(g['patch%d' % i])
        if known[i]:
            assert box == known[i]
    for i, (img, border) in enumerate(gi.noise_cases()):
        ne = clean.NoiseEstTemplate(ctx, np.float32, img.shape[0]).instantiate(q, img.shape, border)
        ne.ensure_all_bound()
        ne.buffer('dirty').set(q, img)
        assert ne() == g['noise%d' % i]            # exact median, as the host path
    # odd element count
    img = np.random.RandomState(4).standard_normal((1, 67, 59)).astype(np.float32)
    ne = clean.NoiseEstTemplate(ctx, np.float32, 1).instantiate(q, img.shape, 0.1)
    ne.ensure_all_bound()
    ne.buffer('dirty').set(q, img)
    assert ne() == orc.noise_est(img, 0.1)
    with pytest.raises(ValueError):
        clean.NoiseEstTemplate(ctx, np.float32, 1).instantiate(q, img.shape, 0.5)


@pytest.mark.parametrize('streams', [1, 2])
@pytest.mark.parametrize('batched', [False, True])
@pytest.mark.parametrize('name', list(gi.E2E_CONFIGS))
def test_end_to_end_vs_golden(golden, name, batched, streams):
    """G9: the whole per-channel loop (weights -> PSF -> 2 major cycles of grid / FFT / CLEAN /
    degrid-or-predict) on the Imaging facade vs the reference's ImagingHost."""
    from katsdpimager_amd import imaging, parameters, weight
    ctx, q = context_queue()
    c = gi.E2E_CONFIGS[name]
    g = golden('g9_e2e_' + name)
    ip, gp, ap = make_params(c)
    wp = parameters.WeightParameters(weight.WeightType(c['weight_type']), c['robustness'])
    cp = parameters.CleanParameters(c['minor'], c['loop_gain'], c['major_gain'], c['threshold'],
                                    c['mode'], c['psf_cutoff'], c['psf_limit'], c['border'])
    template = imaging.ImagingTemplate(ctx, ap, ip.fixed, wp, gp.fixed, cp)
    # streams=2: consecutive chunks alternate between two HIP streams (same results)
    im = template.instantiate(q, ip, gp, c['vis_block'], 0, c['major'], streams=streams)
    im.ensure_all_bound()
    data = gi.e2e_inputs(c)
    if batched:
        # route the minor loop through the device-resident batch API
        out = run_batched(im, c, data)
    else:
        out = gi.run_major_cycle(im, c, data, host=False)
    G = c['pixels']
    Gg = out['weights_grid'].shape[-1]
    np.testing.assert_allclose(out['weights_grid'],
                               gi.middle(g['weights_grid'], out['weights_grid'].shape),
                               rtol=1e-5, atol=0)
    np.testing.assert_allclose(out['weights_nrms'], g['weights_nrms'], rtol=1e-5)
    np.testing.assert_allclose(out['psf_peak'], g['psf_peak'], rtol=1e-5)
    assert tuple(out['psf_patch']) == tuple(g['psf_patch'])
    assert relerr(out['psf_core'], g['psf_core']) < 1e-5
    taper = kernel_taper(c)
    inner = np.s_[:, G // 8:-G // 8, G // 8:-G // 8]
    assert tapered_relerr(out['dirty0'], g['dirty0'], taper) < 1e-5
    assert relerr(out['dirty0'][inner], g['dirty0'][inner]) < 1e-4
    # the noise estimate is a median of |dirty|, i.e. 1-Lipschitz in the max-norm: two images within
    # 1e-5 of the peak of each other have estimates within 1e-5 of the peak (times 1.4826)
    assert abs(float(out['noise0']) - float(g['noise0'])) \
        <= max(1e-4 * float(g['noise0']), 1.4826e-5 * np.abs(g['dirty0']).max())
    np.testing.assert_array_equal(out['n_minor'], g['n_minor'])
    # first major cycle: identical peak sequence
    n0 = int(g['n_minor'][0])
    np.testing.assert_allclose(out['peak_values'][:n0], g['peak_values'][:n0], rtol=1e-4)
    np.testing.assert_array_equal(out['component_pos'], g['component_pos'])
    np.testing.assert_allclose(out['component_flux'], g['component_flux'], rtol=2e-4)
    assert relerr(out['residual_vis'], g['residual_vis']) < (1e-4 if c['degrid'] else 2e-3)
    assert tapered_relerr(out['dirty1'], g['dirty1'], taper) < 2e-4
    assert tapered_relerr(out['dirty_final'], g['dirty_final'], taper) < 2e-4
    assert relerr(out['dirty_final'][inner], g['dirty_final'][inner]) < 1e-3
    assert relerr(out['model_final'], g['model_final']) < 2e-4
    assert Gg <= G


def run_batched(im, c, data):
    """run_major_cycle with the minor loop replaced by Imaging.clean_cycles."""
    class Batched:
        def __init__(self, im):
            self._im = im
            self._queue = []

        def __getattr__(self, name):
            return getattr(self._im, name)

        @property
        def num_vis(self):
            return self._im.num_vis

        @num_vis.setter
        def num_vis(self, v):
            self._im.num_vis = v

        def clean_reset(self):
            self._queue = []
            self._im.clean_reset()

        def clean_cycle(self, psf_patch, threshold=0.0):
            if threshold == 0.0:
                return self._im.clean_cycle(psf_patch, threshold)
            if not self._queue:
                vals = self._im.clean_cycles(psf_patch, threshold, c['minor'] - 1)
                self._queue = list(vals) + [None]
            return self._queue.pop(0)
    return gi.run_major_cycle(Batched(im), c, data, host=False)


# ---- restoring beam (beam.py:204-398) ---------------------------------------
@pytest.mark.parametrize('case', range(3))
def test_convolve_beam_vs_golden(golden, case):
    """ConvolveBeam (R2C rocFFT -> fourier_beam kernel -> C2R) per polarization against the
    reference's host convolve_beam (G10) and the restatement; tolerance as test_beam.py:62
    (rtol 1e-5, atol 1e-5 of the peak)."""
    from katsdpimager_amd import beam
    ctx, q = context_queue()
    name, model, b = gi.beam_cases()[case]
    g = golden('g10_beam_' + name)
    bm = beam.Beam(b['amplitude'], b['x_stddev'], b['y_stddev'], b['theta'])
    np.testing.assert_allclose(beam.beam_covariance_sqrt(bm), g['cov_sqrt'], rtol=1e-13)
    template = beam.ConvolveBeamTemplate(ctx, model.shape[1:], model.dtype)
    fn = template.instantiate(q)
    with pytest.raises(ValueError):
        fn()                                    # beam not set (beam.py:284-285)
    fn.beam = bm
    fn.ensure_all_bound()
    peak = np.abs(g['restored']).max()
    ref = orc.convolve_beam(model, **b)
    for pol in range(model.shape[0]):
        fn.buffer('image').set(q, model[pol])
        fn()
        actual = fn.buffer('image').get(q)
        np.testing.assert_allclose(actual, g['restored'][pol], rtol=1e-5, atol=1e-5 * peak)
        np.testing.assert_allclose(actual, ref[pol], rtol=1e-5, atol=1e-5 * peak)


@pytest.mark.gpu
@pytest.mark.parametrize('G', [16, 64, 120, 126, 1000, 2048, 4096, 4800])
def test_convolve_beam_own_transform(G):
    """kimg_convolve_beam (square images of a size the library's own transforms take: rows, then
    per column forward transform x beam x inverse transform, then rows back) against the route on
    the FFT library's real <-> half-complex plans, and against the restatement at small sizes."""
    from katsdpimager_amd import beam
    ctx, q = context_queue()
    rs = np.random.RandomState(G)
    model = np.zeros((G, G), np.float32)
    for _ in range(50):
        model[rs.randint(G), rs.randint(G)] += rs.uniform(-1, 2)
    model += (0.01 * rs.standard_normal((G, G))).astype(np.float32)
    b = dict(amplitude=1.3, x_stddev=2.2, y_stddev=3.7, theta=0.6)
    out = {}
    for route, tuning in (('own', None), ('library', {'own_transform': False})):
        template = beam.ConvolveBeamTemplate(ctx, (G, G), np.float32, tuning=tuning)
        assert template.own_transform == (route == 'own')
        fn = template.instantiate(q)
        fn.beam = beam.Beam(**b)
        fn.ensure_all_bound()
        fn.buffer('image').set(q, model)
        fn()
        out[route] = fn.buffer('image').get(q)
        del fn
    peak = np.abs(out['library']).max()
    np.testing.assert_allclose(out['own'], out['library'], rtol=0, atol=3e-6 * peak)
    if G <= 128:
        ref = orc.convolve_beam(model[np.newaxis], **b)[0]
        np.testing.assert_allclose(out['own'], ref, rtol=1e-5, atol=1e-5 * peak)
    # a rectangular image stays with the FFT library
    assert not beam.ConvolveBeamTemplate(ctx, (96, 160), np.float32).own_transform


def test_restore_step():
    """frontend.py:623-641 on the facade buffers: model (x) beam + residuals."""
    from katsdpimager_amd import beam, imaging, parameters, weight
    ctx, q = context_queue()
    c = gi.E2E_CONFIGS['degrid']
    ip, gp, ap = make_params(c)
    wp = parameters.WeightParameters(weight.WeightType(c['weight_type']), c['robustness'])
    cp = parameters.CleanParameters(c['minor'], c['loop_gain'], c['major_gain'], c['threshold'],
                                    c['mode'], c['psf_cutoff'], c['psf_limit'], c['border'])
    im = imaging.ImagingTemplate(ctx, ap, ip.fixed, wp, gp.fixed, cp).instantiate(
        q, ip, gp, c['vis_block'], 0, c['major'])
    im.ensure_all_bound()
    G = c['pixels']
    rs = np.random.RandomState(5)
    model = np.zeros((1, G, G), np.float32)
    for _ in range(30):
        model[0, rs.randint(G), rs.randint(G)] += rs.uniform(0.1, 2)
    resid = (0.01 * rs.standard_normal((1, G, G))).astype(np.float32)
    im.set_buffer('model', model)
    im.set_buffer('dirty', resid)
    b = dict(amplitude=1.0, x_stddev=2.5, y_stddev=1.7, theta=0.3)
    conv = beam.restore(im, beam.Beam(**b))
    expected = orc.convolve_beam(model, **b) + resid
    got = im.get_buffer('dirty')
    np.testing.assert_allclose(got, expected, rtol=1e-5, atol=1e-5 * np.abs(expected).max())
    # the operator is reusable
    im.set_buffer('model', model)
    im.set_buffer('dirty', resid)
    assert beam.restore(im, beam.Beam(**b), conv) is conv
    np.testing.assert_allclose(im.get_buffer('dirty'), expected, rtol=1e-5,
                               atol=1e-5 * np.abs(expected).max())


@pytest.mark.gpu
def test_clear_dirty_deferred():
    """Imaging.clear_dirty defers its fill: a grid_to_image that can write the image (the
    library's own transforms, w = 0 or not) does so and neither fills nor reads it; any other
    access sees the zeros.  Same images either way."""
    from katsdpimager_amd import imaging, parameters, weight
    ctx, q = context_queue()
    c = gi.E2E_CONFIGS['degrid']
    ip, gp, ap = make_params(c)
    wp = parameters.WeightParameters(weight.WeightType(c['weight_type']), c['robustness'])
    cp = parameters.CleanParameters(c['minor'], c['loop_gain'], c['major_gain'], c['threshold'],
                                    c['mode'], c['psf_cutoff'], c['psf_limit'], c['border'])
    im = imaging.ImagingTemplate(ctx, ap, ip.fixed, wp, gp.fixed, cp).instantiate(
        q, ip, gp, c['vis_block'], 0, c['major'])
    im.ensure_all_bound()
    G = c['pixels']
    rs = gi.RandomState(77)
    shape = im.buffer('grid').shape
    im.set_buffer('grid', rs.complex_uniform(-1, 1, shape).astype(np.complex64))
    junk = np.full((1, G, G), 7.0, np.float32)
    zeros = np.zeros((1, G, G), np.float32)
    im.set_buffer('dirty', junk)
    im.clear_dirty()
    assert im._dirty_cleared
    np.testing.assert_array_equal(im.get_buffer('dirty'), zeros)       # (the fill happened now)
    assert not im._dirty_cleared
    for w in (0.0, 12.5):
        im.set_buffer('dirty', junk)
        im.clear_dirty()
        im.grid_to_image(w)
        assert not im._dirty_cleared
        assert im._grid_to_image.can_overwrite()
        deferred = im.get_buffer('dirty')
        im.set_buffer('dirty', zeros)
        im.grid_to_image(w)
        explicit = im.get_buffer('dirty')
        np.testing.assert_array_equal(deferred, explicit)
        assert np.abs(explicit).max() > 0
        im.grid_to_image(w)                         # the second call accumulates
        np.testing.assert_allclose(im.get_buffer('dirty'), 2 * explicit, rtol=1e-6,
                                   atol=1e-6 * np.abs(explicit).max())
    # the direct buffer access fills too
    im.set_buffer('dirty', junk)
    im.clear_dirty()
    np.testing.assert_array_equal(im.buffer('dirty').get(q), zeros)


@pytest.mark.parametrize('arith', ARITHS)
@pytest.mark.parametrize('seed', range(12))
def test_grid_degrid_fuzz(seed, arith):
    """Seeded random configurations (kernel width 1..64, oversampling, plane count incl. tables that
    do not fit LDS, polarizations, chunk sizes, track speed from static to teleporting) through the
    automatic gridder and degridder variants, against the oracle."""
    from katsdpimager_amd import grid
    ctx, q = context_queue()
    rs = gi.RandomState(1000 + seed)
    OV = int(rs.choice([2, 4, 8, 16]))
    K = int(rs.randint(1, 65))
    if (K * OV) % 2:
        K += 1
    K = min(K, 64)
    W = int(rs.choice([1, 2, 5, 16, 33, 64, 90, 200]))
    P = int(rs.randint(1, 5))
    n = int(rs.choice([1, 37, 64, 200, 1000, 2500]))
    pixels = int(rs.choice([192, 256, 384]))
    cover = int(pixels * 0.6) & ~1
    c = gi.make_config(pixels, 0.0001, 0.01, P, K, W, oversample=OV, grid_cover=cover, n_vis=n)
    t = gi.grid_track(c)
    # vary the locality: scale the step between consecutive visibilities
    speed = float(rs.choice([0.0, 0.3, 1.0, 4.0, 40.0]))
    half = t['weights_grid'].shape[-1] // 2 - 1
    uv = t['uv'].astype(np.int64)
    uv = uv[0] + np.round((uv - uv[0]) * speed).astype(np.int64)
    uv = ((uv + half) % (2 * half)) - half
    t['uv'] = uv.astype(np.int16)
    fn, q = _gridder(c, 'auto:' + arith, max_vis=4096)
    kernel = fn.convolve_kernel.data
    actual = _run_gridder(fn, q, t)
    expected = np.zeros(actual.shape, np.complex64)
    wg = np.zeros(actual.shape, np.float32)
    gi.middle(wg, t['weights_grid'].shape)[:] = t['weights_grid']
    orc.grid(kernel, expected, wg, t['uv'], t['sub_uv'], t['w_plane'], t['vis'])
    assert relerr(actual, expected) < GRID_TOL, (K, OV, W, P, n, speed)
    ip, gp, ap = make_params(c)
    dg = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed, {'arith': arith}).instantiate(q, ap, ip, gp, 4096)
    dg.ensure_all_bound()
    G = dg.buffer('grid').shape[-1]
    gdata = rs.complex_uniform(-1, 1, size=(P, G, G)).astype(np.complex64)
    dg.buffer('grid').set(q, gdata)
    vis = rs.complex_uniform(-1, 1, size=(n, P)).astype(np.complex64)
    w = rs.uniform(0.5, 1.5, size=(n, P)).astype(np.float32)
    dg.num_vis = n
    dg.buffer('uv').set_region(q, np.concatenate((t['uv'], t['sub_uv']), axis=1), np.s_[:n], np.s_[:])
    dg.buffer('w_plane').set_region(q, t['w_plane'], np.s_[:n], np.s_[:])
    dg.buffer('vis').set_region(q, vis, np.s_[:n], np.s_[:])
    dg.buffer('weights').set_region(q, w, np.s_[:n], np.s_[:])
    dg()
    want = vis.copy()
    orc.degrid(kernel, gdata, t['uv'], t['sub_uv'], t['w_plane'], w, want)
    got = dg.buffer('vis').get(q)[:n]
    assert np.abs(got - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-3), (K, OV, W, P, n, speed)


@pytest.mark.parametrize('seed', range(10))
def test_clean_fuzz(seed):
    """Seeded random CLEAN problems (non-square images, polarizations, both peak metrics, borders,
    patch sizes from tiny to larger than the image, thresholds that stop the loop early) through
    the device-resident loop: positions, values, fluxes and images BIT-EXACT against the restated
    CleanHost."""
    from katsdpimager_amd import clean, parameters
    ctx, q = context_queue()
    rs = np.random.RandomState(5000 + seed)
    P = int(rs.randint(1, 5))
    mode = int(rs.randint(0, 2))
    G = int(rs.choice([96, 144, 200, 256]))
    border = float(rs.choice([0.0, 0.02, 0.1, 0.2]))
    loop_gain = float(rs.choice([0.05, 0.1, 0.5]))
    g1 = np.exp(-0.5 * ((np.arange(G) - G // 2) / rs.uniform(1.5, 8.0)) ** 2)
    psf = np.empty((P, G, G), np.float32)
    for p in range(P):
        psf[p] = np.outer(g1, g1) + 0.01 * rs.standard_normal((G, G))
    psf /= psf[:, G // 2, G // 2][:, None, None]
    dirty = (0.3 * rs.standard_normal((P, G, G))).astype(np.float32)
    for _ in range(10):
        y, x = rs.randint(0, G, 2)
        dirty[:, y, x] += rs.uniform(2.0, 10.0, P).astype(np.float32) * rs.choice([-1, 1])
    ph = int(rs.choice([1, 5, 33, 63, G - 1 if (G - 1) % 2 else G - 2, G + 1 if (G + 1) % 2 else G + 2]))
    pw = int(rs.choice([1, 7, 31, 65, 127]))
    ph, pw = min(ph, G) | 1 if min(ph, G) < G else ph, min(pw, G)
    ph, pw = min(ph, G), min(pw, G)
    patch = (P, ph, pw)
    cycles = int(rs.choice([1, 40, 150]))
    fixed = parameters.FixedImageParameters(list(range(P)), np.float32)
    ip = parameters.ImageParameters(fixed, 1.0, None, 0.2, None, pixel_size=1e-5, pixels=G)
    cp = parameters.CleanParameters(1000, loop_gain, 0.85, 5.0, mode, 0.01, 0.5, border)
    fn = clean.CleanTemplate(ctx, cp, np.float32, P).instantiate(q, ip)
    fn.ensure_all_bound()
    fn.buffer('dirty').set(q, dirty)
    fn.buffer('psf').set(q, psf)
    fn.buffer('model').zero(q)
    fn.reset()
    ref_img = dirty.copy()
    ref_model = np.zeros_like(dirty)
    ref = orc.Clean(G, border, loop_gain, mode, ref_img, psf, ref_model)
    ref.reset()
    np.testing.assert_array_equal(fn.buffer('tile_max').get(q), ref._tile_max)
    first = float(np.max(ref._tile_max))
    threshold = float(rs.choice([0.0, 0.3 * first, 2.0 * first]))
    want = []
    for _ in range(cycles):
        v, pos, pix = ref(patch, threshold)
        if v is None:
            break
        want.append((v, ref.last_pos, np.array(pix)))
    got = fn.run_cycles(patch, threshold, cycles)
    assert len(got) == len(want), (P, mode, G, border, patch, threshold)
    for a, b in zip(got, want):
        assert a[0] == b[0] and tuple(a[1]) == tuple(b[1])
        np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(fn.buffer('dirty').get(q), ref_img)
    np.testing.assert_array_equal(fn.buffer('model').get(q), ref_model)


@pytest.mark.parametrize('C,P,mode,G,border', [(2, 1, 0, 256, 0.02), (4, 1, 0, 512, 0.02),
                                               (8, 1, 0, 320, 0.0), (3, 4, 1, 200, 0.1),
                                               (5, 2, 0, 144, 0.2)])
def test_clean_batch_matches_single_channels(C, P, mode, G, border):
    """kimg_clean_cycles_batch (cycle i of C channels in ONE launch, blockIdx.z = channel): every
    channel -- its own image, PSF, patch size, threshold and cycle limit; some stop early, one asks
    for fewer cycles -- gives BIT FOR BIT what the restated CleanHost gives for that channel alone
    (components, dirty, model) and what the single-channel loop leaves in the tile arrays; a second
    batched call continues where the first stopped."""
    from katsdpimager_amd import clean, parameters
    ctx, q = context_queue()
    rs = np.random.RandomState(900 + C * 10 + P)
    fixed = parameters.FixedImageParameters(list(range(P)), np.float32)
    ip = parameters.ImageParameters(fixed, 1.0, None, 0.2, None, pixel_size=1e-5, pixels=G)
    cp = parameters.CleanParameters(1000, 0.1, 0.85, 5.0, mode, 0.01, 0.5, border)
    template = clean.CleanTemplate(ctx, cp, np.float32, P)
    queues = [q] + [ctx.create_command_queue() for _ in range(C - 1)]
    fns, refs, patches, thresholds, cycles, singles = [], [], [], [], [], []
    for c in range(C):
        g1 = np.exp(-0.5 * ((np.arange(G) - G // 2) / rs.uniform(1.5, 6.0)) ** 2)
        psf = np.empty((P, G, G), np.float32)
        for p in range(P):
            psf[p] = np.outer(g1, g1) + 0.01 * rs.standard_normal((G, G))
        psf /= psf[:, G // 2, G // 2][:, None, None]
        dirty = (0.3 * rs.standard_normal((P, G, G))).astype(np.float32)
        for _ in range(12):
            y, x = rs.randint(0, G, 2)
            dirty[:, y, x] += rs.uniform(2.0, 10.0, P).astype(np.float32) * rs.choice([-1, 1])
        fn = template.instantiate(queues[c], ip)
        fn.ensure_all_bound()
        single = template.instantiate(queues[c], ip)
        single.ensure_all_bound()
        for op in (fn, single):
            op.buffer('dirty').set(queues[c], dirty)
            op.buffer('psf').set(queues[c], psf)
            op.buffer('model').zero(queues[c])
            op.reset()
        ref_img, ref_model = dirty.copy(), np.zeros_like(dirty)
        ref = orc.Clean(G, border, 0.1, mode, ref_img, psf, ref_model)
        ref.reset()
        first = float(np.max(ref._tile_max))
        patches.append((P, int(rs.choice([1, 9, 33, 65, 111])), int(rs.choice([1, 7, 31, 65, 133]))))
        patches[-1] = (P, min(patches[-1][1], G), min(patches[-1][2], G))
        thresholds.append(float(rs.choice([0.0, 0.0, 0.35 * first, 2.0 * first])))
        cycles.append(int(rs.choice([70, 150, 33])))
        fns.append(fn)
        singles.append(single)
        refs.append((ref, ref_img, ref_model))
    assert all(clean.batch_supported(f, p_) for f, p_ in zip(fns, patches))

    def check_round(got):
        for c in range(C):
            ref, ref_img, ref_model = refs[c]
            want = []
            for _ in range(cycles[c]):
                v, pos, pix = ref(patches[c], thresholds[c])
                if v is None:
                    break
                want.append((v, ref.last_pos, np.array(pix)))
            assert len(got[c]) == len(want), (c, patches[c], thresholds[c], cycles[c])
            for a, b in zip(got[c], want):
                assert a[0] == b[0] and tuple(a[1]) == tuple(b[1])
                np.testing.assert_array_equal(a[2], b[2])
            np.testing.assert_array_equal(fns[c].buffer('dirty').get(queues[c]), ref_img)
            np.testing.assert_array_equal(fns[c].buffer('model').get(queues[c]), ref_model)
            alone = singles[c].run_cycles(patches[c], thresholds[c], cycles[c])
            assert len(alone) == len(want)
            np.testing.assert_array_equal(fns[c].buffer('tile_max').get(queues[c]),
                                          singles[c].buffer('tile_max').get(queues[c]))
            np.testing.assert_array_equal(fns[c].buffer('tile_pos').get(queues[c]),
                                          singles[c].buffer('tile_pos').get(queues[c]))
    check_round(clean.run_cycles_batch(fns, patches, thresholds, cycles))
    # a second call goes on from the state the first left (lower thresholds, another queue)
    thresholds = [0.0] * C
    cycles = [40] * C
    check_round(clean.run_cycles_batch(fns, patches, thresholds, cycles, queues[-1]))
    with pytest.raises(ValueError):
        clean.run_cycles_batch(fns + fns, patches * 2, thresholds * 2, cycles * 2)
    big = (P, G, G)
    if not clean.batch_supported(fns[0], big):
        from katsdpimager_amd import _lib
        with pytest.raises(_lib.KimgError):
            clean.run_cycles_batch(fns[:2], [big, big], thresholds[:2], cycles[:2])


@pytest.mark.parametrize('seed', range(8))
def test_grid_image_weights_fuzz(seed):
    """Seeded random sizes through grid -> image, image -> grid and the weights pipeline against
    the oracle: image sizes that are not powers of two, grids smaller than the image, w != 0,
    off-centre lm bias, accumulation into a non-zero image; runs of equal cells and scattered
    cells in the weight scatter, all three weight types."""
    from katsdpimager_amd import image, weight
    ctx, q = context_queue()
    rs = gi.RandomState(7000 + seed)
    G = int(rs.choice([64, 96, 120, 160, 250, 256]))
    Gg = int(rs.choice([g for g in (16, 40, 64, 90, 120, 200, 256) if g <= G]))
    P = int(rs.randint(1, 5))
    w = float(rs.choice([0.0, 3.7, -12.5, 250.0]))
    lm_scale = float(rs.uniform(2e-4, 2e-3))
    lm_bias = -0.5 * G * lm_scale * float(rs.choice([1.0, 0.9, 1.1]))
    k1d = rs.uniform(0.5, 2.0, G).astype(np.float32)
    small = rs.complex_uniform(-1, 1, (P, Gg, Gg)).astype(np.complex64)
    start = rs.uniform(-1, 1, (P, G, G)).astype(np.float32)
    template = image.GridImageTemplate(ctx, np.float32)
    plan = template.make_fft_plan((G, G))
    g2i = template.instantiate_grid_to_image(q, (P, Gg, Gg), lm_scale, lm_bias, plan)
    g2i.ensure_all_bound()
    g2i.buffer('kernel1d').set(q, k1d)
    g2i.buffer('grid').set(q, small)
    g2i.buffer('image').set(q, start)
    g2i.set_w(w)
    g2i()
    full = np.zeros((P, G, G), np.complex64)
    gi.middle(full, small.shape)[:] = small
    expected = start.copy()
    orc.grid_to_image(full, expected, k1d, lm_scale, lm_bias, w)
    assert relerr(g2i.buffer('image').get(q), expected) < 1e-5, (G, Gg, P, w)
    i2g = template.instantiate_image_to_grid(q, (P, Gg, Gg), lm_scale, lm_bias, plan)
    i2g.bind(layer=g2i.buffer('layer'), kernel1d=g2i.buffer('kernel1d'))
    i2g.ensure_all_bound()
    model = rs.uniform(-1, 1, (P, G, G)).astype(np.float32)
    i2g.buffer('image').set(q, model)
    i2g.set_w(w)
    i2g()
    full_grid, _ = orc.image_to_grid(model, k1d, lm_scale, lm_bias, w)
    assert relerr(i2g.buffer('grid').get(q), gi.middle(full_grid, small.shape)) < 1e-5, (G, Gg, P, w)

    # weights: runs of equal cells (tracks) mixed with scattered cells, some zero weights
    n = int(rs.choice([1, 63, 500, 3000]))
    half = Gg // 2 - 1
    cells = rs.randint(-half, half, (n, 2))
    run = rs.randint(1, 12, n)
    uv = np.repeat(cells, run, axis=0)[:n].astype(np.int16)
    wts = rs.uniform(0.0, 2.0, (n, P)).astype(np.float32)
    wts[rs.uniform(size=n) < 0.1] = 0
    for wt in (weight.WeightType.NATURAL, weight.WeightType.UNIFORM, weight.WeightType.ROBUST):
        fn = weight.WeightsTemplate(ctx, wt, P).instantiate(q, (P, Gg, Gg), 4096)
        fn.ensure_all_bound()
        fn.robustness = 0.5
        fn.clear()
        ref = np.zeros((P, Gg, Gg), np.float32)
        if wt != weight.WeightType.NATURAL:
            fn.buffer('uv').set_region(q, uv, (np.s_[:n], np.s_[:2]), np.s_[:])
            fn.buffer('weights').set_region(q, wts, np.s_[:n], np.s_[:])
            fn.grid(n)
            orc.weights_grid_add(ref, uv, wts)
            np.testing.assert_allclose(fn.buffer('grid').get(q), ref, rtol=1e-5, atol=1e-6)
        rms, nrms = fn.finalize()
        if np.any(ref) or wt == weight.WeightType.NATURAL:
            e_rms, e_nrms = orc.weights_finalize(wt.value, ref, 0.5)
            np.testing.assert_allclose(fn.buffer('grid').get(q), ref, rtol=2e-5, atol=1e-6)
            np.testing.assert_allclose(nrms, e_nrms, rtol=1e-4)


def test_find_peak_and_totals():
    """frontend.find_peak on the recipe of test_frontend.py:8-27 at its full size (4096^2, 4
    polarizations; exact) and get_totals against the restatement."""
    from katsdpimager_amd import accel, beam, frontend
    ctx, q = context_queue()
    size, noise, peak = 4096, 15.0, 200.0
    x = np.linspace(-np.pi / 2, np.pi / 2, size)
    y = np.cos(x)
    pbeam = (y[np.newaxis, :] * y[:, np.newaxis]).astype(np.float32)
    pbeam[pbeam < 0.01] = np.nan
    rs = np.random.RandomState(seed=1)
    with np.errstate(invalid='ignore'):
        image = (rs.normal(scale=noise, size=(4, size, size)).astype(np.float32) / pbeam)
    d_image = accel.DeviceArray(ctx, image.shape, np.float32)
    d_pbeam = accel.DeviceArray(ctx, pbeam.shape, np.float32)
    d_pbeam.set(q, pbeam)
    d_image.set(q, image)
    assert np.isnan(frontend.find_peak(q, d_image, d_pbeam, noise))
    assert np.isnan(orc.find_peak(image, pbeam, noise))
    image[1, size // 2 + 5, size // 2 - 10] = peak
    d_image.set(q, image)
    assert frontend.find_peak(q, d_image, d_pbeam, noise) == peak
    image *= -1                                     # negative peaks
    d_image.set(q, image)
    assert frontend.find_peak(q, d_image, d_pbeam, noise) == peak
    # without a primary beam the noise itself qualifies: plain maximum above 7.5 sigma
    clean_img = np.where(np.isnan(image), 0, image).astype(np.float32)
    d_image.set(q, clean_img)
    assert frontend.find_peak(q, d_image, None, noise) == np.abs(clean_img).max()
    bm = beam.Beam(1.0, 2.0, 5.0, 1.0)
    d_image.set(q, image)
    got = frontend.get_totals(q, d_image, bm)
    want = orc.get_totals(image, bm.major, bm.minor)
    np.testing.assert_allclose(got, want, rtol=1e-9)
    psf = rs.uniform(size=(2, 64, 80)).astype(np.float32)
    d_psf = accel.DeviceArray(ctx, psf.shape, np.float32)
    d_psf.set(q, psf)
    np.testing.assert_array_equal(frontend.extract_psf(q, d_psf, (11, 7)), psf[0, 26:37, 36:43])


# ---- kernel table generated on the device (kimg_kernel_table) -------------------------------
def _table_close(dev, ref):
    """float32 rounding of a float64 result computed two ways: every tap within 1.5 float32 ulps
    of the table's peak magnitude... and, much tighter, of its own plane's scale."""
    scale = np.abs(ref).max(axis=(1, 2), keepdims=True)
    err = np.abs(dev - ref) / scale
    return float(err.max())


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(gi.KERNEL_CONFIGS))
def test_device_kernel_table_vs_golden(golden, name):
    """G1: the device-generated table against the table of the reference's ConvolutionKernel."""
    from katsdpimager_amd import grid
    c = gi.KERNEL_CONFIGS[name]
    g = golden('g1_kernel_' + name)
    ctx, q = context_queue()
    ip, gp, _ = make_params(c)
    kernel = grid.ConvolutionKernelDevice(ctx, ip, gp)
    assert kernel.beta == g['beta']
    data = kernel.data
    assert data.shape == g['data'].shape and data.dtype == np.complex64
    # tolerance: 1e-7 of the plane's peak (float32 eps is 1.2e-7; the float64 sums differ by ~1e-14)
    assert _table_close(data, g['data']) <= 1e-7
    same = np.mean((data.real == g['data'].real) & (data.imag == g['data'].imag))
    assert same > 0.95          # all but rounding-boundary cases are bit-identical
    np.testing.assert_array_equal(kernel.taper(c['pixels']), g['taper'])


@pytest.mark.gpu
@pytest.mark.parametrize('K,W,OV', [(60, 400, 8), (28, 32, 8), (64, 37, 8), (7, 5, 6), (160, 3, 8),
                                    (33, 9, 4)])
def test_device_kernel_table_vs_oracle(K, W, OV):
    """Sizes of the real defaults (hundreds of planes, width 60) and the extremes of the LDS budget."""
    from katsdpimager_amd import grid
    c = gi.make_config(512, 2.0e-5, 0.21, 1, K, W, w_slices=2, max_w=900.0)
    c['oversample'] = OV
    ctx, q = context_queue()
    ip, gp, _ = make_params(c)
    kernel = grid.ConvolutionKernelDevice(ctx, ip, gp)
    ref, beta = orc.convolution_kernel(c['cell_size'], c['wavelength'], c['max_w'], c['w_slices'],
                                       W, OV, K, c['antialias_width'], c['image_oversample'])
    assert kernel.beta == beta
    assert _table_close(kernel.data, ref) <= 1e-7


@pytest.mark.gpu
def test_device_kernel_table_arguments():
    from katsdpimager_amd import accel, grid
    from katsdpimager_amd._lib import lib
    ctx, q = context_queue()
    table = accel.DeviceArray(ctx, (2, 8, 28), np.complex64)
    ws = accel.DeviceArray(ctx, (2,), np.float64)
    ws.set(q, np.zeros(2))
    call = lib().kimg_kernel_table
    assert call(None, ws.ptr, 2, 28, 8, 4, 2.0, 7.0, 12.6, q.handle) == -10001
    assert call(table.ptr, ws.ptr, 2, 27, 7, 4, 2.0, 7.0, 12.6, q.handle) == -10001   # odd OV*K
    assert call(table.ptr, ws.ptr, 2, 200, 8, 4, 2.0, 7.0, 12.6, q.handle) == -10002  # LDS budget
    # beyond the device generator's size: host construction as in the reference, same result type
    c = gi.make_config(512, 2.0e-5, 0.21, 1, 162, 2, w_slices=1, max_w=100.0)
    ip, gp, _ = make_params(c)
    kernel = grid.ConvolutionKernelDevice(ctx, ip, gp)
    ref, _ = orc.convolution_kernel(c['cell_size'], c['wavelength'], c['max_w'], 1, 2, 8, 162,
                                    c['antialias_width'], c['image_oversample'])
    np.testing.assert_array_equal(kernel.data, ref)
    c = gi.make_config(512, 2.0e-5, 0.21, 1, 27, 2, w_slices=1, max_w=100.0)
    c['oversample'] = 7
    ip, gp, _ = make_params(c)
    with pytest.raises(ValueError):
        grid.ConvolutionKernelDevice(ctx, ip, gp)


@pytest.mark.gpu
@pytest.mark.parametrize('G,P,mode,border,patch', [
    (1512, 1, 0, 0.02, (111, 133)),      # 46 x 46 tiles: 2 x 2 groups of 32 x 32 tiles per thread
    (1200, 4, 1, 0.0, (65, 97)),         # sum of squares over 4 polarizations, no border
    (2592, 1, 0, 0.1, (31, 31)),         # 3 x 3 groups, wide border, tiny patch
    (1120, 2, 0, 0.02, (225, 193)),      # 9 x 8 = 72 lattice blocks, more than one per tile column group
    (1024, 1, 1, 0.05, (65, 97)),        # one polarization, sum of squares (one-workgroup form: pixel loads)
    (4096, 1, 0, 0.02, (191, 161)),      # 7 x 7 lattice blocks, 14 884 tiles: the PSF patch does not fit LDS
    (1024, 1, 0, 0.02, (65, 225)),       # tall patch: a wave of the one-workgroup form has two chunks of rows
    (4096, 1, 0, 0.02, (711, 675)),      # 24 x 23 = 552 lattice blocks: more than CUs, every form ends up with two launches
    (1008, 1, 0, 0.013, (65, 97)),       # border 13: groups of four pixels straddle the edges of image and lattice
])
def test_clean_one_launch_cycle_matches_two_launch(G, P, mode, border, patch):
    """The one-launch-per-cycle form (every workgroup repeats the peak search, tile records in
    flight as deltas) against the two-launch form on the same problem: log, images and the tile
    arrays left behind are bit-identical, over consecutive calls and mixed with single cycles."""
    import os
    from katsdpimager_amd import clean, parameters
    ctx, q = context_queue()
    rs = np.random.RandomState(G + P)
    g1 = np.exp(-0.5 * ((np.arange(G) - G // 2) / 6.0) ** 2)
    psf = np.repeat((np.outer(g1, g1) + 0.02 * np.cos(np.arange(G) / 7.0)[None, :] * g1[:, None])
                    [None].astype(np.float32), P, axis=0)
    dirty = (0.05 * rs.standard_normal((P, G, G))).astype(np.float32)
    for _ in range(60):
        y, x = rs.randint(0, G, 2)
        dirty[:, y, x] += rs.uniform(1.0, 5.0, P).astype(np.float32) * rs.choice([-1, 1])
    dirty[:, 3, 5] = 40.0                # inside the border zone: must never be picked
    dirty[:, G // 2, G // 2] = dirty[:, G // 2 + 40, G // 2 - 64] = 7.0     # an exact tie
    fixed = parameters.FixedImageParameters(list(range(P)), np.float32)
    ip = parameters.ImageParameters(fixed, 1.0, None, 0.2, None, pixel_size=1e-5, pixels=G)
    cp = parameters.CleanParameters(1000, 0.1, 0.85, 5.0, mode, 0.01, 0.5, border)
    full_patch = (P,) + patch

    def run(form):
        fn = clean.CleanTemplate(ctx, cp, np.float32, P, {'form': form}).instantiate(q, ip)
        fn.ensure_all_bound()
        fn.buffer('dirty').set(q, dirty)
        fn.buffer('psf').set(q, psf)
        fn.buffer('model').zero(q)
        fn.reset()
        log = fn.run_cycles(full_patch, 0.0, 150)
        log.append(fn(full_patch, 0.0))                  # a single cycle on the tiles left behind
        log += fn.run_cycles(full_patch, 0.0, 37)        # odd count, below the graph size
        first = log[0][0]
        log += fn.run_cycles(full_patch, 0.7 * first, 500)        # stops at the threshold
        # a different threshold replays the same captured graph (the threshold lives in device
        # state, not in the kernel arguments)
        log += fn.run_cycles(full_patch, 0.5 * first, 500)
        return (log, fn.buffer('dirty').get(q), fn.buffer('model').get(q),
                fn.buffer('tile_max').get(q), fn.buffer('tile_pos').get(q))
    b = run('two_launch')
    for form in ('one_launch', 'persistent', 'one_workgroup', 'auto', 'multi'):
        a = run(form)
        assert len(a[0]) == len(b[0]) and 188 <= len(a[0]) < 1188
        for u, w in zip(a[0], b[0]):
            assert u[0] == w[0] and tuple(u[1]) == tuple(w[1])
            np.testing.assert_array_equal(u[2], w[2])
        for u, w in zip(a[1:], b[1:]):
            np.testing.assert_array_equal(u, w)
        assert border == 0.0 or (3, 5) not in [tuple(e[1]) for e in a[0]]


@pytest.mark.gpu
@pytest.mark.parametrize('H,W,P,border', [(96, 96, 1, 0.0), (97, 64, 1, 0.0), (200, 144, 4, 0.1),
                                          (255, 255, 1, 0.02), (64, 1000, 2, 0.05)])
def test_noise_est_device_selection(H, W, P, border):
    """kimg_noise_est (radix-select bytes chosen on the device, no host round trips) is the exact
    median * 1.4826 in float32: equal to numpy's median of |x| inside the border, and to the
    host-driven selection, for odd and even sample counts, ties, zeros, huge and tiny values."""
    from katsdpimager_amd import clean
    ctx, q = context_queue()
    rs = np.random.RandomState(H * 7 + W + P)
    for case in range(7):
        img = rs.standard_normal((P, H, W)).astype(np.float32)
        if case == 4:
            # the two middle elements part at the first byte: exactly half of the interior is zero
            bp = int(round(border * min(H, W)))
            inner = img[:, bp:H - bp, bp:W - bp]
            flat = np.abs(inner).ravel() + np.float32(1e3)
            flat[rs.permutation(flat.size)[:flat.size // 2]] = 0.0
            inner[...] = flat.reshape(inner.shape)
        elif case == 5:
            img[...] = np.float32(-2.5)                # one value only
        elif case == 6:
            # ... and at the last byte: two neighbouring bit patterns, half and half
            vals = np.array([1.0, np.nextafter(np.float32(1.0), np.float32(2.0))], np.float32)
            img = vals[rs.randint(0, 2, img.shape)] * rs.choice([-1, 1], img.shape).astype(np.float32)
        elif case == 1:
            img = np.round(img * 3) / 3               # many exact ties around the median
        elif case == 2:
            img[rs.random_sample(img.shape) < 0.6] = 0.0       # the median itself is zero
        elif case == 3:
            img *= np.float32(10.0) ** rs.randint(-30, 30, img.shape).astype(np.float32)
        op = clean.NoiseEstTemplate(ctx, np.float32, P).instantiate(q, (P, H, W), border)
        op.ensure_all_bound()
        op.buffer('dirty').set(q, img)
        got = op()
        bp = op.border_pixels
        inner = np.abs(img[:, bp:H - bp, bp:W - bp])
        want = np.median(inner) * np.float32(clean._MEDIAN_TO_RMS)
        assert got.dtype == np.float32
        assert got == np.float32(want), (case, got, want)
        assert got == op.host_select()


@pytest.mark.gpu
@pytest.mark.parametrize('P,K,W', [(1, 28, 32), (2, 28, 32), (1, 28, 160), (1, 45, 16), (4, 60, 96)])
@pytest.mark.parametrize('pattern', ['1e-20', '1e-6', '1', '1e6', '1e20', 'jump_up', 'jump_down',
                                     'alternate', 'zeros', 'one_in_zeros', 'nan'])
def test_gridder_f16_form_ranges(pattern, P, K, W):
    """The fp16 hi/lo form of the window gridder (two visibilities per matrix instruction) keeps its
    operands in fp16 range with power-of-two scales chosen on the fly: table scale from the largest
    tap, sample scale per wave, re-chosen (after a flush) when a larger sample arrives.  Whatever
    the magnitudes -- tiny, huge, jumping by 10^12 either way, mixed, mostly zero -- the result
    stays within the 1e-5 gate of the exact-fp32 form and of the oracle.  (P, K, W) pick the forms:
    table in LDS, two polarizations per launch, table in HBM, 2 x 2 tap blocks of a wide kernel.)"""
    import os
    c = gi.make_config(512, 0.0001, 0.01, P, K, W, grid_cover=300, n_vis=1200)
    t = gi.grid_track(c)
    n = len(t['uv'])
    rs = np.random.RandomState(7)
    scale = np.ones(n, np.float32)
    if pattern == 'jump_up':
        scale[:] = 1e-6
        scale[n // 2 + 13:] = 1e6
    elif pattern == 'jump_down':
        scale[:] = 1e6
        scale[n // 3 + 5:] = 1e-6
    elif pattern == 'alternate':
        scale = (10.0 ** rs.randint(-8, 9, n)).astype(np.float32)
    elif pattern == 'zeros':
        scale[:] = 0.0
    elif pattern == 'one_in_zeros':
        scale[:] = 0.0
        scale[777] = 3.0e4
    elif pattern != 'nan':
        scale[:] = float(pattern)
    t = dict(t)
    t['vis'] = (t['vis'] * scale[:, None]).astype(np.complex64)
    if pattern == 'nan':
        # non-finite samples spoil their own footprint only, in both forms
        t['vis'][100, 0] = np.nan
        t['vis'][700, -1] = complex(np.inf, 1.0)

    def run(arith):
        fn, q = _gridder(c, 'mfma:' + arith, max_vis=2048)
        return _run_gridder(fn, q, t), fn.convolve_kernel.data
    exact, kernel = run('fp32')
    split, _ = run('split_fp16')
    if pattern == 'nan':
        bad = ~np.isfinite(exact)
        assert 0 < bad.sum() < 0.2 * exact.size
        np.testing.assert_array_equal(~np.isfinite(split), bad)
        peak = np.abs(exact[~bad]).max()
        assert np.abs(split[~bad] - exact[~bad]).max() <= 2e-6 * peak
        return
    assert np.all(np.isfinite(split))
    peak = np.abs(exact).max()
    if peak == 0:
        assert np.abs(split).max() == 0
        return
    assert np.abs(split - exact).max() <= 2e-6 * peak
    if pattern not in ('1', 'jump_up', 'alternate'):
        return              # (the oracle comparison of one pattern per kind is enough)
    want = np.zeros(exact.shape, np.complex64)
    wg = np.zeros(exact.shape, np.float32)
    gi.middle(wg, t['weights_grid'].shape)[:] = t['weights_grid']
    orc.grid(kernel, want, wg, t['uv'], t['sub_uv'], t['w_plane'], t['vis'])
    assert relerr(split, want) < GRID_TOL


@pytest.mark.gpu
@pytest.mark.parametrize('W', [32, 64, 160])          # doubled rows, single rows, table in HBM
@pytest.mark.parametrize('pattern', ['1e-20', '1', '1e20', 'spike', 'smooth_ramp', 'zeros', 'nan'])
def test_degridder_f16_form_ranges(pattern, W):
    """The fp16 hi/lo form of the window degridder (two window rows per matrix instruction) scales
    every window it loads by a power of two chosen from the window's largest value, and the taps
    by one chosen from the table.  Model grids of any magnitude, with a 10^9 spike next to ordinary
    values, with a steep ramp, all zero, or with NaN cells stay within 2e-6 (of the largest
    prediction) of the exact-fp32 form; NaN reaches the same visibilities in both."""
    import os
    from katsdpimager_amd import grid
    ctx, q = context_queue()
    c = gi.make_config(512, 0.0001, 0.01, 1, 28, W, grid_cover=300, n_vis=1500)
    t = gi.grid_track(c)
    ip, gp, ap = make_params(c)
    n = c['n_vis']
    rs = np.random.RandomState(11)
    fns = {arith: grid.DegridderTemplate(ctx, ip.fixed, gp.fixed, {'arith': arith}).instantiate(
        q, ap, ip, gp, 2048) for arith in ARITHS}
    for fn in fns.values():
        fn.ensure_all_bound()
    shape = fn.buffer('grid').shape
    model = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    if pattern == 'spike':
        # (on the track, so that windows really hold the spike next to ordinary values)
        model[0, int(t['uv'][n // 2, 1]) + shape[1] // 2 + 3, int(t['uv'][n // 2, 0]) + shape[2] // 2 - 5] = 1e9
    elif pattern == 'smooth_ramp':
        model *= (10.0 ** np.linspace(-12, 12, shape[2])).astype(np.float32)[None, None, :]
    elif pattern == 'zeros':
        model[:] = 0
    elif pattern == 'nan':
        model[0, shape[1] // 2, shape[2] // 2] = np.nan
    else:
        model *= np.float32(float(pattern))
    vis0 = (rs.standard_normal((n, 1)) + 1j * rs.standard_normal((n, 1))).astype(np.complex64)
    weights = rs.uniform(0.5, 2.0, (n, 1)).astype(np.float32)

    def run(arith):
        fn = fns[arith]
        fn.buffer('grid').set(q, model)
        fn.num_vis = n
        fn.buffer('uv').set_region(q, np.concatenate((t['uv'], t['sub_uv']), axis=1), np.s_[:n], np.s_[:])
        fn.buffer('w_plane').set_region(q, t['w_plane'], np.s_[:n], np.s_[:])
        fn.buffer('vis').set_region(q, vis0, np.s_[:n], np.s_[:])
        fn.buffer('weights').set_region(q, weights, np.s_[:n], np.s_[:])
        fn()
        return fn.buffer('vis').get(q)[:n]
    exact, split = run('fp32'), run('split_fp16')
    bad = ~np.isfinite(exact)
    np.testing.assert_array_equal(~np.isfinite(split), bad)
    assert (pattern == 'nan') == bool(bad.any())
    pred = np.abs(exact[~bad] - vis0[~bad])
    scale = max(float(pred.max()), 1e-30) if pred.size else 1.0
    # (plus one rounding of the stored residual, which is of the size of the visibility)
    assert np.abs(split[~bad] - exact[~bad]).max() <= 2e-6 * scale + 2.4e-7 * np.abs(exact[~bad]).max()


@pytest.mark.gpu
@pytest.mark.parametrize('arith', ARITHS)
@pytest.mark.parametrize('K,W,P', [(28, 32, 1), (8, 8, 2), (32, 16, 1), (60, 16, 1), (45, 160, 3)])
def test_gridder_binned_variant(K, W, P, arith):
    """KIMG_VARIANT_BINNED (device sort by grid tile + the window kernel) on a stream with no
    locality -- a smooth track and scattered positions, shuffled together -- against the oracle, the
    direct window kernel and the per-tap kernel; kimg_grid_jumps counts exactly the records whose
    cell moved by more than the window slack."""
    from katsdpimager_amd import grid
    c = gi.make_config(512, 0.0001, 0.01, P, K, W, grid_cover=300, n_vis=3000)
    t = gi.grid_track(c)
    rs = gi.RandomState(K + W)
    half = t['weights_grid'].shape[-1] // 2 - 1
    n2 = 2000
    data = dict(
        uv=np.concatenate([t['uv'], rs.randint(-half, half, (n2, 2)).astype(np.int16)]),
        sub_uv=np.concatenate([t['sub_uv'], rs.randint(0, 8, (n2, 2)).astype(np.int16)]),
        w_plane=np.concatenate([t['w_plane'], rs.randint(0, W, n2).astype(np.int16)]),
        vis=np.concatenate([t['vis'], rs.complex_uniform(-1, 1, size=(n2, P)).astype(np.complex64)]),
        weights_grid=t['weights_grid'])
    order = np.random.RandomState(7).permutation(len(data['uv']))
    for k in ('uv', 'sub_uv', 'w_plane', 'vis'):
        data[k] = np.ascontiguousarray(data[k][order])
    fb, q = _gridder(c, 'binned:' + arith, max_vis=8192)
    binned = _run_gridder(fb, q, data)
    assert fb.last_variant == 'binned' and fb._workspace_bytes >= fb._binned_bytes > 0
    expected = np.zeros(binned.shape, np.complex64)
    wg = np.zeros(binned.shape, np.float32)
    gi.middle(wg, data['weights_grid'].shape)[:] = data['weights_grid']
    orc.grid(fb.convolve_kernel.data, expected, wg, data['uv'], data['sub_uv'], data['w_plane'],
             data['vis'])
    assert relerr(binned, expected) < GRID_TOL
    fd, _ = _gridder(c, 'mfma:' + arith, max_vis=8192)
    assert relerr(_run_gridder(fd, q, data), binned) < GRID_TOL
    # the jump count: by definition
    slack = 32 - (K if K <= 32 else (K + 1) // 2)
    d = np.abs(np.diff(data['uv'].astype(np.int32), axis=0))
    want_jumps = int(np.count_nonzero((d[:, 0] > slack) | (d[:, 1] > slack)))
    assert fd.jump_fraction() == want_jumps / len(data['uv'])
    # a second call reuses the scratch; fewer visibilities than max_vis are fine
    sub = {k: (v[:1234] if k != 'weights_grid' else v) for k, v in data.items()}
    exp2 = np.zeros(binned.shape, np.complex64)
    orc.grid(fb.convolve_kernel.data, exp2, wg, sub['uv'], sub['sub_uv'], sub['w_plane'], sub['vis'])
    assert relerr(_run_gridder(fb, q, sub), exp2) < GRID_TOL


@pytest.mark.gpu
def test_gridder_auto_variant_follows_the_stream():
    """`auto` measures the stream on the device: a track goes to the window kernel as it is, the
    same visibilities shuffled are binned first, calls below AUTO_MIN_VIS are never measured, and
    locality_hint overrides the measurement; every route gives the same grid."""
    from katsdpimager_amd import grid
    c = gi.make_config(512, 0.0001, 0.01, 1, 28, 32, grid_cover=300, n_vis=70000)
    t = gi.grid_track(c)
    fn, q = _gridder(c, 'auto', max_vis=70000)
    direct = _run_gridder(fn, q, t)
    assert fn.last_variant == 'mfma' and fn.jump_fraction() < 0.03
    order = np.random.RandomState(3).permutation(c['n_vis'])
    shuffled = {k: (np.ascontiguousarray(v[order]) if k != 'weights_grid' else v) for k, v in t.items()}
    got = _run_gridder(fn, q, shuffled)
    assert fn.last_variant == 'binned' and fn.jump_fraction() > 0.9
    assert relerr(got, direct) < GRID_TOL
    fn.locality_hint = True                 # the caller insists: window kernel on the shuffled stream
    assert relerr(_run_gridder(fn, q, shuffled), direct) < GRID_TOL and fn.last_variant == 'mfma'
    fn.locality_hint = False
    assert relerr(_run_gridder(fn, q, t), direct) < GRID_TOL and fn.last_variant == 'binned'
    fn.locality_hint = None
    small = {k: (v[:grid.AUTO_MIN_VIS - 1] if k != 'weights_grid' else v) for k, v in shuffled.items()}
    _run_gridder(fn, q, small)
    assert fn.last_variant == 'mfma'
    with pytest.raises(ValueError):
        grid.GridderTemplate(None, fn.image_parameters.fixed, fn.grid_parameters.fixed, {'variant': 'fast'})


@pytest.mark.gpu
@pytest.mark.parametrize('P,K,W', [(1, 28, 32), (2, 28, 32), (1, 28, 160), (1, 60, 16)])
@pytest.mark.parametrize('pattern', ['jump_down', 'jump_up', 'spikes', 'ramp', 'pol_ratio'])
def test_gridder_f16_form_small_samples_per_cell(pattern, P, K, W):
    """ADVICE r1 (grid_mfma.hip sample scale): small samples must not be judged against the peak of
    large ones gridded by the same wave.  The grid of the SMALL samples alone (exact form) is the
    reference: on the cells that no large sample touches, the split form of the full input must
    reproduce it to 2e-5 of ITS OWN largest value -- after a 10^12 drop, before a 10^12 rise,
    between isolated 10^9 spikes, along a smooth ramp over 12 decades (every cell against its own
    neighbourhood), and (two polarizations per launch) with one polarization 10^-7 of the other.
    (What the form cannot do: a sample's error floor is 2^-40 of its OWN peak contribution, so the
    far corners of a strong sample's footprint can disturb a 10^8 times weaker neighbour.)"""
    c = gi.make_config(512, 0.0001, 0.01, P, K, W, grid_cover=300, n_vis=1500)
    t = dict(gi.grid_track(c))
    n = len(t['uv'])
    rs = np.random.RandomState(11)
    scale = np.ones((n, P), np.float32)
    large = np.zeros(n, bool)
    if pattern == 'jump_down':
        large[:n // 3 + 5] = True
    elif pattern == 'jump_up':
        large[n // 2 + 13:] = True
    elif pattern == 'spikes':
        large[rs.choice(n, 12, replace=False)] = True
    elif pattern == 'ramp':
        # magnitudes falling smoothly by 12 decades along the track: every cell is compared
        # with its own neighbourhood (below), not with the start of the track
        scale *= (10.0 ** np.linspace(6, -6, n))[:, None].astype(np.float32)
    if pattern == 'pol_ratio':
        if P == 1:
            pytest.skip('needs two polarizations in one launch')
        scale[:, 1] = 1e-7          # the weak polarization is "small" everywhere: checked below
    elif pattern != 'ramp':
        scale[large] = 1e6 if pattern != 'spikes' else 1e9
        scale[~large] = 1e-6 if pattern != 'spikes' else 1.0
    vis = (t['vis'] * scale).astype(np.complex64)

    def run(arith, v):
        fn, q = _gridder(c, 'mfma:' + arith, max_vis=2048)
        return _run_gridder(fn, q, dict(t, vis=v))
    split = run('split_fp16', vis)
    if pattern == 'pol_ratio':
        exact = run('fp32', vis)
        for p in range(P):
            peak = np.abs(exact[p]).max()
            assert np.abs(split[p] - exact[p]).max() <= 2e-6 * peak, p
        return
    if pattern == 'ramp':
        from scipy import ndimage
        exact = run('fp32', vis)
        local = ndimage.maximum_filter(np.abs(exact), size=(1, 2 * K + 1, 2 * K + 1))
        touched = local > 0
        assert np.all(np.abs(split - exact)[touched] <= 2e-5 * local[touched])
        assert local[touched].min() < 1e-9 * local.max()        # the ramp really spans the decades
        return
    small_only = vis.copy()
    small_only[large] = 0
    large_only = vis.copy()
    large_only[~large] = 0
    ref_small = run('fp32', small_only)
    touched_by_large = run('fp32', large_only) != 0
    cells = (~touched_by_large) & (ref_small != 0)
    assert cells.sum() > 200            # (the track moves on: cells that see only small samples)
    local_peak = np.abs(ref_small[cells]).max()
    assert np.abs(split[cells] - ref_small[cells]).max() <= 2e-5 * local_peak
    # and the whole grid still meets the global gate
    exact = run('fp32', vis)
    assert np.abs(split - exact).max() <= 2e-6 * np.abs(exact).max()


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(gi.E2E_CONFIGS))
def test_dirty_image_vs_fp64_truth(golden, name):
    """the image-plane gates against a float64 TRUTH instead of only against
    each other.  The first dirty image of the G9 configurations (weights -> grid -> inverse FFT ->
    taper division -> PSF normalisation) is evaluated in float64 from the same records, once with
    this package's imaging weights and once with the reference's, and compared with the HIP image and
    with the reference's float32 ImagingHost image (the golden):
      * where outer(taper, taper) >= 1e-2 of its peak: both within 1e-5 of the truth, UNWEIGHTED
        (north_star's tolerance), relative to the truth's maximum;
      * everywhere: both within 1e-5 in the taper-weighted metric (the quantity the FFT computes);
      * in the badly conditioned rest (the division amplifies any float32 FFT's rounding by up to
        1 / taper^2 ~ 10^5): the HIP image is no farther from the truth than 2x the reference's own
        float32 host path plus 1e-5 -- "equally far from the truth", measured."""
    from katsdpimager_amd import imaging, parameters, weight
    from helpers import grid_to_image_truth, grid_truth_numpy, taper_zones
    ctx, q = context_queue()
    c = gi.E2E_CONFIGS[name]
    g = golden('g9_e2e_' + name)
    ip, gp, ap = make_params(c)
    wp = parameters.WeightParameters(weight.WeightType(c['weight_type']), c['robustness'])
    cp = parameters.CleanParameters(c['minor'], c['loop_gain'], c['major_gain'], c['threshold'],
                                    c['mode'], c['psf_cutoff'], c['psf_limit'], c['border'])
    im = imaging.ImagingTemplate(ctx, ap, ip.fixed, wp, gp.fixed, cp).instantiate(
        q, ip, gp, c['vis_block'], 0, c['major'])
    im.ensure_all_bound()
    data = gi.e2e_inputs(c)
    out = gi.run_major_cycle(im, c, data, host=False)
    G = c['pixels']
    kernel = im._gridder.convolve_kernel.data
    taper = kernel_taper(c)
    slice_w_step = float(c['max_w'] / c['wavelength'] / (c['w_slices'] - 0.5))
    mid_w = np.arange(c['w_slices']) * slice_w_step
    lm_scale = float(ip.pixel_size)
    lm_bias = -0.5 * G * lm_scale

    def truth(weights_grid):
        Gg = weights_grid.shape[-1]
        lo = (G - Gg) // 2

        def dirty(field):
            image = np.zeros((c['P'], G, G), np.float64)
            for s, rec in enumerate(data['slices']):
                if len(rec) == 0:
                    continue
                vis = np.asarray(rec[field]).astype(np.complex128).reshape(len(rec), c['P'])
                grid_ = grid_truth_numpy(kernel, np.asarray(rec.uv), np.asarray(rec.sub_uv),
                                         np.asarray(rec.w_plane), vis, weights_grid)
                full = np.zeros((c['P'], G, G), np.complex128)
                full[:, lo:lo + Gg, lo:lo + Gg] = grid_
                image += grid_to_image_truth(full, taper, lm_scale, lm_bias, mid_w[s])
            return image
        psf = dirty('weights')
        return dirty('vis') / psf[:, G // 2, G // 2][:, None, None]

    truth_ours = truth(out['weights_grid'].astype(np.float64))
    truth_ref = truth(gi.middle(g['weights_grid'], out['weights_grid'].shape).astype(np.float64))
    good, bad, t2 = taper_zones(taper)
    assert 0.2 < good.mean() < 0.95 and t2[bad].min() < 1e-4 * t2.max()
    peak = np.abs(truth_ours).max()
    err_hip = np.abs(out['dirty0'] - truth_ours)
    err_ref = np.abs(g['dirty0'] - truth_ref)
    # well-conditioned zone: the plain 1e-5 gate, unweighted, for both
    assert err_hip[:, good].max() <= 1e-5 * peak
    assert err_ref[:, good].max() <= 1e-5 * peak
    # everywhere, in the metric of the FFT's own output
    wpeak = np.abs(truth_ours * t2).max()
    assert (err_hip * t2).max() <= 1e-5 * wpeak
    assert (err_ref * t2).max() <= 1e-5 * wpeak
    # the rest: amplified for every float32 implementation alike
    assert err_hip[:, bad].max() <= 2 * err_ref[:, bad].max() + 1e-5 * peak
    # (for the record: how large that amplified error is; 1e-5 would not hold for either)
    assert err_ref[:, bad].max() > 1e-5 * peak or err_hip[:, bad].max() <= 1e-5 * peak


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(gi.IMAGE_CONFIGS))
def test_grid_image_vs_fp64_truth(golden, name):
    """G5 against a float64 truth: GridToImage (zero-pad + rocFFT + layer_to_image) and the
    reference's float32 host result are both within 1e-5 (unweighted: the G5 kernel1d stays within
    [1, 2]) of the float64 evaluation at w = 0, and within 1e-5 + 4e-7 |w| at w != 0 (the float32
    evaluation of n - 1 that both share)."""
    from katsdpimager_amd import image
    from helpers import grid_to_image_truth
    ctx, q = context_queue()
    c = gi.IMAGE_CONFIGS[name]
    g = golden('g5_image_' + name)
    ii = gi.image_inputs(c)
    shape = ii['image_shape']
    template = image.GridImageTemplate(ctx, np.float32)
    g2i = template.instantiate_grid_to_image(q, shape, c['lm_scale'], c['lm_bias'],
                                             template.make_fft_plan(shape[1:], shape[1:]))
    g2i.ensure_all_bound()
    g2i.buffer('kernel1d').set(q, ii['kernel1d'])
    g2i.buffer('grid').set(q, ii['grid'])
    for wi, w in enumerate(c['ws']):
        g2i.buffer('image').zero(q)
        g2i.set_w(w)
        g2i()
        g2i()
        want = 2 * grid_to_image_truth(ii['grid'], ii['kernel1d'], c['lm_scale'], c['lm_bias'], w)
        peak = np.abs(want).max()
        # the w correction e^{2 pi i w (n - 1)} is evaluated in float32 by the reference's host path
        # (image.py:786-790) and, op for op, by the kernel: n - 1 carries an absolute rounding of
        # ~6e-8, i.e. a phase error of 2 pi |w| 6e-8 that BOTH share (they agree with each other to
        # 1e-5, test_grid_image_vs_golden); against the float64 truth it adds 4e-7 |w|
        tol = 1e-5 + 4e-7 * abs(w)
        err_hip = np.abs(g2i.buffer('image').get(q) - want).max() / peak
        err_ref = np.abs(g['g2i_w%d' % wi] - want).max() / peak
        assert err_hip <= tol and err_ref <= tol
        if w == 0.0:
            assert err_hip <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('arith', ARITHS)
@pytest.mark.parametrize('K,W,P', [(28, 32, 1), (8, 8, 2), (60, 16, 1), (45, 160, 3)])
def test_degridder_binned_variant(K, W, P, arith):
    """The degridder's KIMG_VARIANT_BINNED (device sort by grid tile, gather of coordinates /
    weights / visibilities, window degridder, scatter back into the caller's order) on a stream
    without locality, against the oracle and the direct window kernel; `auto` takes it for such a
    stream and leaves a track alone."""
    from katsdpimager_amd import grid
    ctx, q = context_queue()
    c = gi.make_config(512, 0.0001, 0.01, P, K, W, grid_cover=300, n_vis=70000)
    t = gi.grid_track(c)
    ip, gp, ap = make_params(c)
    rs = gi.RandomState(K + W + 5)
    n = c['n_vis']
    order = np.random.RandomState(9).permutation(n)
    uv4 = np.concatenate((t['uv'], t['sub_uv']), axis=1)
    weights = rs.uniform(0.5, 2.0, (n, P)).astype(np.float32)
    vis0 = rs.complex_uniform(-1, 1, size=(n, P)).astype(np.complex64)

    def run(variant, perm, count=n):
        fn = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed, {'variant': variant, 'arith': arith}) \
            .instantiate(q, ap, ip, gp, n)
        fn.ensure_all_bound()
        fn.buffer('grid').set(q, gdata)
        fn.num_vis = count
        fn.buffer('uv').set_region(q, uv4[perm][:count], np.s_[:count], np.s_[:])
        fn.buffer('w_plane').set_region(q, t['w_plane'][perm][:count], np.s_[:count], np.s_[:])
        fn.buffer('weights').set_region(q, weights[perm][:count], np.s_[:count], np.s_[:])
        fn.buffer('vis').set_region(q, vis0[perm][:count], np.s_[:count], np.s_[:])
        fn()
        return fn.buffer('vis').get(q)[:count], fn

    G = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed).instantiate(q, ap, ip, gp, 16).slots['grid'].shape[-1]
    gdata = rs.complex_uniform(-1, 1, size=(P, G, G)).astype(np.complex64)
    got, fb = run('binned', order)
    assert fb.last_variant == 'binned'
    # the oracle on a sample of the shuffled stream (all of it would take minutes in C at K = 60)
    m = 6000
    expected = vis0[order][:m].copy()
    orc.degrid(fb.convolve_kernel.data, gdata, np.ascontiguousarray(uv4[order][:m, :2]),
               np.ascontiguousarray(uv4[order][:m, 2:]), t['w_plane'][order][:m], weights[order][:m],
               expected)
    scale = np.abs(expected - vis0[order][:m]).max()
    assert np.abs(got[:m] - expected).max() <= 1e-5 * scale + 2.4e-7 * np.abs(expected).max()
    direct, fd = run('mfma', order)
    assert np.abs(got - direct).max() <= 1e-5 * scale + 2.4e-7 * np.abs(direct).max()
    auto, fa = run('auto', order)
    assert fa.last_variant == 'binned' and np.array_equal(auto, got)
    _, ft = run('auto', np.arange(n))
    assert ft.last_variant == 'mfma'
    # fewer visibilities than max_vis, scratch reused
    part, _ = run('binned', order, 12345)
    assert np.abs(part - got[:12345]).max() <= 1e-5 * scale + 2.4e-7 * np.abs(direct).max()
