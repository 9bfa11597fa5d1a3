"""Known-answer vectors held by the reference's own unit tests, restated
against the oracle.  CPU only.  Citations: katsdpimager/test/*.py."""
import math

import numpy as np
import pytest

import golden_inputs as gi
from oracle import kimg_oracle as orc


def _kernel(c):
    return orc.convolution_kernel(c['cell_size'], c['wavelength'], c['max_w'], c['w_slices'],
                                  c['w_planes'], c['oversample'], c['kernel_width'],
                                  c['antialias_width'], c['image_oversample'])[0]


def test_grid_bruteforce_f64():
    """test_grid.py:91-112 (do_grid): float64 brute force, rtol 1e-5 / atol 1e-8, 256^2."""
    c = gi.make_config(256, 0.0001, 0.01, 4, 28, 32, real_dtype='float64',
                       grid_cover=180, n_vis=1000)
    t = gi.grid_track(c)
    kernel = _kernel(c)
    G = c['pixels']
    actual = np.zeros((4, G, G), np.complex128)
    wg = np.zeros((4, G, G), np.float32)
    gi.middle(wg, t['weights_grid'].shape)[:] = t['weights_grid']
    orc.grid(kernel, actual, wg, t['uv'], t['sub_uv'], t['w_plane'], t['vis'])
    expected = np.zeros_like(actual)
    uv_bias = (kernel.shape[-1] - 1) // 2 - G // 2
    for i in range(c['n_vis']):
        k = np.outer(kernel[t['w_plane'][i], t['sub_uv'][i, 1], :],
                     kernel[t['w_plane'][i], t['sub_uv'][i, 0], :])
        k = np.conj(k)
        u = t['uv'][i, 0] - uv_bias
        v = t['uv'][i, 1] - uv_bias
        wu = t['uv'][i, 0] + t['weights_grid'].shape[2] // 2
        wv = t['uv'][i, 1] + t['weights_grid'].shape[1] // 2
        for j in range(4):
            expected[j, v:v + 28, u:u + 28] += (t['vis'][i, j].astype(np.complex128)
                                                * t['weights_grid'][j, wv, wu] * k)
    np.testing.assert_allclose(expected, actual, 1e-5, 1e-8)


def test_degrid_bruteforce_f64():
    """test_grid.py:114-135 (do_degrid), rtol 1e-5."""
    c = gi.make_config(256, 0.0001, 0.01, 4, 28, 32, real_dtype='float64',
                       grid_cover=180, n_vis=1000)
    t = gi.grid_track(c)
    dg = gi.degrid_inputs(c)
    kernel = _kernel(c)
    G = c['pixels']
    uv_bias = (kernel.shape[-1] - 1) // 2 - G // 2
    expected = np.zeros_like(dg['vis'])
    for i in range(c['n_vis']):
        k = np.outer(kernel[t['w_plane'][i], t['sub_uv'][i, 1], :],
                     kernel[t['w_plane'][i], t['sub_uv'][i, 0], :])
        u = t['uv'][i, 0] - uv_bias
        v = t['uv'][i, 1] - uv_bias
        for j in range(4):
            fp = dg['grid'][j, v:v + 28, u:u + 28]
            expected[i, j] = dg['vis'][i, j] - dg['weights'][i, j] * np.dot(k.ravel(), fp.ravel())
    vis = dg['vis'].copy()
    orc.degrid(kernel, dg['grid'], t['uv'], t['sub_uv'], t['w_plane'], dg['weights'], vis)
    np.testing.assert_allclose(expected, vis, 1e-5)


def test_grid_weights_known():
    """test_weight.py:10-57: exact equality."""
    shape = (4, 100, 200)
    uv = np.array([[-10, 5], [23, 17], [-10, 5], [-10, 5], [-10, 6], [-11, 5]], np.int16)
    w = np.array([[1.0, 10.0, 100.0, 1000.0], [2.0, 20.0, 200.0, 2000.0],
                  [4.0, 40.0, 400.0, 4000.0], [8.0, 80.0, 800.0, 8000.0],
                  [16.0, 160.0, 1600.0, 16000.0], [32.0, 320.0, 3200.0, 32000.0]], np.float32)
    grid = np.zeros(shape, np.float32)
    orc.weights_grid_add(grid, uv, w)
    expected = np.zeros(shape, np.float32)
    for i in range(4):
        expected[i, 55, 90] = 13 * 10 ** i
        expected[i, 67, 123] = 2 * 10 ** i
        expected[i, 56, 90] = 16 * 10 ** i
        expected[i, 55, 89] = 32 * 10 ** i
    np.testing.assert_equal(expected, grid)


def test_robust_weights_vs_formula():
    """Formulae of weight.py:12-26 / test_weight.py:60-118 against weights_finalize."""
    rs = np.random.RandomState(1)
    shape = (4, 50, 107)
    data = np.zeros(shape, np.float32)
    idx = rs.choice(data.size, 100, replace=False)
    data.flat[idx] = rs.uniform(0.1, 2.0, 100)
    w0 = data[0].astype(np.float64)
    mean_weight = np.sum(w0 * w0) / np.sum(w0)
    R = 0.3
    S2 = (5 * 10 ** (-R)) ** 2 / mean_weight
    d = np.where(data != 0, 1.0 / (S2 * data.astype(np.float64) + 1.0), 0.0)
    grid = data.copy()
    rms, nrms = orc.weights_finalize(orc.ROBUST, grid, R)
    np.testing.assert_allclose(grid, d, rtol=1e-5, atol=1e-5)
    sum_w, sum_dw, sum_d2w = np.sum(w0), np.sum(d[0] * w0), np.sum(d[0] ** 2 * w0)
    np.testing.assert_allclose(rms, np.sqrt(sum_d2w) / sum_dw, rtol=1e-5)
    np.testing.assert_allclose(nrms, np.sqrt(sum_d2w * sum_w) / sum_dw, rtol=1e-5)


def test_layer_to_image_first_principles():
    """test_image.py:13-45 applied to the whole grid->image step: the inverse FFT
    of the grid replaces the random layer."""
    c = gi.IMAGE_CONFIGS['offcentre']
    ii = gi.image_inputs(c)
    size, w = c['size'], 12.3
    img = np.zeros(ii['image_shape'], np.float32)
    layer = orc.grid_to_image(ii['grid'], img, ii['kernel1d'], c['lm_scale'], c['lm_bias'], w)
    src = np.fft.ifft2(np.fft.ifftshift(ii['grid'].astype(np.complex128), axes=(1, 2)),
                       axes=(1, 2)) * size * size
    lm = np.arange(size) * c['lm_scale'] + c['lm_bias']
    lm2 = lm * lm
    n = np.sqrt(1 - lm2[np.newaxis, :, np.newaxis] - lm2[np.newaxis, np.newaxis, :])
    corrected = np.fft.fftshift(src, axes=(1, 2)) * np.exp(2j * math.pi * w * (n - 1))
    expected = corrected.real * n / np.outer(ii['kernel1d'], ii['kernel1d'])[np.newaxis, ...]
    assert np.max(np.abs(img - expected)) / np.max(np.abs(expected)) < 1e-5
    assert layer.shape == ii['grid'].shape


def test_psf_patch_known():
    """test_clean.py:13-37."""
    cases = gi.psf_patch_cases()
    assert orc.psf_patch(*cases[0]) == (4, 1, 1)
    assert orc.psf_patch(*cases[1]) == (4, 206, 304)
    assert orc.psf_patch(*cases[2]) == (4, 205, 303)
    box = orc.psf_patch(*cases[3])
    target = cases[3][0][1, 0, :152]
    hw = box[2] // 2
    assert sum(target[:-hw] >= 50.5) == 0 and target[-hw] >= 50.5
    assert orc.psf_patch(*cases[4]) == (4, 15, 5)


def test_noise_known():
    """test_clean.py:174-200: robust estimate of std 3.2 (rtol 1e-2) and exactly 0."""
    (img, border), (img0, border0) = gi.noise_cases()
    np.testing.assert_allclose(orc.noise_est(img, border), 3.2, rtol=2e-2)
    assert orc.noise_est(img0, border0) == 0.0


def test_extract_sky_image_known():
    """test_predict.py:107-133."""
    comps = {(0, 4095): np.array([4.0, 0.0, 0.0]), (1024, 512): np.array([2.5, 1.5, 0.0]),
             (2048, 2048): np.array([1.0, 2.0, 3.0]), (4095, 0): np.array([5.0, 1.0, 2.0])}
    lmn, flux = orc.extract_sky_image(4096, 0.00001, 0.00001 * 4096, 8, comps, np.float64)
    np.testing.assert_allclose(lmn[:, 0:2], [[2047e-5, -2048e-5], [-1536e-5, -1024e-5], [0, 0],
                                             [-2048e-5, 2047e-5]], rtol=1e-6, atol=1e-12)
    ef = np.array([[4.0, 0, 0], [2.5, 1.5, 0.0], [1, 2, 3], [5, 1, 2]])
    ef[0] *= np.sinc(0.5 / 8) * np.sinc(2047 / 4096 / 8)
    ef[1] *= np.sinc(0.25 / 8) * np.sinc(0.375 / 8)
    ef[3] *= np.sinc(2047 / 4096 / 8) * np.sinc(0.5 / 8)
    np.testing.assert_allclose(flux, ef)


def test_preprocess_known():
    """test_preprocess.py:76-136: hand-computed quantised + compressed records
    (identity Mueller matrix), both channels."""
    uvw = np.array([[12.1, 2.3, 4.7], [12.102, 2.299, 4.6], [-5.2, -10.6, 7.2], [-1.0, 2.0, 3.0]],
                   np.float32)
    weights = np.array([[[1.3, 0.6, 1.2, 0.1], [1.1, 1.2, 1.3, 1.4], [0.5, 0.6, 0.7, 0.8],
                         [1.0, 0.0, 1.0, 1.0]],
                        [[0.2, 2.4, 1.2, 2.6], [2.8, 2.6, 2.4, 2.2], [1.6, 1.4, 1.2, 1.0],
                         [2.0, 2.0, 0.0, 2.0]]], np.float32)
    vis = np.array([[[0.5 - 2.3j, 0.1 + 4.2j, 0.0 - 3j, 1.5 + 0j],
                     [1.2 + 3.4j, 5.6 + 7.8j, 9.0 + 1.2j, 3.4 + 5.6j],
                     [1.5 + 1.3j, 1.1 + 2.7j, 1.0 - 2j, 2.5 + 1j], [10.0, 10.0, 10.0, 10.0]],
                    [[3.0 + 0j, 0.0 - 6j, 0.2 + 8.4j, 1.0 - 4.6j],
                     [6.8 + 11.2j, 18.0 + 2.4j, 11.2 + 15.6j, 2.4 + 6.8j],
                     [3.0 + 2j, 2.0 - 4j, 2.2 + 5.4j, 3.0 + 2.6j], [20.0, 20.0, 20.0, 20.0]]],
                   np.complex64)
    expected = [
        dict(uv=[[96, 18], [-42, -85]], sub_uv=[[6, 3], [3, 1]], w_plane=[64, 65],
             weights=[[2.4, 1.8, 2.5, 1.5], [0.5, 0.6, 0.7, 0.8]],
             vis=[[1.97 + 0.75j, 6.78 + 11.88j, 11.7 - 2.04j, 4.91 + 7.84j],
                  [0.75 + 0.65j, 0.66 + 1.62j, 0.7 - 1.4j, 2.0 + 0.8j]]),
        dict(uv=[[387, 73], [387, 73], [-167, -340]], sub_uv=[[1, 4], [2, 4], [4, 6]],
             w_plane=[64, 64, 65],
             weights=[[0.2, 2.4, 1.2, 2.6], [2.8, 2.6, 2.4, 2.2], [1.6, 1.4, 1.2, 1.0]],
             vis=[[0.6 + 0.0j, 0.0 - 14.4j, 0.24 + 10.08j, 2.6 - 11.96j],
                  [19.04 + 31.36j, 46.8 + 6.24j, 26.88 + 37.44j, 5.28 + 14.96j],
                  [4.8 + 3.2j, 2.8 - 5.6j, 2.64 + 6.48j, 3.0 + 2.6j]])]
    for ch, wavelength in enumerate([0.25, 0.125]):
        pixel_size = 1.0 / (4096.0 * wavelength)
        cell_size = wavelength / (pixel_size * 2048)
        rec = orc.quantise_uvw(uvw, vis[ch], weights[ch], cell_size, 400.0, 1, 128, 8)
        rec = orc.compress(rec)
        e = expected[ch]
        np.testing.assert_array_equal(rec['uv'], e['uv'])
        np.testing.assert_array_equal(rec['sub_uv'], e['sub_uv'])
        np.testing.assert_array_equal(rec['w_plane'], e['w_plane'])
        np.testing.assert_allclose(rec['weights'], e['weights'], rtol=1e-6)
        np.testing.assert_allclose(rec['vis'], np.array(e['vis']), rtol=1e-5)


def _known_preprocess_inputs():
    uvw = np.array([[12.1, 2.3, 4.7], [12.102, 2.299, 4.6], [-5.2, -10.6, 7.2], [-1.0, 2.0, 3.0]],
                   np.float32)
    weights = np.array([[[1.3, 0.6, 1.2, 0.1], [1.1, 1.2, 1.3, 1.4], [0.5, 0.6, 0.7, 0.8],
                         [1.0, 0.0, 1.0, 1.0]],
                        [[0.2, 2.4, 1.2, 2.6], [2.8, 2.6, 2.4, 2.2], [1.6, 1.4, 1.2, 1.0],
                         [2.0, 2.0, 0.0, 2.0]]], np.float32)
    vis = np.array([[[0.5 - 2.3j, 0.1 + 4.2j, 0.0 - 3j, 1.5 + 0j],
                     [1.2 + 3.4j, 5.6 + 7.8j, 9.0 + 1.2j, 3.4 + 5.6j],
                     [1.5 + 1.3j, 1.1 + 2.7j, 1.0 - 2j, 2.5 + 1j], [10.0, 10.0, 10.0, 10.0]],
                    [[3.0 + 0j, 0.0 - 6j, 0.2 + 8.4j, 1.0 - 4.6j],
                     [6.8 + 11.2j, 18.0 + 2.4j, 11.2 + 15.6j, 2.4 + 6.8j],
                     [3.0 + 2j, 2.0 - 4j, 2.2 + 5.4j, 3.0 + 2.6j], [20.0, 20.0, 20.0, 20.0]]],
                   np.complex64)
    expected = [
        dict(uv=[[96, 18], [-42, -85]], sub_uv=[[6, 3], [3, 1]], w_plane=[64, 65],
             weights=[[2.4, 1.8, 2.5, 1.5], [0.5, 0.6, 0.7, 0.8]],
             vis=[[1.97 + 0.75j, 6.78 + 11.88j, 11.7 - 2.04j, 4.91 + 7.84j],
                  [0.75 + 0.65j, 0.66 + 1.62j, 0.7 - 1.4j, 2.0 + 0.8j]]),
        dict(uv=[[387, 73], [387, 73], [-167, -340]], sub_uv=[[1, 4], [2, 4], [4, 6]],
             w_plane=[64, 64, 65],
             weights=[[0.2, 2.4, 1.2, 2.6], [2.8, 2.6, 2.4, 2.2], [1.6, 1.4, 1.2, 1.0]],
             vis=[[0.6 + 0.0j, 0.0 - 14.4j, 0.24 + 10.08j, 2.6 - 11.96j],
                  [19.04 + 31.36j, 46.8 + 6.24j, 26.88 + 37.44j, 5.28 + 14.96j],
                  [4.8 + 3.2j, 2.8 - 5.6j, 2.64 + 6.48j, 3.0 + 2.6j]])]
    configs = []
    for wavelength in [0.25, 0.125]:
        pixel_size = 1.0 / (4096.0 * wavelength)
        configs.append(dict(max_w=400.0, w_slices=1, w_planes=128, oversample=8,
                            cell_size=wavelength / (pixel_size * 2048)))
    return uvw, weights, vis, configs, expected


@pytest.mark.parametrize('use_feed_angles', [False, True])
def test_preprocess_collector_known(use_feed_angles):
    """BaseTestVisibilityCollector._test_impl (test_preprocess.py:76-136) on the collector
    restatement, with and without (trivial) feed angles, buffer of 64."""
    uvw, weights, vis, configs, expected = _known_preprocess_inputs()
    ident = np.identity(4, np.complex64)
    fa = np.zeros(4, np.float32) if use_feed_angles else None
    coll = orc.VisibilityCollector(configs, 4, 64)
    coll.add(uvw, weights, vis, fa, fa, ident, ident if use_feed_angles else None)
    assert coll.num_input == 8 and coll.num_output == 5
    for ch, e in enumerate(expected):
        rec = coll.slice_arrays(ch, 0)
        np.testing.assert_array_equal(rec['uv'], e['uv'])
        np.testing.assert_array_equal(rec['sub_uv'], e['sub_uv'])
        np.testing.assert_array_equal(rec['w_plane'], e['w_plane'])
        np.testing.assert_allclose(rec['weights'], e['weights'], rtol=1e-6)
        np.testing.assert_allclose(rec['vis'], np.array(e['vis']), rtol=1e-5)


def test_preprocess_collector_empty():
    """test_preprocess.py:71-74."""
    _, _, _, configs, _ = _known_preprocess_inputs()
    coll = orc.VisibilityCollector(configs, 4, 2)
    for ch in range(2):
        assert len(coll.slice_arrays(ch, 0)['uv']) == 0


def test_preprocess_c_vs_numpy():
    """The C collector restatement against the numpy restatement (identity Mueller), bit for bit,
    on clustered coordinates with flags, NaNs, negative w and several w-slices."""
    rng = np.random.default_rng(11)
    n, P = 4000, 2
    base = rng.uniform(-300, 300, (n // 8, 3)).astype(np.float32)
    uvw = (np.repeat(base, 8, axis=0) + rng.normal(0, 0.02, (n, 3))).astype(np.float32)
    weights = rng.uniform(0.5, 2, (1, n, P)).astype(np.float32)
    weights[0, rng.random(n) < 0.1, 0] = 0
    weights[0, rng.random(n) < 0.05, 1] = 0
    vis = (rng.normal(size=(1, n, P)) + 1j * rng.normal(size=(1, n, P))).astype(np.complex64)
    vis[0, rng.random(n) < 0.02, 1] = np.nan
    vis[0, rng.random(n) < 0.02, 0] = np.nan
    conf = dict(max_w=320.0, w_slices=5, w_planes=16, oversample=8, cell_size=1.7)
    coll = orc.VisibilityCollector([conf], P, n)
    coll.add(uvw, weights, vis, None, None, np.identity(P, np.complex64), None)
    rec = orc.quantise_uvw(uvw, vis[0], weights[0], conf['cell_size'], conf['max_w'], 5, 16, 8)
    keep = rec['weights'][:, 0] != 0          # compress() drops records whose first weight is 0
    rec = orc.compress({k: v[keep] for k, v in rec.items()})
    total = 0
    for s in range(5):
        got = coll.slice_arrays(0, s)
        sel = rec['w_slice'] == s
        total += sel.sum()
        for k in ('uv', 'sub_uv', 'w_plane'):
            np.testing.assert_array_equal(got[k], rec[k][sel])
        np.testing.assert_array_equal(got['weights'], rec['weights'][sel])
        np.testing.assert_array_equal(got['vis'].view(np.float32), rec['vis'][sel].view(np.float32))
    assert total == coll.num_output and 0 < total < n


def test_preprocess_mueller_float64():
    """Full Mueller path (parallactic rotation, P=4 from Q=4 and P=1 from Q=2) against a direct
    float64 evaluation of the formulas in preprocess.cpp:244-258,456-471."""
    rng = np.random.default_rng(12)
    n = 500
    uvw = rng.uniform(-100, 100, (n, 3)).astype(np.float32)
    conf = dict(max_w=120.0, w_slices=2, w_planes=8, oversample=8, cell_size=2.0)
    for P, Q in [(4, 4), (1, 2), (2, 4)]:
        weights = rng.uniform(0.5, 2, (n, Q)).astype(np.float32)
        vis = (rng.normal(size=(n, Q)) + 1j * rng.normal(size=(n, Q))).astype(np.complex64)
        stokes = (rng.normal(size=(P, 4)) + 1j * rng.normal(size=(P, 4))).astype(np.complex64)
        circ = (rng.normal(size=(4, Q)) + 1j * rng.normal(size=(4, Q))).astype(np.complex64)
        fa1 = rng.uniform(-3, 3, n).astype(np.float32)
        fa2 = rng.uniform(-3, 3, n).astype(np.float32)
        key, w, v = orc.preprocess_convert(uvw, weights, vis, fa1, fa2, stokes, circ, conf, P)
        r1 = np.exp(1j * fa1.astype(np.float64))
        r2 = np.exp(1j * fa2.astype(np.float64))
        rr, rl = r1 * np.conj(r2), r1 * r2
        scale = np.stack([rr, rl, np.conj(rl), np.conj(rr)], axis=1)          # n x 4
        M = np.einsum('pk,nk,kq->npq', stokes.astype(np.complex128), scale, circ.astype(np.complex128))
        xvis = np.einsum('npq,nq->np', M, vis.astype(np.complex128))
        xw = 1.0 / np.einsum('npq,nq->np', np.abs(M) ** 2, 1.0 / weights.astype(np.float64))
        xvis = np.where(uvw[:, 2:3] < 0, np.conj(xvis), xvis) * xw
        np.testing.assert_allclose(w, xw, rtol=2e-5)
        assert np.max(np.abs(v - xvis)) <= 2e-5 * np.max(np.abs(xvis))
    # simple generator: the matrix is used as is
    stokes = (rng.normal(size=(2, 3)) + 1j * rng.normal(size=(2, 3))).astype(np.complex64)
    weights = rng.uniform(0.5, 2, (n, 3)).astype(np.float32)
    vis = (rng.normal(size=(n, 3)) + 1j * rng.normal(size=(n, 3))).astype(np.complex64)
    key, w, v = orc.preprocess_convert(uvw, weights, vis, None, None, stokes, None, conf, 2)
    xvis = vis.astype(np.complex128) @ stokes.astype(np.complex128).T
    xw = 1.0 / ((1.0 / weights.astype(np.float64)) @ (np.abs(stokes.astype(np.complex128)) ** 2).T)
    xvis = np.where(uvw[:, 2:3] < 0, np.conj(xvis), xvis) * xw
    assert np.max(np.abs(v - xvis)) <= 2e-5 * np.max(np.abs(xvis))
    np.testing.assert_allclose(w, xw, rtol=2e-5)


def test_preprocess_buffer_boundary():
    """Merging never crosses a buffer boundary and each buffer emits its slices in order
    (preprocess.cpp:431-509): two equal records split over two buffers stay separate."""
    uvw = np.array([[10.0, 10.0, 1.0]] * 4, np.float32)
    weights = np.ones((1, 4, 1), np.float32)
    vis = np.ones((1, 4, 1), np.complex64)
    conf = dict(max_w=100.0, w_slices=1, w_planes=4, oversample=8, cell_size=1.0)
    for cap, lens in [(4, [4.0]), (2, [2.0, 2.0]), (3, [3.0, 1.0])]:
        coll = orc.VisibilityCollector([conf], 1, cap)
        coll.add(uvw, weights, vis, None, None, np.ones((1, 1), np.complex64), None)
        np.testing.assert_array_equal(coll.slice_arrays(0, 0)['weights'][:, 0], lens)


def test_convolve_beam_sampled_reference():
    """test_beam.py:13-46: the analytic-transform convolution equals a direct convolution with
    the sampled Gaussian2D (rtol 1e-5, atol 1e-5) on the reference's own test case."""
    import scipy.signal
    name, model, b = gi.beam_cases()[0]
    hh, hw = model.shape[1] // 2, model.shape[2] // 2
    x, y = np.meshgrid(np.arange(-hh + 1, hh), np.arange(-hw + 1, hw), indexing='ij')
    sx, sy, th = b['x_stddev'], b['y_stddev'], b['theta']
    # astropy.modeling.models.Gaussian2D.evaluate
    ca = np.cos(th) ** 2 / (2 * sx ** 2) + np.sin(th) ** 2 / (2 * sy ** 2)
    cb = np.sin(2 * th) / (2 * sx ** 2) - np.sin(2 * th) / (2 * sy ** 2)
    cc = np.sin(th) ** 2 / (2 * sx ** 2) + np.cos(th) ** 2 / (2 * sy ** 2)
    beam_pixels = b['amplitude'] * np.exp(-(ca * x * x + cb * x * y + cc * y * y))
    expected = np.stack([scipy.signal.fftconvolve(model[p], beam_pixels, 'same')
                         for p in range(model.shape[0])])
    actual = orc.convolve_beam(model, **b)
    np.testing.assert_allclose(expected, actual, rtol=1e-5, atol=1e-5)


def test_find_peak_known():
    """test_frontend.py:8-27 (reduced from 4096^2 to 512^2)."""
    size, noise, peak = 512, 15.0, 200.0
    x = np.linspace(-np.pi / 2, np.pi / 2, size)
    y = np.cos(x)
    pbeam = y[np.newaxis, :] * y[:, np.newaxis]
    pbeam[pbeam < 0.01] = np.nan
    rs = np.random.RandomState(seed=1)
    with np.errstate(invalid='ignore'):
        image = rs.normal(scale=noise, size=(4, size, size)) / pbeam
    assert np.isnan(orc.find_peak(image, pbeam, noise))
    image[1, size // 2 + 5, size // 2 - 10] = peak
    assert orc.find_peak(image, pbeam, noise) == peak
    image *= -1
    assert orc.find_peak(image, pbeam, noise) == peak
