"""Pin the oracle (oracle/kimg_oracle.{py,c}) to golden vectors produced by the
imported reference host classes (tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest

import golden_inputs as gi
from oracle import kimg_oracle as orc


def relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b))


@pytest.mark.parametrize('name', list(gi.KERNEL_CONFIGS))
def test_g1_kernel(golden, name):
    c = gi.KERNEL_CONFIGS[name]
    g = golden('g1_kernel_' + name)
    data, beta = orc.convolution_kernel(c['cell_size'], c['wavelength'], c['max_w'], c['w_slices'],
                                        c['w_planes'], c['oversample'], c['kernel_width'],
                                        c['antialias_width'], c['image_oversample'])
    assert data.shape == g['data'].shape and data.dtype == np.complex64
    assert beta == g['beta']
    np.testing.assert_array_equal(data, g['data'])       # same numpy ops -> bit-identical
    np.testing.assert_array_equal(
        orc.taper(c['pixels'], c['antialias_width'], beta, c['oversample']), g['taper'])


@pytest.mark.parametrize('name', list(gi.GRID_CONFIGS))
def test_g2_grid(golden, name):
    c = gi.GRID_CONFIGS[name]
    t = gi.grid_track(c)
    kernel, _ = orc.convolution_kernel(c['cell_size'], c['wavelength'], c['max_w'], c['w_slices'],
                                       c['w_planes'], c['oversample'], c['kernel_width'],
                                       c['antialias_width'], c['image_oversample'])
    G = c['pixels']
    for impl in (orc.grid, orc.grid_py):
        grid = np.zeros((c['P'], G, G), c['complex_dtype'])
        wg = np.zeros((c['P'], G, G), np.float32)
        gi.middle(wg, t['weights_grid'].shape)[:] = t['weights_grid']
        impl(kernel, grid, wg, t['uv'], t['sub_uv'], t['w_plane'], t['vis'])
        expected = golden('g2_grid_' + name)['grid']
        tol = 2e-6 if c['real_dtype'] == 'float32' else 1e-7
        assert relerr(grid, expected) < tol


@pytest.mark.parametrize('name', list(gi.GRID_CONFIGS))
def test_g3_degrid(golden, name):
    c = gi.GRID_CONFIGS[name]
    t = gi.grid_track(c)
    dg = gi.degrid_inputs(c)
    kernel, _ = orc.convolution_kernel(c['cell_size'], c['wavelength'], c['max_w'], c['w_slices'],
                                       c['w_planes'], c['oversample'], c['kernel_width'],
                                       c['antialias_width'], c['image_oversample'])
    expected = golden('g3_degrid_' + name)['residual']
    vis = dg['vis'].copy()
    orc.degrid(kernel, dg['grid'], t['uv'], t['sub_uv'], t['w_plane'], dg['weights'], vis)
    np.testing.assert_allclose(vis, expected, rtol=1e-5, atol=2e-6)
    n = 40
    vis = dg['vis'][:n].copy()
    orc.degrid_py(kernel, dg['grid'], t['uv'][:n], t['sub_uv'][:n], t['w_plane'][:n],
                  dg['weights'][:n], vis)
    np.testing.assert_allclose(vis, expected[:n], rtol=1e-5, atol=2e-6)


def test_g4_predict(golden):
    c = gi.PREDICT_CONFIG
    g = golden('g4_predict')
    pi = gi.predict_inputs(c)
    uv_scale, w_scale, w_bias = orc.uvw_scale_bias(c['cell_size'], c['wavelength'], c['max_w'],
                                                   c['w_slices'], c['w_planes'], c['oversample'])
    assert (uv_scale, w_scale, w_bias) == (g['uv_scale'], g['w_scale'], g['w_bias'])
    lmn, flux = orc.extract_sky_image(c['pixels'], c['pixel_size'], c['image_size'],
                                      c['oversample'], pi['components'])
    np.testing.assert_array_equal(lmn, g['lmn'])
    np.testing.assert_array_equal(flux, g['flux'])
    vis = pi['vis'].copy()
    orc.predict(vis, pi['uv'], pi['sub_uv'], pi['w_plane'], pi['weights'], lmn, flux,
                c['oversample'], uv_scale, w_scale, w_bias + pi['w'])
    # The float32 phase carries up to ~1e3 whole turns here (|u| ~ 5e4 wavelengths, l ~ 0.02),
    # so each term is only good to ~4e-4 rad whatever the summation order (test_predict.py:88-92
    # makes the same remark).  Compare the *predicted* part norm-wise, and both against float64.
    pred = pi['vis'] - vis
    pred_ref = pi['vis'] - g['residual']
    assert relerr(pred, pred_ref) < 2e-3
    u = (pi['uv'].astype(np.float64) * c['oversample'] + pi['sub_uv'] + 0.5) * uv_scale
    w = pi['w_plane'] * w_scale + w_bias + pi['w']
    phase = (u[:, 0:1] * lmn[:, 0].astype(np.float64) + u[:, 1:2] * lmn[:, 1].astype(np.float64)
             + w[:, None] * lmn[:, 2].astype(np.float64))
    exact = (np.exp(-2j * np.pi * phase) @ flux.astype(np.float64)) * pi['weights']
    assert relerr(pred, exact) < 2e-3
    assert relerr(pred_ref, exact) < 2e-3


def test_g4_predict_mild(golden):
    """The reference test's own geometry (test_predict.py:45-92: sources within 2.5 arcmin of the
    phase centre): element-wise at its own tolerance, rtol 5e-4."""
    c = gi.PREDICT_CONFIG
    g = golden('g4_predict_mild')
    pi = gi.predict_inputs(c)
    uv_scale, w_scale, w_bias = orc.uvw_scale_bias(c['cell_size'], c['wavelength'], c['max_w'],
                                                   c['w_slices'], c['w_planes'], c['oversample'])
    lmn, flux = orc.extract_sky_image(c['pixels'], c['pixel_size'], c['image_size'],
                                      c['oversample'], gi.predict_mild_components())
    np.testing.assert_array_equal(lmn, g['lmn'])
    np.testing.assert_array_equal(flux, g['flux'])
    vis = pi['vis'].copy()
    orc.predict(vis, pi['uv'], pi['sub_uv'], pi['w_plane'], pi['weights'], lmn, flux,
                c['oversample'], uv_scale, w_scale, w_bias + pi['w'])
    np.testing.assert_allclose(vis, g['residual'], rtol=5e-4)


@pytest.mark.parametrize('name', list(gi.IMAGE_CONFIGS))
def test_g5_image(golden, name):
    c = gi.IMAGE_CONFIGS[name]
    g = golden('g5_image_' + name)
    ii = gi.image_inputs(c)
    for wi, w in enumerate(c['ws']):
        img = np.zeros(ii['image_shape'], c['real_dtype'])
        orc.grid_to_image(ii['grid'], img, ii['kernel1d'], c['lm_scale'], c['lm_bias'], w)
        orc.grid_to_image(ii['grid'], img, ii['kernel1d'], c['lm_scale'], c['lm_bias'], w)
        assert relerr(img, g['g2i_w%d' % wi]) < 1e-5
        grid, _ = orc.image_to_grid(ii['model'], ii['kernel1d'], c['lm_scale'], c['lm_bias'], w)
        assert relerr(grid, g['i2g_w%d' % wi]) < 1e-5


def test_g6_weights(golden):
    g = golden('g6_weights')
    wi = gi.weights_inputs()
    for name, wt in [('natural', orc.NATURAL), ('uniform', orc.UNIFORM), ('robust', orc.ROBUST)]:
        wg = np.zeros(wi['shape'], np.float32)
        if wt != orc.NATURAL:
            orc.weights_grid_add(wg, wi['uv'], wi['weights'])
        rms, nrms = orc.weights_finalize(wt, wg, wi['robustness'])
        np.testing.assert_array_equal(wg, g[name + '_grid'])
        assert nrms == g[name + '_nrms']
        if rms is None:
            assert np.isnan(g[name + '_rms'])
        else:
            assert rms == g[name + '_rms']


@pytest.mark.parametrize('name', list(gi.CLEAN_CONFIGS))
def test_g7_clean(golden, name):
    c = gi.CLEAN_CONFIGS[name]
    g = golden('g7_clean_' + name)
    ci = gi.clean_inputs(c)
    dirty = ci['dirty'].copy()
    model = np.zeros_like(dirty)
    ch = orc.Clean(c['pixels'], c['border'], c['loop_gain'], c['mode'], dirty, ci['psf'], model)
    ch.reset()
    np.testing.assert_array_equal(ch._tile_max, g['tile_max0'])
    np.testing.assert_array_equal(ch._tile_pos, g['tile_pos0'])
    values, pos, pix, true_pos = [], [], [], []
    for i in range(c['cycles']):
        v, p, m = ch(ci['psf_patch'], c['threshold'])
        if v is None:
            break
        values.append(v)
        pos.append(p)
        pix.append(m)
        true_pos.append(ch.last_pos)
    # bit-exact: positions (returned-with-aliasing and actually subtracted), metric values,
    # component fluxes, residual image
    np.testing.assert_array_equal(np.array(pos, np.int32), g['pos'])
    np.testing.assert_array_equal(np.array(true_pos, np.int32), g['true_pos'])
    np.testing.assert_array_equal(np.array(values, np.float32), g['values'])
    np.testing.assert_array_equal(np.array(pix, np.float32), g['pixels'])
    np.testing.assert_array_equal(dirty, g['dirty_final'])
    np.testing.assert_array_equal(model, g['model_final'])
    if name == 'threshold':
        assert len(values) < c['cycles']


def test_g8_psf_patch_noise(golden):
    g = golden('g8_psfpatch_noise')
    for i, (psf, thr, lim) in enumerate(gi.psf_patch_cases()):
        assert tuple(g['patch%d' % i]) == orc.psf_patch(psf, thr, lim)
    for i, (img, border) in enumerate(gi.noise_cases()):
        assert orc.noise_est(img, border) == g['noise%d' % i]


@pytest.mark.parametrize('case', range(3))
def test_g10_convolve_beam(golden, case):
    """beam.convolve_beam (beam.py:172-201) vs the restatement."""
    name, model, b = gi.beam_cases()[case]
    g = golden('g10_beam_' + name)
    np.testing.assert_allclose(orc.beam_covariance_sqrt(b['x_stddev'], b['y_stddev'], b['theta']),
                               g['cov_sqrt'], rtol=1e-13)
    out = orc.convolve_beam(model, **b)
    assert out.dtype == np.float32
    np.testing.assert_allclose(out, g['restored'], rtol=0, atol=1e-6 * np.abs(g['restored']).max())
