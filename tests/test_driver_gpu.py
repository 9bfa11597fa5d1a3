"""What frontend.process_channel uses beyond the reference's calls, against those calls: the PSF stage
without the host in between (Imaging.scale_dirty_by_centre / scale_dirty_by_kept / psf_patch_start /
psf_patch_finish against scale_dirty(1 / central pixel) and psf_patch(), frontend.py:541-548), bit for
bit.  (The driver's whole result with and without the short cuts: test_preprocess_gpu.py runs
process_channel both ways and against the G9 goldens; the one-call major cycles against the reference's
two steps: test_clean_multi_gpu.py.)"""
import numpy as np
import pytest

from helpers import context_queue, make_params
import golden_inputs as gi

pytestmark = pytest.mark.gpu


def _imager(c):
    from katsdpimager_amd import imaging, parameters, weight
    ctx, q = context_queue()
    ip, gp, ap = make_params(c)
    wp = parameters.WeightParameters(weight.WeightType(c['weight_type']), c['robustness'])
    cp = parameters.CleanParameters(c['minor'], c['loop_gain'], c['major_gain'], c['threshold'],
                                    c['mode'], c['psf_cutoff'], c['psf_limit'], c['border'])
    im = imaging.ImagingTemplate(ctx, ap, ip.fixed, wp, gp.fixed, cp).instantiate(
        q, ip, gp, c['vis_block'], 0, c['major'])
    im.ensure_all_bound()
    return im, q


@pytest.mark.parametrize('name', list(gi.E2E_CONFIGS)[:2])
def test_psf_stage_on_the_device(name):
    c = gi.E2E_CONFIGS[name]
    im, q = _imager(c)
    P, H, W = im.buffer('dirty').shape
    rs = np.random.RandomState(3)
    g1 = np.exp(-0.5 * ((np.arange(H) - H // 2) / 2.5) ** 2)
    psf = (np.outer(g1, g1)[None] * rs.uniform(0.3, 3.0, (P, 1, 1))
           + 0.004 * rs.standard_normal((P, H, W))).astype(np.float32)
    other = rs.standard_normal((P, H, W)).astype(np.float32)
    # the reference's calls
    im.set_buffer('dirty', psf)
    peak = psf[:, H // 2, H // 2].copy()
    scale = np.reciprocal(peak)
    im.scale_dirty(scale)
    want_psf = im.get_buffer('dirty')
    im.dirty_to_psf()
    want_patch = im.psf_patch()
    im.set_buffer('dirty', other)
    im.scale_dirty(scale)
    want_other = im.get_buffer('dirty')
    im.dirty_to_psf()               # (back: `dirty` is the buffer the PSF went through)
    # the same on the device
    im.set_buffer('dirty', psf)
    im.scale_dirty_by_centre()
    np.testing.assert_array_equal(im.get_buffer('dirty'), want_psf)
    im.dirty_to_psf()
    started = im.psf_patch_start()
    im.set_buffer('dirty', other)
    im.scale_dirty_by_kept()
    np.testing.assert_array_equal(im.get_buffer('dirty'), want_other)
    patch, got_scale = im.psf_patch_finish(started)
    assert tuple(patch) == tuple(want_patch)
    np.testing.assert_array_equal(got_scale, scale)
    im.dirty_to_psf()
    # a channel without data: the central pixel is 0, its reciprocal infinite
    empty = psf.copy()
    empty[0, H // 2, H // 2] = 0.0
    im.set_buffer('dirty', empty)
    im.scale_dirty_by_centre()
    im.dirty_to_psf()
    patch, got_scale = im.psf_patch_finish(im.psf_patch_start())
    assert np.isinf(got_scale[0])
