"""CPU-only tests of the product's host-side logic and of the C ABI surface
(no compute calls: there is no GPU in the build container)."""
import ctypes
import os
import re

import numpy as np
import pytest

import golden_inputs as gi
from helpers import make_params, relerr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from katsdpimager_amd import _lib, build
    build.build_lib()
    header = open(os.path.join(ROOT, 'include', 'kimg.h')).read()
    declared = set(re.findall(r'\b(kimg_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 25
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    assert declared == set(_lib.PROTOTYPES)
    lib = _lib.lib()
    version = int(re.search(r'#define KIMG_VERSION (\d+)', header).group(1))
    assert lib.kimg_version() == version == _lib.VERSION
    assert not re.search(r'\bgetenv\b', ''.join(
        open(os.path.join(ROOT, 'katsdpimager_amd', 'csrc', f)).read()
        for f in os.listdir(os.path.join(ROOT, 'katsdpimager_amd', 'csrc')) if f.endswith(('.hip', '.h')))), \
        'the C ABI takes every choice as an argument; nothing is read from the environment'
    assert _lib.error_string(0) == 'success'
    assert 'unsupported' in _lib.error_string(-10002)


def test_argument_errors_without_gpu():
    """Argument validation happens before any HIP call."""
    from katsdpimager_amd import _lib
    lib = _lib.lib()
    assert lib.kimg_fill(None, 10, 1.0, None) == -10001
    # arithmetic / variant / form selectors are validated like any other argument
    one = ctypes.c_void_p(1)
    assert lib.kimg_grid(one, 8, 64, 8, 1, one, 8, 64, one, one, one, 4, one, 1, 8, 4, None, 0,
                         0, 7, None) == -10001
    assert lib.kimg_grid(one, 8, 64, 8, 1, one, 8, 64, one, one, one, 4, one, 1, 8, 4, None, 0,
                         4, 0, None) == -10001
    assert lib.kimg_degrid(one, 8, 64, 8, 1, one, one, one, one, 4, one, 1, 8, 4, None, 0,
                           0, 2, None) == -10001
    assert lib.kimg_grid_weights(None, 1, 1, 2, 2, 1, None, None, 0, None) == -10001
    # the two-launch grid <-> image routes: which sizes they take, and their argument checks
    assert lib.kimg_grid_image_real_supported(4096, 1244) == 1
    assert lib.kimg_grid_image_real_supported(4800, 1440) == 1      # 2^6 3 5^2
    assert lib.kimg_grid_image_real_supported(6720, 6720) == 1      # 2^6 3 5 7
    assert lib.kimg_grid_image_real_supported(4224, 1000) == 0      # 11
    assert lib.kimg_grid_image_real_supported(16384, 1000) == 0     # LDS
    assert lib.kimg_grid_image_real_supported(4095, 1000) == 0 and \
        lib.kimg_grid_image_real_supported(4096, 1001) == 0         # odd sizes
    assert lib.kimg_grid_image_real_supported(4096, 4098) == 0      # grid larger than the layer
    assert lib.kimg_grid_image_real_workspace_bytes(4096, 1244) == 623 * 4096 * 8
    assert lib.kimg_grid_image_w_workspace_bytes(4096, 1244) == 1244 * 4096 * 8
    assert lib.kimg_grid_image_real_workspace_bytes(4224, 1000) == 0
    sixteen = ctypes.c_void_p(16)
    assert lib.kimg_grid_to_image_real(None, 64, 64, one, 16, 16, one, 0.0, 0.0, 1, sixteen,
                                       1 << 20, None) == -10001
    assert lib.kimg_grid_to_image_real(one, 64, 64, one, 16, 16, one, 0.0, 0.0, 1, sixteen,
                                       100, None) == -10001          # workspace too small
    assert lib.kimg_grid_to_image_real(one, 64, 64, one, 16, 16, one, 0.0, 0.0, 1, one,
                                       1 << 20, None) == -10001          # workspace not aligned
    assert lib.kimg_grid_to_image_real(one, 66, 66, one, 16, 16, one, 0.0, 0.0, 1, sixteen,
                                       1 << 20, None) == -10001          # 66 = 2 3 11
    assert lib.kimg_image_to_grid_w(one, 16, 16, one, 64, 64, None, 0.0, 0.0, 3.0, sixteen,
                                    1 << 20, None) == -10001
    assert lib.kimg_convolve_beam(one, 64, 64, 1.0, 0.0, 0.0, 0.0, sixteen, 100, None) == -10001
    assert lib.kimg_convolve_beam(one, 60, 64, 1.0, 0.0, 0.0, 0.0, sixteen, 1 << 20, None) == -10001
    with pytest.raises(_lib.KimgError):
        _lib.check(-10001, 'kimg_fill')


@pytest.mark.parametrize('name', list(gi.KERNEL_CONFIGS))
def test_kernel_table_matches_reference(golden, name):
    """Host-side kernel generation (grid.ConvolutionKernel) vs the reference's table."""
    from katsdpimager_amd import grid
    c = gi.KERNEL_CONFIGS[name]
    ip, gp, _ = make_params(c)
    g = golden('g1_kernel_' + name)
    ck = grid.ConvolutionKernel(ip, gp)
    assert ck.data.shape == g['data'].shape and ck.data.dtype == np.complex64
    assert ck.beta == g['beta']
    assert relerr(ck.data, g['data']) < 1e-6
    np.testing.assert_allclose(ck.taper(c['pixels']), g['taper'], rtol=1e-12)


def test_parameters_match_reference_formulas():
    from katsdpimager_amd import parameters
    c = gi.KERNEL_CONFIGS['testgrid']
    ip, gp, _ = make_params(c)
    assert ip.image_size == pytest.approx(0.0256)
    assert ip.cell_size == pytest.approx(0.390625)
    with pytest.raises(ValueError):
        parameters.ImageParameters(ip.fixed, 1.0, None, 0.01, None, pixel_size=1e-4, pixels=250)
    assert parameters.is_smooth(4096) and not parameters.is_smooth(4100)
    with pytest.raises(ValueError):
        parameters.CleanParameters(10, 0.1, 0.85, 5.0, 0, 1.5, 0.5, 0.02)
    # automatic sizing (parameters.py:84-110)
    ap = parameters.ArrayParameters(13.5, 8000.0)
    auto = parameters.ImageParameters(ip.fixed, 1.0, 5.0, 0.21, ap)
    assert parameters.is_smooth(auto.pixels) and auto.pixels % 2 == 0
    assert auto.pixel_size == pytest.approx(0.21 / (2.0 / 3.0 * 5.0 * 8000.0))
    slices = parameters.w_slices(auto, 1000.0, 0.001, 60, 7.0)
    assert slices >= 1
    assert parameters.w_kernel_width(auto, 500.0 / (slices - 0.5), 0.001, 7.0) <= 60


def test_parameter_formulas_vs_golden(golden):
    """G12: is_smooth for every size below 20000 and w_slices / w_kernel_width on nine
    configurations against the values of the reference's own functions (parameters.py:17-26,
    :135-183; tools/gen_golden.py)."""
    from types import SimpleNamespace
    from katsdpimager_amd import parameters
    g = golden('g12_parameters')
    smooth = [x for x in range(1, gi.IS_SMOOTH_RANGE) if parameters.is_smooth(x)]
    assert smooth == g['smooth'].tolist()
    assert not parameters.is_smooth(0) and not parameters.is_smooth(-8)
    for k, (ps, px, wl, max_w, eps, kw, aa) in enumerate(gi.W_SLICES_CASES):
        ip = SimpleNamespace(image_size=ps * px, wavelength=wl)
        n = parameters.w_slices(ip, max_w, eps, kw, aa)
        assert n == g['w_slices'][k]
        assert float(parameters.w_kernel_width(ip, 0.5 * max_w / (n - 0.5), eps, aa)) == \
            g['w_kernel_width'][k]
    assert len(set(g['w_slices'].tolist())) >= 4          # the cases do exercise the bisection


def test_extract_sky_image_known():
    """test_predict.py:107-133 against the product's host code."""
    from katsdpimager_amd import predict, parameters
    fixed = parameters.FixedImageParameters([0, 1, 3], np.float64)
    ip = parameters.ImageParameters(fixed, 1.0, None, 0.2, None, pixel_size=0.00001, pixels=4096)
    gp = parameters.GridParameters(
        parameters.FixedGridParameters(7.0, 8, 4, 5.0, 7), w_slices=10, w_planes=100)
    comps = {(0, 4095): np.array([4.0, 0.0, 0.0]), (1024, 512): np.array([2.5, 1.5, 0.0]),
             (2048, 2048): np.array([1.0, 2.0, 3.0]), (4095, 0): np.array([5.0, 1.0, 2.0])}
    lmn, flux = predict.extract_sky_image(ip, gp, comps)
    np.testing.assert_allclose(lmn[:, 0:2], [[2047e-5, -2048e-5], [-1536e-5, -1024e-5], [0, 0],
                                             [-2048e-5, 2047e-5]], rtol=1e-6, atol=1e-12)
    ef = np.array([[4.0, 0, 0], [2.5, 1.5, 0.0], [1, 2, 3], [5, 1, 2]])
    ef[0] *= np.sinc(0.5 / 8) * np.sinc(2047 / 4096 / 8)
    ef[1] *= np.sinc(0.25 / 8) * np.sinc(0.375 / 8)
    ef[3] *= np.sinc(2047 / 4096 / 8) * np.sinc(0.5 / 8)
    np.testing.assert_allclose(flux, ef)
    c = gi.PREDICT_CONFIG
    ip2, gp2, _ = make_params(c)
    from oracle import kimg_oracle as orc
    assert predict.uvw_scale_bias(ip2, gp2) == orc.uvw_scale_bias(
        c['cell_size'], c['wavelength'], c['max_w'], c['w_slices'], c['w_planes'], c['oversample'])


def test_float64_rejected():
    from katsdpimager_amd import types
    with pytest.raises(ValueError):
        types.require_float32(np.float64, 'x')


def test_store_dtype_layout():
    """The record type of VisibilityReaderDevice.iter_slice has the reference's field offsets
    (vis_t<P> without w_slice, preprocess.cpp:39-52 + preprocess.py:42-56)."""
    from katsdpimager_amd import preprocess
    for P in (1, 2, 4):
        dt = preprocess.make_store_dtype(P)
        assert dt.names == ('uv', 'sub_uv', 'w_plane', 'weights', 'vis')
        assert [dt.fields[n][1] for n in dt.names] == [0, 4, 8, 12, 12 + 4 * P]
        assert dt.itemsize == 12 + 12 * P
        assert dt['weights'].shape == (P,) and dt['vis'].base == np.complex64


def test_preprocess_argument_validation():
    """Bad arguments are refused before anything touches the device."""
    from katsdpimager_amd import _lib
    L = _lib.lib()
    m = np.identity(4, np.complex64)
    ok = dict(P=4, Q=4, n=0)
    assert L.kimg_preprocess_convert(4, 4, 0, None, None, None, None, None, m.ctypes.data, None,
                                     400.0, 1, 128, 8, 1.0, None, None, None, None) == 0
    bad = [
        (5, 4, 0, 400.0, 1, 128, 8, 1.0), (4, 0, 0, 400.0, 1, 128, 8, 1.0),
        (4, 4, -1, 400.0, 1, 128, 8, 1.0), (4, 4, 0, 0.0, 1, 128, 8, 1.0),
        (4, 4, 0, 400.0, 0, 128, 8, 1.0), (4, 4, 0, 400.0, 1, 40000, 8, 1.0),
        (4, 4, 0, 400.0, 1, 128, 0, 1.0), (4, 4, 0, 400.0, 1, 128, 8, 0.0)]
    for P, Q, n, max_w, ws, wp, ov, cell in bad:
        assert L.kimg_preprocess_convert(P, Q, n, None, None, None, None, None, m.ctypes.data, None,
                                         max_w, ws, wp, ov, cell, None, None, None, None) == -10001
    # feed angles without mueller_circular
    assert L.kimg_preprocess_convert(4, 4, 0, None, None, None, 8, 8, m.ctypes.data, None,
                                     400.0, 1, 128, 8, 1.0, None, None, None, None) == -10001
    assert L.kimg_preprocess_compress(5, 0, 1, None, None, None, None, None, None, None, 8, 0,
                                      None, 0, None) == -10001
    assert L.kimg_preprocess_compress(1, 10, 1, None, None, None, None, None, None, None, 8, 0,
                                      None, 0, None) == -10001
    assert L.kimg_preprocess_compress(1, 10, 1, None, None, None, None, None, None, None, 8, -4,
                                      None, 0, None) == -10001
    assert L.kimg_real_to_complex(None, None, -1, None) == -10001
    assert L.kimg_real_to_complex(None, None, 0, None) == 0


class _Recorder:
    """Stands in for Imaging: records the calls of the driver (no device needed)."""

    def __init__(self, num_pols=1, peaks=(1.0,), cycles_before_threshold=3):
        self.calls = []
        self.command_queue = None
        self._peaks = list(peaks)
        self._cycles = cycles_before_threshold
        self._dirty = _FakeDirty(num_pols)
        self.num_vis = 0

    def buffer(self, name):
        assert name == 'dirty'
        return self._dirty

    def finalize_weights(self):
        self.calls.append(('finalize_weights',))
        return 0.5, 2.0

    def psf_patch(self):
        self.calls.append(('psf_patch',))
        return (1, 9, 11)

    def noise_est(self):
        self.calls.append(('noise_est',))
        return 0.01

    def clean_cycle(self, psf_patch, threshold=0.0):
        self.calls.append(('clean_cycle', threshold))
        if threshold == 0.0:
            return self._peaks.pop(0)
        self._cycles -= 1
        return 0.5 if self._cycles >= 0 else None

    def clean_cycles(self, psf_patch, threshold, max_cycles):
        self.calls.append(('clean_cycles', max_cycles))
        return [0.5] * min(self._cycles, max_cycles)

    def __getattr__(self, name):
        def record(*args, **kwargs):
            self.calls.append((name,) + tuple(a for a in args if isinstance(a, (str, int, float))))
        return record


class _FakeDirty:
    def __init__(self, P):
        self.shape = (P, 8, 8)
        self.dtype = np.dtype(np.float32)

    def get_region(self, queue, out, region, out_region):
        out[...] = 2.0


class _HostReader:
    def __init__(self, lengths, P=1):
        self.lengths = lengths
        self.dtype = np.dtype([('uv', 'i2', (2,)), ('sub_uv', 'i2', (2,)), ('w_plane', 'i2'),
                               ('weights', 'f4', (P,)), ('vis', 'c8', (P,))])

    def num_w_slices(self, channel):
        return len(self.lengths)

    def len(self, channel, w_slice):
        return self.lengths[w_slice]

    def iter_slice(self, channel, w_slice, block_size=None):
        n = self.lengths[w_slice]
        for start in range(0, n, block_size):
            yield np.rec.recarray((min(block_size, n - start),), self.dtype)


class _DeviceReader(_HostReader):
    def iter_slice_device(self, channel, w_slice, block_size=None):
        from katsdpimager_amd import preprocess
        n = self.lengths[w_slice]
        for start in range(0, n, block_size):
            yield preprocess.DeviceChunk(min(block_size, n - start), None, None, None, None)


def _driver_params(major_gain=0.85, threshold=5.0, minor=10):
    import types
    from katsdpimager_amd import clean
    image_p = types.SimpleNamespace(fixed=types.SimpleNamespace(polarizations=[0]), wavelength=0.2)
    grid_p = types.SimpleNamespace(fixed=types.SimpleNamespace(max_w=100.0), w_slices=3)
    clean_p = types.SimpleNamespace(mode=clean.CLEAN_I, threshold=threshold, major_gain=major_gain,
                                    minor=minor)
    return image_p, grid_p, clean_p


def test_frontend_driver_call_sequence():
    """The driver issues the calls of frontend.make_weights / make_dirty / process_channel
    (frontend.py:86-142, 497-585) in the reference's order, on either kind of reader."""
    from katsdpimager_amd import frontend, weight
    image_p, grid_p, clean_p = _driver_params()
    mid_w = frontend.slice_mid_w(image_p, grid_p)
    np.testing.assert_allclose(mid_w, np.arange(3) * (100.0 / 0.2 / 2.5))

    # host reader, degridding, uniform weights, 2 major cycles; slice 1 is empty
    im = _Recorder(peaks=[1.0, 1.0], cycles_before_threshold=3)
    out = frontend.process_channel(_HostReader([5, 0, 3]), 0, im, image_p, grid_p, clean_p,
                                   weight.WeightType.UNIFORM, 4, 2, True, batched_clean=False)
    names = [c[0] for c in im.calls]
    assert names[:2] == ['clear_model', 'clear_weights']
    assert names.count('grid_weights') == 3                  # chunks of 4: 4+1, (empty), 3
    first_dirty = names.index('clear_dirty')
    assert names[first_dirty - 1] == 'finalize_weights'
    # PSF pass: per non-empty slice clear_grid, chunks (set_coordinates, set_vis, grid), FFT
    psf = names[first_dirty:names.index('scale_dirty')]
    assert psf == ['clear_dirty',
                   'clear_grid', 'set_coordinates', 'set_vis', 'grid',
                   'set_coordinates', 'set_vis', 'grid', 'grid_to_image',
                   'clear_grid', 'set_coordinates', 'set_vis', 'grid', 'grid_to_image']
    assert 'set_weights' not in psf and 'predict' not in psf
    assert names[names.index('scale_dirty'):][:3] == ['scale_dirty', 'dirty_to_psf', 'psf_patch']
    # second major cycle predicts through the model grid before gridding each chunk
    second = names[len(names) - names[::-1].index('clear_dirty') - 1:]
    assert second[:3] == ['clear_dirty', 'model_to_grid', 'clear_grid']
    assert second.count('predict') == 3 and second.count('set_weights') == 3
    assert second.index('predict') < second.index('grid')
    assert 'model_to_predict' not in names
    # minor cycles: 3 successful + the one that hit the threshold, in each of the 2 major cycles?
    # the recorder only allows 3 in total: first cycle 3 + 1 (None), second 0 + 1 (None)
    assert out['major'] == 2 and out['minor'] == 5
    assert out['psf_patch'] == (1, 9, 11) and out['weights_noise'] == 0.5
    np.testing.assert_array_equal(out['scale'], [0.5])
    assert names[-1] == 'noise_est'                          # frontend.py:583-585

    # device reader, DFT predictor, natural weights: zero-copy calls, no grid_weights at all
    im = _Recorder(peaks=[1.0, 1.0], cycles_before_threshold=20)
    out = frontend.process_channel(_DeviceReader([5, 0, 3]), 0, im, image_p, grid_p, clean_p,
                                   weight.WeightType.NATURAL, 4, 2, False, batched_clean=True)
    names = [c[0] for c in im.calls]
    assert 'grid_weights' not in names and 'grid_weights_device' not in names
    assert names.count('set_chunk_device') == 9 and 'set_coordinates' not in names
    assert [c for c in im.calls if c[0] == 'set_chunk_device'][:3] == [('set_chunk_device', 'weights')] * 3
    assert names.count('model_to_predict') == 1 and 'model_to_grid' not in names
    assert [c for c in im.calls if c[0] == 'clean_cycles'] == [('clean_cycles', 9)] * 2
    assert out['minor'] == 18                                # 9 + 9, none hit the threshold

    # stops when the first peak is already below the threshold; nothing to do without data
    im = _Recorder(peaks=[0.01])
    out = frontend.process_channel(_HostReader([5]), 0, im, image_p, grid_p, clean_p,
                                   weight.WeightType.UNIFORM, 4, 3, True)
    assert out['major'] == 1 and out['minor'] == 0
    assert frontend.process_channel(_HostReader([0, 0]), 0, _Recorder(), image_p, grid_p, clean_p,
                                    weight.WeightType.UNIFORM, 4, 3, True) is None


def test_fft_plan_pool_hands_out_each_plan_once():
    """Idle plans are reused by shape, never shared, and the pool is bounded."""
    from katsdpimager_amd import image
    pool = image._PlanPool()
    assert pool.take((8, 8)) is None
    for h in range(pool.MAX_IDLE):
        assert pool.give((8, 8), h)
    assert not pool.give((8, 8), 99)          # full: the caller destroys it
    assert pool.take((4, 4)) is None
    got = {pool.take((8, 8)) for _ in range(pool.MAX_IDLE)}
    assert got == set(range(pool.MAX_IDLE))
    assert pool.take((8, 8)) is None


def test_polarization_matrices_vs_golden(golden):
    """G11: every (outputs, inputs) combination of golden_inputs.polarization_cases, including
    the unsolvable ones, against the reference's polarization_matrix (bit-identical)."""
    import golden_inputs as gi
    from katsdpimager_amd import polarization
    g = golden('g11_polarization')
    np.testing.assert_array_equal(polarization.STOKES_COEFF, g['coeff'])
    solvable = 0
    for idx, (outputs, inputs) in enumerate(gi.polarization_cases()):
        expected = g['m%d' % idx]
        if expected.size == 0:
            with pytest.raises(ValueError):
                polarization.polarization_matrix(outputs, inputs)
        else:
            got = polarization.polarization_matrix(outputs, inputs)
            assert got.dtype == np.complex64 and got.shape == (len(outputs), len(inputs))
            np.testing.assert_array_equal(got, expected)
            solvable += 1
    assert 20 < solvable < len(gi.polarization_cases())
    from_c, to_c = polarization.polarization_matrices([1, 2, 3, 4], [9, 10, 11, 12])
    np.testing.assert_array_equal(from_c, g['from_circular'])
    np.testing.assert_array_equal(to_c, g['to_circular'])
    # IQUV from linear feeds: I = (XX + YY) / 2, V = (XY - YX) / 2i
    m = polarization.polarization_matrix(polarization.STOKES_IQUV, [9, 10, 11, 12])
    np.testing.assert_array_equal(m[0], [0.5, 0, 0, 0.5])
    np.testing.assert_array_equal(m[3], [0, -0.5j, 0.5j, 0])


@pytest.mark.parametrize('sx,sy,theta', [(2.0, 5.0, 1.0), (3.0, 1.5, -0.4), (4.1, 1.2, 2.5),
                                         (1.3, 1.3, 0.0)])
def test_fit_beam_recovers_a_gaussian(sx, sy, theta):
    """beam.fit_beam (beam.py:91-155) on an exactly Gaussian PSF returns that Gaussian, in the
    normalised (major, minor, theta mod pi) form of Beam; ``step`` scales the widths."""
    import math
    from katsdpimager_amd import beam
    H, W = 45, 39
    i0, i1 = np.meshgrid(np.arange(H) - H // 2, np.arange(W) - W // 2, indexing='ij')
    psf = beam._gaussian2d(i0.astype(float), i1.astype(float), sx, sy, theta).astype(np.float32)
    fwhm = math.sqrt(8 * math.log(2))
    for step in (1.0, 0.25):
        b = beam.fit_beam(psf, step=step)
        assert b.major == pytest.approx(max(sx, sy) * fwhm * step, rel=1e-5)
        assert b.minor == pytest.approx(min(sx, sy) * fwhm * step, rel=1e-5)
        if sx != sy:
            want = (theta + (math.pi / 2 if sx < sy else 0.0)) % math.pi
            assert abs((b.theta - want + math.pi / 2) % math.pi - math.pi / 2) < 1e-5
        # the model that FourierBeam consumes reproduces the PSF
        M = beam.beam_covariance_sqrt(b) / step
        inv = np.linalg.inv(M @ M)
        quad = inv[0, 0] * i0 * i0 + 2 * inv[0, 1] * i0 * i1 + inv[1, 1] * i1 * i1
        assert np.max(np.abs(np.exp(-0.5 * quad) - psf)) < 1e-4


def test_fit_beam_is_a_least_squares_stationary_point():
    """On a PSF that is not Gaussian (sidelobes, as a real synthesised beam) the result is the
    least-squares optimum over the samples above the threshold: no nearby parameters do better."""
    from katsdpimager_amd import beam
    H = W = 41
    i0, i1 = np.meshgrid(np.arange(H) - H // 2, np.arange(W) - W // 2, indexing='ij')
    r = np.hypot(i0 / 3.0, (i1 + 0.3 * i0) / 2.0)
    psf = np.sinc(r / 2.5) ** 2 * (1 + 0.05 * np.cos(i0))
    psf /= psf[H // 2, W // 2]
    b = beam.fit_beam(psf, threshold=0.05)
    mask = psf > 0.05
    x, y, v = i0[mask].astype(float), i1[mask].astype(float), psf[mask]

    def cost(sx, sy, th):
        return float(np.sum((beam._gaussian2d(x, y, sx, sy, th) - v) ** 2))
    m = b.model
    best = cost(m.x_stddev.value, m.y_stddev.value, m.theta)
    for d in (1e-3, -1e-3):
        assert cost(m.x_stddev.value + d, m.y_stddev.value, m.theta) >= best
        assert cost(m.x_stddev.value, m.y_stddev.value + d, m.theta) >= best
        assert cost(m.x_stddev.value, m.y_stddev.value, m.theta + d) >= best
    with pytest.raises(ValueError):
        beam.fit_beam(np.zeros((3, 3, 3)))


def test_extract_sky_model_matches_component_path():
    """predict.extract_sky_model (predict.py:30-70): n-1, IQUV selection in the image's
    polarization order, and the same sub-cell taper that extract_sky_image removes."""
    from katsdpimager_amd import parameters, polarization, predict

    class Model:
        def __init__(self, lmn, flux):
            self._lmn, self._flux = lmn, flux

        def __len__(self):
            return len(self._lmn)

        def lmn(self, phase_centre):
            assert phase_centre == 'centre'
            return self._lmn

        def flux_density(self, wavelength):
            return self._flux

    c = gi.make_config(256, 1e-4, 0.2, 2, 28, 8)
    ip, gp, _ = make_params(c)
    ip.fixed.polarizations = [polarization.STOKES_V, polarization.STOKES_I]
    pix = float(ip.pixel_size)
    pos = [(140, 100), (10, 250), (128, 128)]
    l = np.array([(x - 128) * pix for y, x in pos])
    m = np.array([(y - 128) * pix for y, x in pos])
    lmn = np.stack([l, m, np.sqrt(1 - l * l - m * m)], axis=1)
    lmn.flags.writeable = False           # the reference's models hand out read-only arrays
    iquv = np.array([[1.0, 0.1, 0.2, 0.3], [2.0, 0, 0, -0.5], [0.5, 0, 0, 0]])
    got_lmn, got_flux = predict.extract_sky_model(ip, gp, Model(lmn, iquv), 'centre')
    comps = {p: np.array([iquv[i, 3], iquv[i, 0]], np.float32) for i, p in enumerate(pos)}
    want_lmn, want_flux = predict.extract_sky_image(ip, gp, comps)
    assert got_lmn.dtype == np.float32 and got_flux.dtype == np.float32
    np.testing.assert_allclose(got_lmn, want_lmn, rtol=0, atol=1e-7)
    np.testing.assert_allclose(got_flux, want_flux, rtol=1e-6)
    assert np.all(lmn[:, 2] > 0.9)        # the model's array was not modified


def _read_fits(path):
    """Minimal reader for what io.write_fits writes: (header dict, history list, data)."""
    raw = open(path, 'rb').read()
    assert len(raw) % 2880 == 0
    header, history, pos = {}, [], 0
    while True:
        card = raw[pos:pos + 80].decode('ascii')
        pos += 80
        assert len(card) == 80
        if card.startswith('END'):
            break
        if card.startswith('HISTORY'):
            history.append(card[8:].rstrip())
            continue
        assert card[8:10] == '= '
        value = card[10:].split(' / ')[0].strip()
        if value.startswith("'"):
            header[card[:8].strip()] = value[1:-1].rstrip()
        elif value in ('T', 'F'):
            header[card[:8].strip()] = value == 'T'
        else:
            header[card[:8].strip()] = float(value) if ('.' in value or 'E' in value) else int(value)
    pos += -pos % 2880
    shape = tuple(header['NAXIS%d' % i] for i in range(header['NAXIS'], 0, -1))
    dtype = {-32: '>f4', -64: '>f8', 16: '>i2', 32: '>i4'}[header['BITPIX']]
    n = int(np.prod(shape))
    data = np.frombuffer(raw, dtype, n, pos).reshape(shape)
    assert not any(raw[pos + n * np.dtype(dtype).itemsize:])
    return header, history, data


def test_write_fits_image(tmp_path):
    """io.write_fits_image: header keywords and values of the reference's writer (io.py:126-181),
    RA axis reversed, degenerate frequency axis, big-endian data, 2880-byte blocks."""
    import math
    from katsdpimager_amd import beam, io, parameters, polarization
    pols = [polarization.STOKES_I, polarization.STOKES_Q, polarization.STOKES_U, polarization.STOKES_V]
    fixed = parameters.FixedImageParameters(pols, np.float32)
    ip = parameters.ImageParameters(fixed, 1.0, None, 0.21, None, pixel_size=2e-5, pixels=48)
    rs = np.random.RandomState(3)
    image = rs.standard_normal((4, 48, 48)).astype(np.float32)
    image[2, 5, 7] = np.nan
    b = beam.Beam(1.0, 3.0, 1.5, 0.4)
    path = str(tmp_path / 'image-%05d.fits')
    out, cards = io.write_fits_image(image, ip, path, 12, (1.25, -0.6), beam=b,
                                     extra_fits_headers={'OBJECT': 'test field', 'BUNIT': 'JY/BEAM'},
                                     date='2024-10-08T00:00:00.000')
    header, history, data = _read_fits(str(tmp_path / 'image-00012.fits'))
    assert data.shape == (1, 4, 48, 48) and data.dtype == np.dtype('>f4')
    np.testing.assert_array_equal(data[0], image[:, :, ::-1])
    np.testing.assert_array_equal(out, image[np.newaxis, :, :, ::-1])
    delt = math.degrees(math.asin(2e-5))
    expect = {
        'SIMPLE': True, 'BITPIX': -32, 'NAXIS': 4, 'NAXIS1': 48, 'NAXIS2': 48, 'NAXIS3': 4,
        'NAXIS4': 1, 'BUNIT': 'JY/BEAM', 'ORIGIN': 'katsdpimager_amd', 'TIMESYS': 'UTC',
        'DATE': '2024-10-08T00:00:00.000', 'CRPIX1': 24.0, 'CRPIX2': 25.0, 'CRPIX4': 1.0,
        'CDELT1': -delt, 'CDELT2': delt, 'CDELT4': 1.0, 'EQUINOX': 2000.0, 'RADESYS': 'FK5',
        'CUNIT1': 'deg', 'CUNIT2': 'deg', 'CUNIT4': 'Hz', 'CTYPE1': 'RA---SIN',
        'CTYPE2': 'DEC--SIN', 'CTYPE4': 'FREQ', 'CRVAL1': math.degrees(1.25),
        'CRVAL2': math.degrees(-0.6), 'CRVAL4': 299792458.0 / 0.21,
        'BMAJ': b.major * math.degrees(2e-5), 'BMIN': b.minor * math.degrees(2e-5),
        'BPA': math.degrees(b.theta), 'CTYPE3': 'STOKES', 'CRPIX3': 1.0, 'CRVAL3': 1.0,
        'CDELT3': 1.0, 'DATAMIN': float(np.nanmin(image)), 'DATAMAX': float(np.nanmax(image)),
        'OBJECT': 'test field',
    }
    assert header == expect
    assert history == ['Created by katsdpimager_amd']
    # polarization axis: circular products count downwards; unsorted or non-linear lists are refused
    assert io.fits_polarization_axis([polarization.STOKES_RR, polarization.STOKES_LL])[:2] == (-1, -1)
    assert io.fits_polarization_axis([polarization.STOKES_I, polarization.STOKES_V])[:2] == (1, 3)
    with pytest.raises(ValueError):
        io.fits_polarization_axis([polarization.STOKES_I, polarization.STOKES_Q, polarization.STOKES_V])
    fixed2 = parameters.FixedImageParameters([polarization.STOKES_Q, polarization.STOKES_I], np.float32)
    ip2 = parameters.ImageParameters(fixed2, 1.0, None, 0.21, None, pixel_size=2e-5, pixels=48)
    with pytest.raises(ValueError):
        io.write_fits_image(image[:2], ip2, str(tmp_path / 'x.fits'), 0, (0.0, 0.0))
    # an all-NaN image has no DATAMIN / DATAMAX (io.py:176-179); no place for the channel is fine
    _, cards = io.write_fits_image(np.full((1, 48, 48), np.nan, np.float32),
                                   parameters.ImageParameters(
                                       parameters.FixedImageParameters([polarization.STOKES_I], np.float32),
                                       1.0, None, 0.21, None, pixel_size=2e-5, pixels=48),
                                   str(tmp_path / 'nan.fits'), 3, (0.0, 0.0), bunit=None)
    keys = [k for k, _ in cards]
    assert 'DATAMIN' not in keys and 'BUNIT' not in keys and 'BMAJ' not in keys
    assert _read_fits(str(tmp_path / 'nan.fits'))[2].shape == (1, 1, 48, 48)


def test_write_fits_grid(tmp_path):
    """io.write_fits_grid (io.py:228-270): float32 [complex][polarization][v][u], axes in metres
    about the centre cell, STOKES axis with the reference's permutation, COMPLEX axis."""
    from katsdpimager_amd import io, parameters, polarization
    pols = [polarization.STOKES_Q, polarization.STOKES_I]       # FITS order is I, Q: permuted
    fixed = parameters.FixedImageParameters(pols, np.float32)
    ip = parameters.ImageParameters(fixed, 1.0, None, 0.21, None, pixel_size=2e-5, pixels=64)
    rs = np.random.RandomState(5)
    grid = (rs.standard_normal((2, 30, 30)) + 1j * rs.standard_normal((2, 30, 30))).astype(np.complex64)
    out, cards = io.write_fits_grid(grid, ip, str(tmp_path / 'grid-%03d.fits'), 7)
    header, history, data = _read_fits(str(tmp_path / 'grid-007.fits'))
    assert data.shape == (2, 2, 30, 30) and data.dtype == np.dtype('>f4')
    np.testing.assert_array_equal(data[0], grid.real[[1, 0]])
    np.testing.assert_array_equal(data[1], grid.imag[[1, 0]])
    cell = float(ip.cell_size)
    assert header == {
        'SIMPLE': True, 'BITPIX': -32, 'NAXIS': 4, 'NAXIS1': 30, 'NAXIS2': 30, 'NAXIS3': 2,
        'NAXIS4': 2, 'BUNIT': 'Jy', 'ORIGIN': 'katsdpimager_amd', 'CUNIT1': 'm', 'CRPIX1': 16.0,
        'CRVAL1': 0.0, 'CDELT1': cell, 'CUNIT2': 'm', 'CRPIX2': 16.0, 'CRVAL2': 0.0, 'CDELT2': cell,
        'CTYPE3': 'STOKES', 'CRPIX3': 1.0, 'CRVAL3': 1.0, 'CDELT3': 1.0, 'CTYPE4': 'COMPLEX',
        'CRPIX4': 1.0, 'CRVAL4': 1.0, 'CDELT4': 1.0}
    with pytest.raises(ValueError):
        io.write_fits_grid(grid.real, ip, str(tmp_path / 'x.fits'), 0)
    fixed3 = parameters.FixedImageParameters(
        [polarization.STOKES_I, polarization.STOKES_Q, polarization.STOKES_V], np.float32)
    ip3 = parameters.ImageParameters(fixed3, 1.0, None, 0.21, None, pixel_size=2e-5, pixels=64)
    with pytest.raises(ValueError):
        io.write_fits_grid(np.zeros((3, 8, 8), np.complex64), ip3, str(tmp_path / 'y.fits'), 0)


def _fits_cards_and_block(path):
    """(ordered [(key, value)] without the mandatory SIMPLE / BITPIX / NAXIS* cards, raw data bytes)"""
    header, history, data = _read_fits(path)
    raw = open(path, 'rb').read()
    cards, pos = [], 0
    while not raw[pos:pos + 3] == b'END':
        key = raw[pos:pos + 8].decode('ascii').strip()
        if key == 'HISTORY':
            cards.append((key, raw[pos + 8:pos + 80].decode('ascii').rstrip()))
        elif not (key in ('SIMPLE', 'BITPIX') or key.startswith('NAXIS')):
            cards.append((key, header[key]))
        pos += 80
    start = (pos + 80) + (-(pos + 80) % 2880)
    return cards, raw[start:start + data.nbytes], header


def test_fits_writers_vs_restated_reference(tmp_path):
    """The FITS files against a card-by-card restatement of io.py:88-270 (oracle.fits_image /
    fits_grid: what the reference hands to astropy.io.fits): same cards in the same order, same
    values to the last bit, and the raw big-endian data block byte for byte.  NOT pinned by a
    reference-produced file: astropy is absent, the reference's writer cannot run here and has no
    test of its own."""
    import math
    from oracle import kimg_oracle as orc
    from katsdpimager_amd import beam, io, parameters, polarization as pol
    rs = np.random.RandomState(11)
    cases = [
        ([pol.STOKES_I, pol.STOKES_Q, pol.STOKES_U, pol.STOKES_V], (0.3, -1.1), beam.Beam(1.0, 4.2, 2.2, -0.7),
         'Jy/beam', {'OBJECT': 'J1939-6342', 'BUNIT': 'JY/BEAM', 'OBSERVER': 'x'}),
        ([pol.STOKES_I], (5.9, 0.2), None, 'Jy/beam', None),
        ([pol.STOKES_RR, pol.STOKES_LL], (0.0, -0.5), None, None, None),
        ([pol.STOKES_XX, pol.STOKES_YY], (1.0, 0.5), beam.Beam(1.0, 2.0, 2.0, 0.0), 'Jy/beam', None),
    ]
    for n, (pols, centre, b, bunit, extra) in enumerate(cases):
        fixed = parameters.FixedImageParameters(pols, np.float32)
        ip = parameters.ImageParameters(fixed, 1.0, None, 0.2142, None, pixel_size=3.1e-5, pixels=40)
        image = rs.standard_normal((len(pols), 40, 40)).astype(np.float32)
        image[0, 3, 4] = np.nan
        path = str(tmp_path / ('i%d.fits' % n))
        try:
            io.write_fits_image(image, ip, path, 0, centre, beam=b, bunit=bunit, extra_fits_headers=extra)
        except ValueError:
            # (XX, YY) is FITS (-6, -5): the file would need the permuted order, which the
            # reference's write_fits_image silently ignores (io.py:172) and this writer refuses
            assert pols == [pol.STOKES_XX, pol.STOKES_YY]
            continue
        want_cards, want_data = orc.fits_image(
            image, ip.pixel_size, ip.wavelength, pols, centre,
            None if b is None else (b.major, b.minor, b.theta), bunit, extra)
        cards, block, header = _fits_cards_and_block(path)
        assert header['BITPIX'] == -32 and header['NAXIS'] == 4
        assert [header['NAXIS%d' % i] for i in (4, 3, 2, 1)] == list(want_data.shape)
        # ours names itself in ORIGIN / HISTORY and stamps DATE; everything else is identical
        strip = lambda cs: [(k, v) for k, v in cs if k not in ('ORIGIN', 'HISTORY', 'DATE')]
        assert [k for k, _ in cards if k != 'DATE'] == [k for k, _ in want_cards]
        for (k, v), (wk, wv) in zip(strip(cards), strip(want_cards)):
            assert k == wk and type(v) is type(wv) and v == wv, (k, v, wv)
        assert dict(cards)['ORIGIN'] == 'katsdpimager_amd'
        assert block == np.ascontiguousarray(want_data, '>f4').tobytes()
    # grids: every linear polarization list, permuted where FITS wants another order
    for n, pols in enumerate([[pol.STOKES_I], [pol.STOKES_Q, pol.STOKES_I],
                              [pol.STOKES_LL, pol.STOKES_RR], [pol.STOKES_XX, pol.STOKES_YY],
                              [pol.STOKES_V, pol.STOKES_U, pol.STOKES_Q, pol.STOKES_I]]):
        fixed = parameters.FixedImageParameters(pols, np.float32)
        ip = parameters.ImageParameters(fixed, 1.0, None, 0.2142, None, pixel_size=3.1e-5, pixels=40)
        g = (rs.standard_normal((len(pols), 22, 22)) + 1j * rs.standard_normal((len(pols), 22, 22))) \
            .astype(np.complex64)
        path = str(tmp_path / ('g%d.fits' % n))
        io.write_fits_grid(g, ip, path, 0)
        want_cards, want_data = orc.fits_grid(g, ip.cell_size, pols)
        cards, block, header = _fits_cards_and_block(path)
        assert [header['NAXIS%d' % i] for i in (4, 3, 2, 1)] == list(want_data.shape)
        assert [k for k, _ in cards] == [k for k, _ in want_cards]
        for (k, v), (wk, wv) in zip(cards, want_cards):
            if k != 'ORIGIN':
                assert type(v) is type(wv) and v == wv, (k, v, wv)
        assert block == np.ascontiguousarray(want_data, '>f4').tobytes()
    assert math.isclose(dict(want_cards)['CDELT1'], ip.cell_size)


def test_process_channel_stream_bounds_live_jobs(monkeypatch):
    """frontend.process_channel_stream / parallel.image_assigned_channels: jobs are made lazily, at
    most `workers` exist at a time (a rank with hundreds of channels must not hold
    hundreds of imagers), results come back in channel order, errors propagate."""
    import threading
    import time as _time
    from katsdpimager_amd import frontend, parallel
    lock = threading.Lock()
    live, peak, made = set(), [0], []

    class Job(dict):
        def __del__(self):
            with lock:
                live.discard(self['channel'])

    def make_job(channel, worker):
        assert 0 <= worker < 3
        with lock:
            live.add(channel)
            made.append((channel, worker))
            peak[0] = max(peak[0], len(live))
        return Job(channel=channel)

    def fake_process_channel(channel, clean_batcher=None):
        _time.sleep(0.002)
        if channel == 77:
            raise RuntimeError('boom')
        return channel * 2
    monkeypatch.setattr(frontend, 'process_channel', fake_process_channel)
    out = frontend.process_channel_stream(make_job, range(40), workers=3)
    assert out == [2 * c for c in range(40)]
    assert peak[0] <= 3 and len(made) == 40 and len({w for _, w in made}) >= 2
    assert frontend.process_channel_stream(lambda c: Job(channel=c), [5, 3], workers=1) == [10, 6]
    with pytest.raises(RuntimeError):
        frontend.process_channel_stream(lambda c: Job(channel=c), [1, 77, 2], workers=2)
    assert frontend.process_channel_stream(make_job, [], workers=2) == []
    got = parallel.image_assigned_channels(make_job, 7, workers=3)
    assert got == {c: 2 * c for c in range(7)}


def test_channel_stream_makes_jobs_on_the_streams_device(monkeypatch):
    """process_channel_stream: everything of a channel -- the making of its job included, where an
    imager may be built lazily -- runs with the stream's device current in the worker thread (new
    host threads start on HIP device 0), and the imagers in flight together are told their share of
    the CUs per imager (no process-wide setting)."""
    import contextlib
    import threading
    import torch
    from katsdpimager_amd import frontend
    current = threading.local()
    seen = []

    @contextlib.contextmanager
    def fake_device(index):
        before = getattr(current, 'index', 0)
        current.index = index
        try:
            yield
        finally:
            current.index = before
    monkeypatch.setattr(torch.cuda, 'device', fake_device)

    class FakeImager:
        def __init__(self):
            self.cus = []

        def set_window_cus(self, cus):
            self._window_cus = cus
            self.cus.append(cus)
    imagers = {}

    def make_job(channel, worker):
        seen.append(('make', channel, getattr(current, 'index', 0)))
        imagers[channel] = FakeImager()
        return dict(channel=channel, imager=imagers[channel])

    def fake_process_channel(channel, imager, clean_batcher=None):
        seen.append(('run', channel, getattr(current, 'index', 0), imager._window_cus))
        return channel
    monkeypatch.setattr(frontend, 'process_channel', fake_process_channel)
    assert frontend.process_channel_stream(make_job, range(6), workers=3, device=5,
                                           batch_clean=False) == list(range(6))
    assert all(e[2] == 5 for e in seen) and len(seen) == 12
    assert all(e[3] == frontend.WINDOW_CUS_SHARED for e in seen if e[0] == 'run')
    assert all(im.cus == [frontend.WINDOW_CUS_SHARED, 0] for im in imagers.values())
    # one channel at a time: the whole device
    seen.clear()
    frontend.process_channel_stream(make_job, [9], workers=4, device=2, batch_clean=False)
    assert seen == [('make', 9, 2), ('run', 9, 2, 0)]


def test_window_cus_setting():
    """kimg_set_window_cus / kimg_get_window_cus (host state only): 0 means all 256, values outside
    0 ... 256 are refused and leave the setting alone."""
    from katsdpimager_amd._lib import lib
    handle = lib()
    assert handle.kimg_get_window_cus() == 256
    try:
        assert handle.kimg_set_window_cus(192) == 0 and handle.kimg_get_window_cus() == 192
        assert handle.kimg_set_window_cus(300) != 0 and handle.kimg_get_window_cus() == 192
        assert handle.kimg_set_window_cus(-1) != 0 and handle.kimg_get_window_cus() == 192
        assert handle.kimg_set_window_cus(0) == 0 and handle.kimg_get_window_cus() == 256
    finally:
        handle.kimg_set_window_cus(0)


def test_clean_batcher_rendezvous(monkeypatch):
    """clean.CleanBatcher (host logic only; the device loop is faked): channels that arrive
    together share one enqueue, a patch too large for the one-launch form runs alone, a thread
    that leaves is not waited for, and a lone arrival proceeds after the timeout."""
    import threading
    import time as _time
    from katsdpimager_amd import clean
    launched = []

    class FakeClean:
        def __init__(self, name, key='a'):
            self.name, self.key, self.command_queue = name, key, object()

        def _batch_key(self):
            return self.key

        def run_cycles(self, patch, threshold, max_cycles, collect=True):
            launched.append(('solo', self.name))
            self.how = 'solo'

        how = 'batched'

        def _collect_cycles(self):
            return [(self.how, self.name)]

    class FakeQueue:
        def finish(self):
            pass

    def fake_enqueue(cleans, patches, thresholds, max_cycles, queue=None):
        launched.append(('batch', tuple(c.name for c in cleans)))
        return FakeQueue()
    monkeypatch.setattr(clean, 'enqueue_cycles_batch', fake_enqueue)
    monkeypatch.setattr(clean, 'batch_supported', lambda c, patch: patch[1] <= 500)

    def run(batcher, items, delays=None):
        out = {}

        def work(i, item):
            if delays:
                _time.sleep(delays[i])
            out[item[0].name] = batcher.run_cycles(*item)
        threads = [threading.Thread(target=work, args=(i, it)) for i, it in enumerate(items)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(10)
            assert not t.is_alive()
        return out
    small = (1, 111, 133)
    b = clean.CleanBatcher(3, timeout=5.0)
    cl = [FakeClean(n) for n in 'xyz']
    out = run(b, [(c, small, 0.0, 100) for c in cl], [0.0, 0.01, 0.02])
    assert out == {n: [('batched', n)] for n in 'xyz'}
    assert [k for k, _ in launched] == ['batch'] and sorted(launched[0][1]) == list('xyz')
    assert b.batches == [(3, 100)]
    # channels whose patch lets a launch plan several components run on their own
    launched.clear()
    monkeypatch.setattr(clean, 'prefers_solo', lambda c, patch, cycles: c.name in 'xyz')
    b5 = clean.CleanBatcher(5, timeout=5.0)
    out = run(b5, [(FakeClean(n), small, 0.0, 100) for n in 'pqxyz'])
    # (p and q still share their launches)
    assert sorted(k for k, _ in launched) == ['batch', 'solo', 'solo', 'solo']
    assert sorted(n for k, n in launched if k == 'solo') == list('xyz')
    assert [sorted(n) for k, n in launched if k == 'batch'] == [['p', 'q']] and b5.batches == [(2, 100)]
    assert out == {n: [('solo' if n in 'xyz' else 'batched', n)] for n in 'pqxyz'}
    monkeypatch.setattr(clean, 'prefers_solo', lambda c, patch, cycles: False)
    # a large patch and a different image shape go alone; the two compatible ones share
    launched.clear()
    items = [(FakeClean('p'), small, 0.0, 10), (FakeClean('q'), (1, 711, 675), 0.0, 10),
             (FakeClean('r', key='b'), small, 0.0, 10)]
    out = run(b, items)
    assert out == {'p': [('solo', 'p')], 'q': [('solo', 'q')], 'r': [('solo', 'r')]}
    # one party left: the remaining two do not wait for it
    launched.clear()
    b.leave()
    t0 = _time.monotonic()
    out = run(b, [(FakeClean('m'), small, 0.0, 7), (FakeClean('n'), small, 0.0, 9)])
    assert _time.monotonic() - t0 < 2.0 and launched == [('batch', ('m', 'n'))] or \
        launched == [('batch', ('n', 'm'))]
    assert b.batches[-1] == (2, 9)
    # a lone arrival among two parties proceeds after the timeout, alone
    b2 = clean.CleanBatcher(2, timeout=0.05)
    launched.clear()
    assert b2.run_cycles(FakeClean('w'), small, 0.0, 5) == [('solo', 'w')]
    assert b2.run_cycles(FakeClean('w'), small, 0.0, 0) == []


def test_clean_batcher_phased(monkeypatch):
    """CleanBatcher(phased=True): the throughput-bound stages run one channel at a time
    (device_phase), and the rendezvous waits only for threads on their way from such a stage to
    their cycles -- not for one that is inside (or queueing for) a stage, nor for one that
    declared itself idle."""
    import threading
    import time as _time
    from katsdpimager_amd import clean
    launched = []

    class FakeClean:
        def __init__(self, name):
            self.name, self.command_queue = name, object()

        def _batch_key(self):
            return 'k'

        def run_cycles(self, patch, threshold, max_cycles, collect=True):
            launched.append(('solo', self.name))

        def _collect_cycles(self):
            return [self.name]

    class FakeQueue:
        def finish(self):
            pass

    def fake_enqueue(cleans, patches, thresholds, max_cycles, queue=None):
        launched.append(('batch', tuple(sorted(c.name for c in cleans))))
        return FakeQueue()
    monkeypatch.setattr(clean, 'enqueue_cycles_batch', fake_enqueue)
    monkeypatch.setattr(clean, 'batch_supported', lambda c, patch: True)
    small = (1, 111, 133)
    b = clean.CleanBatcher(2, timeout=5.0, phased=True)
    inside, release, order = threading.Event(), threading.Event(), []

    def gridding():
        with b.device_phase():
            order.append('a in')
            inside.set()
            release.wait(5)
            order.append('a out')
        b.run_cycles(FakeClean('a'), small, 0.0, 3)
    ta = threading.Thread(target=gridding)
    ta.start()
    assert inside.wait(5)
    # 'a' is inside its stage: 'b' does not wait for it (the timeout is 5 s)
    t0 = _time.monotonic()
    assert b.run_cycles(FakeClean('b'), small, 0.0, 3) == ['b']
    assert _time.monotonic() - t0 < 2.0 and launched == [('solo', 'b')]
    # the stages exclude each other
    entered = []

    def second_stage():
        with b.device_phase():
            entered.append(_time.monotonic())
    tb = threading.Thread(target=second_stage)
    tb.start()
    _time.sleep(0.05)
    assert not entered
    release.set()
    ta.join(5)
    tb.join(5)
    assert entered and not ta.is_alive() and not tb.is_alive()
    assert launched[-1] == ('solo', 'a') and order == ['a in', 'a out']
    # a thread that left a stage is waited for: the two share their launches
    launched.clear()
    out = {}

    def staged(name, delay):
        with b.device_phase():
            pass
        _time.sleep(delay)
        out[name] = b.run_cycles(FakeClean(name), small, 0.0, 4)
    threads = [threading.Thread(target=staged, args=(n, d)) for n, d in (('c', 0.0), ('d', 0.2))]
    threads[1].start()          # 'd' is through its stage (and expected) before 'c' arrives
    _time.sleep(0.05)
    threads[0].start()
    for t in threads:
        t.join(5)
        assert not t.is_alive()
    assert launched == [('batch', ('c', 'd'))] and out == {'c': ['c'], 'd': ['d']}
    assert b.batches == [(2, 4)]
    # ... unless it says it is not coming
    launched.clear()

    def staged_then_idle():
        with b.device_phase():
            pass
        b.idle()
    t = threading.Thread(target=staged_then_idle)
    t.start()
    t.join(5)
    t0 = _time.monotonic()
    assert b.run_cycles(FakeClean('e'), small, 0.0, 2) == ['e']
    assert _time.monotonic() - t0 < 2.0 and launched == [('solo', 'e')]


def _loader_arrays(rows=600, channels=3, pols=2, antennas=5, seed=8):
    from katsdpimager_amd import loader, polarization
    rs = np.random.RandomState(seed)
    i, j = np.triu_indices(antennas, 1)
    nb = len(i)
    dumps = rows // nb
    rows = dumps * nb
    # time order: every baseline of dump 0, then of dump 1, ...
    baseline = np.tile(i * antennas + j, dumps)
    uvw = rs.uniform(-900, 900, (rows, 3)).astype(np.float32)
    vis = (rs.standard_normal((rows, channels, pols))
           + 1j * rs.standard_normal((rows, channels, pols))).astype(np.complex64)
    weights = rs.uniform(0.5, 2, (rows, channels, pols)).astype(np.float32)
    freq = 1.4e9 + 1e6 * np.arange(channels)
    return loader.LoaderArrays(uvw, vis, weights, baseline, freq,
                               [polarization.STOKES_XX, polarization.STOKES_YY],
                               phase_centre=(0.3, -0.5), longest_baseline=1500.0), nb, dumps


def test_loader_blocks_are_baseline_sorted(tmp_path):
    """loader.LoaderArrays.data_iter shapes the stream like loader_ms.py:377-467: blocks of at most
    max_chunk_vis visibilities over the selected channels, each block in a stable sort by baseline
    (so a baseline's dumps are adjacent and in time order), channel axis first; round trip
    through the .npz file; loader.data_iter truncates at vis_limit rows (loader.py:36-59)."""
    from katsdpimager_amd import loader
    ds, nb, dumps = _loader_arrays()
    path = str(tmp_path / 'vis.npz')
    ds.save(path)
    ds2 = loader.load(path)
    assert ds2.num_channels() == 3 and ds2.polarizations() == ds.polarizations()
    assert ds2.phase_centre() == (0.3, -0.5) and ds2.longest_baseline() == 1500.0
    assert ds2.array_parameters().longest_baseline == 1500.0 and not ds2.has_feed_angles()
    assert ds2.wavelength(1) == pytest.approx(299792458.0 / 1.401e9)
    with pytest.raises(ValueError):
        loader.load(str(tmp_path / 'vis.ms'))
    rows = nb * dumps
    per_block = 7 * nb + 3                  # blocks cut inside a dump
    seen = 0
    for chunk in ds2.data_iter(1, 3, per_block * 2):
        n = len(chunk['uvw'])
        assert n <= per_block and chunk['vis'].shape == (2, n, 2) == chunk['weights'].shape
        assert np.all(np.diff(chunk['baselines']) >= 0)
        # stable: inside a baseline the rows keep their time order = increasing row index
        start = seen
        order = np.argsort(ds.baseline[start:start + n], kind='stable') + start
        np.testing.assert_array_equal(chunk['uvw'], ds.uvw[order])
        np.testing.assert_array_equal(chunk['vis'], np.swapaxes(ds.vis[order][:, 1:3], 0, 1))
        seen += n
        assert chunk['progress'] == seen and chunk['total'] == rows
    assert seen == rows
    got = sum(len(c['uvw']) for c in loader.data_iter(ds2, 100, per_block * 3, 0, 3))
    assert got == 100
    with pytest.raises(ValueError):
        next(ds2.data_iter(2, 2))


def test_trace_ranges_off_and_on():
    """trace.range is free when off, pushes / pops roctx ranges when on (libroctx64 ships with
    ROCm; a missing library leaves tracing off)."""
    from katsdpimager_amd import trace
    trace.disable()
    with trace.range('nothing'):
        pass
    assert not trace.enabled()
    assert trace.enable('libdoes_not_exist_anywhere.so') is False
    if trace.enable():
        with trace.range('outer'):
            with trace.range('inner'):
                pass
        trace.disable()
    assert not trace.enabled()


class _ShortCutRecorder(_Recorder):
    """An imager that offers what frontend.process_channel may use beyond the reference's calls."""
    one_call_major_cycles = True
    device_psf_stage = True

    def psf_patch_start(self):
        self.calls.append(('psf_patch_start',))
        return 'started'

    def psf_patch_finish(self, started):
        assert started == 'started'
        self.calls.append(('psf_patch_finish',))
        return (1, 9, 11), np.array([0.5], np.float32)

    def clean_major_cycles(self, psf_patch, noise_threshold, left_for_next, max_cycles, batcher=None):
        self.calls.append(('clean_major_cycles', max_cycles, round(float(left_for_next), 6)))
        # (first peak, cycles done -- the first one included)
        return self._peaks.pop(0), 1 + min(self._cycles, max_cycles - 1)


def test_frontend_driver_short_cuts():
    """With an imager that offers them the driver scales the PSF by a value that stays on the device,
    starts the PSF patch's read-back before the first dirty image and looks at it after, and runs a
    major cycle's minor cycles in one call; what it reports is what the reference's sequence reports."""
    from katsdpimager_amd import frontend, weight
    image_p, grid_p, clean_p = _driver_params()
    im = _ShortCutRecorder(peaks=[1.0, 1.0], cycles_before_threshold=3)
    out = frontend.process_channel(_DeviceReader([5, 0, 3]), 0, im, image_p, grid_p, clean_p,
                                   weight.WeightType.UNIFORM, 4, 2, True)
    names = [c[0] for c in im.calls]
    # the PSF stage: no read of the central pixel, no psf_patch() with its read-back
    assert 'psf_patch' not in names and 'scale_dirty' not in names[:names.index('psf_patch_finish')]
    assert names.index('scale_dirty_by_centre') < names.index('dirty_to_psf') < names.index('psf_patch_start')
    # ... the first dirty image is gridded before the patch is looked at
    first_grid_after_psf = [i for i, n in enumerate(names) if n == 'grid' and i > names.index('psf_patch_start')][0]
    assert first_grid_after_psf < names.index('psf_patch_finish')
    assert names.index('scale_dirty_by_kept') < names.index('psf_patch_finish')
    # ... and the later major cycles scale by the value the host now has
    assert 'scale_dirty' in names[names.index('psf_patch_finish'):]
    # one call per major cycle, first cycle included; no clean_cycle, no clean_cycles
    assert names.count('clean_major_cycles') == 2 and 'clean_cycle' not in names and 'clean_cycles' not in names
    assert [c for c in im.calls if c[0] == 'clean_major_cycles'][0][1:] == (10, 0.15)
    assert out['psf_patch'] == (1, 9, 11) and out['major'] == 2
    # 3 cycles after the first found a peak, the next one did not: counted as frontend.py:579-582 does
    assert out['minor'] == 2 * 4 and out['peaks'] == [1.0, 1.0]
    np.testing.assert_array_equal(out['scale'], np.array([0.5], np.float32))
    # the reference's own sequence reports the same
    ref = _Recorder(peaks=[1.0, 1.0], cycles_before_threshold=3)
    want = frontend.process_channel(_DeviceReader([5, 0, 3]), 0, ref, image_p, grid_p, clean_p,
                                    weight.WeightType.UNIFORM, 4, 2, True)
    assert want['minor'] == out['minor'] and want['major'] == out['major'] and want['peaks'] == out['peaks']


def test_process_channel_refuses_an_imager_made_for_other_parameters():
    import types
    from katsdpimager_amd import frontend, weight
    image_p, grid_p, clean_p = _driver_params()
    im = _Recorder(peaks=[1.0], cycles_before_threshold=1)
    im.image_parameters = types.SimpleNamespace(wavelength=0.21, fixed=image_p.fixed)
    im.grid_parameters = grid_p
    with pytest.raises(ValueError, match='wavelength'):
        frontend.process_channel(_HostReader([5]), 0, im, image_p, grid_p, clean_p,
                                 weight.WeightType.UNIFORM, 4, 1, True)
    im.image_parameters = image_p
    im.grid_parameters = types.SimpleNamespace(fixed=grid_p.fixed, w_slices=2)
    with pytest.raises(ValueError, match='w_slices'):
        frontend.process_channel(_HostReader([5]), 0, im, image_p, grid_p, clean_p,
                                 weight.WeightType.UNIFORM, 4, 1, True)
    im.grid_parameters = grid_p
    assert frontend.process_channel(_HostReader([5]), 0, im, image_p, grid_p, clean_p,
                                    weight.WeightType.UNIFORM, 4, 1, True) is not None


def test_clean_batcher_major_cycles_of_a_channel_on_its_own():
    """CleanBatcher.run_major_cycles: a channel whose cycles run on their own takes the one call (and
    the batcher's count of running channels comes back to where it was); one whose cycles share
    launches is told to run the first cycle itself."""
    from katsdpimager_amd import clean

    class Solo:
        template = None
        calls = []

        def run_major_cycles(self, patch, noise_threshold, left, max_cycles):
            self.calls.append((patch, max_cycles))
            return 7, 1.5

    b = clean.CleanBatcher(2)
    solo = Solo()
    orig = clean.prefers_solo
    try:
        clean.prefers_solo = lambda c, patch, cycles: c is solo
        assert b.run_major_cycles(solo, (1, 9, 9), 0.1, 0.15, 100) == (7, 1.5)
        assert solo.calls == [((1, 9, 9), 100)] and b._cleaning == 0
        assert b.run_major_cycles(object(), (1, 9, 9), 0.1, 0.15, 100) is None
        assert b.run_major_cycles(solo, (1, 9, 9), 0.1, 0.15, 0) is None
    finally:
        clean.prefers_solo = orig
