"""CPU restatement of the katsdpimager imaging hot path -- TEST INFRASTRUCTURE ONLY.

This module is the parity *oracle*: a plain numpy / C restatement of the
reference's CPU ("Host") algorithms for the per-channel imaging path.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  The product package (``katsdpimager_amd``) never does.

Parity status: PINNED.  Every function here is checked in
``tests/test_oracle_golden.py`` against golden vectors in ``tests/golden/``
that were produced by importing the reference itself in the build container
(``tools/gen_golden.py``; the reference's third-party deps numba /
katsdpsigproc / astropy are replaced by the stand-ins in
``tools/oracle_shims``), plus the known-answer vectors of the reference's
own unit tests (``tests/test_oracle_known_answers.py``).

All ``file:line`` citations are relative to the reference checkout
(ska-sa/katsdpimager @ 2024_10_08).
"""

import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

CLEAN_I = 0          # clean.py:29
CLEAN_SUMSQ = 1      # clean.py:31
MEDIAN_TO_RMS = 1.4826022185056031   # clean.py:34

NATURAL, UNIFORM, ROBUST = 0, 1, 2   # weight.py:55-58


# --------------------------------------------------------------------------
# fast_math.py:7-16
# --------------------------------------------------------------------------
def expj2pi(x):
    """e^{2 pi i x} with range reduction, f32->c64 / f64->c128 (fast_math.py:7-16)."""
    x = np.asarray(x)
    if x.dtype == np.float32:
        # numba semantics: the reduction x - rint(x) is float32, the product
        # with the float64 constant 2*pi and cos/sin are float64, result -> c64.
        y = 2 * np.pi * (x - np.rint(x)).astype(np.float64)
        return (np.cos(y) + 1j * np.sin(y)).astype(np.complex64)
    x = x.astype(np.float64)
    y = 2 * np.pi * (x - np.rint(x))
    return np.cos(y) + 1j * np.sin(y)


# --------------------------------------------------------------------------
# grid.py:136-334  convolution kernel generation
# --------------------------------------------------------------------------
def kaiser_bessel(x, width, beta):
    """grid.py:136-155."""
    param = 1 - (2 * x / width) ** 2
    values = np.i0(beta * np.sqrt(np.maximum(0, param))) / np.i0(beta)
    return np.select([param >= 0], [values])


def kaiser_bessel_fourier(f, width, beta):
    """grid.py:158-184."""
    alpha = beta / math.pi
    return width / np.i0(beta) * np.sinc(
        np.lib.scimath.sqrt((width * f) ** 2 - alpha * alpha)).real


def kernel_beta(antialias_width):
    """grid.py:374-378."""
    return 1.2 * math.pi * math.sqrt(0.25 * antialias_width ** 2 - 1.0)


def antialias_w_kernel(cell_wavelengths, w, width, oversample, antialias_width,
                       image_oversample, beta):
    """grid.py:235-334.  Returns complex128 [len(w)][oversample][width]."""
    w = np.asarray(w, np.float64)
    out_pixels = oversample * width
    assert out_pixels % 2 == 0
    pixels = out_pixels * image_oversample
    uv_width = width * cell_wavelengths * image_oversample
    image_step = 1 / uv_width
    l = (np.arange(pixels) - (pixels // 2)) * image_step
    shift_by = -0.5 * cell_wavelengths / oversample
    scale_l = l * cell_wavelengths
    aa_factor = cell_wavelengths * kaiser_bessel_fourier(scale_l, antialias_width, beta)
    shift_arg = shift_by * l
    l2 = l * l
    l4 = l2 * l2
    w_arg = np.outer(-w, -0.5 * l2 - 5.0 / 24.0 * l4)
    image_values = aa_factor * expj2pi(w_arg + shift_arg)
    uv_values = np.fft.fft(np.fft.ifftshift(image_values, axes=-1), axis=-1) * image_step
    uv_values = np.concatenate(
        (uv_values[..., -(out_pixels // 2):], uv_values[..., :(out_pixels // 2)]), axis=-1)
    kernel = np.reshape(uv_values, np.shape(w) + (width, oversample))[..., ::-1]
    return np.ascontiguousarray(np.swapaxes(kernel, 1, 2))


def convolution_kernel(cell_size, wavelength, max_w, w_slices, w_planes, oversample,
                       kernel_width, antialias_width, image_oversample):
    """ConvolutionKernel.__init__ (grid.py:358-389) -> (complex64 table, beta)."""
    cell_wavelengths = float(cell_size / wavelength)
    w_slice_wavelengths = float(max_w / (w_slices * wavelength))
    w_plane_wavelengths = w_slice_wavelengths / w_planes
    beta = kernel_beta(antialias_width)
    max_w_wavelengths = (w_slice_wavelengths - w_plane_wavelengths) * 0.5
    ws = np.linspace(-max_w_wavelengths, max_w_wavelengths, w_planes)
    data = antialias_w_kernel(cell_wavelengths, ws, kernel_width, oversample,
                              antialias_width, image_oversample, beta)
    return data.astype(np.complex64), beta


def taper(N, antialias_width, beta, oversample, dtype=np.float64):
    """ConvolutionKernel.taper (grid.py:404-423)."""
    x = np.arange(N) / N - 0.5
    out = kaiser_bessel_fourier(x, antialias_width, beta)
    out = out * np.sinc(x / oversample)
    return out.astype(dtype)


# --------------------------------------------------------------------------
# preprocess.cpp:313-323, 435-507   UVW quantisation
# --------------------------------------------------------------------------
def subpixel_coord(x, oversample):
    """preprocess.cpp:313-323 (vectorised).  x float32 array -> (pixel, subpixel) int16."""
    x = np.asarray(x, np.float32)
    xs = np.floor(x * np.float32(oversample)).astype(np.int32)
    pixel = np.trunc(xs / oversample).astype(np.int32)   # C integer division truncates
    sub = xs - pixel * oversample
    neg = sub < 0
    pixel = np.where(neg, pixel - 1, pixel)
    sub = np.where(neg, sub + oversample, sub)
    return pixel.astype(np.int16), sub.astype(np.int16)


def quantise_uvw(uvw, vis, weights, cell_size, max_w, w_slices, w_planes, oversample):
    """The per-visibility arithmetic of visibility_collector::add_impl2 with an
    identity Mueller matrix (preprocess.cpp:435-507): w<0 flip, weight
    pre-multiply, NaN squash, quantisation.  No compression / sorting.

    uvw float32 [N][3] (same length unit as cell_size/max_w), vis c64 [N][P],
    weights f32 [N][P].  Returns dict of arrays (uv, sub_uv, w_plane, w_slice,
    vis, weights); rows with any zero weight are dropped as in :446-454 + compress.
    """
    uvw = np.asarray(uvw, np.float32)
    vis = np.asarray(vis, np.complex64).copy()
    weights = np.asarray(weights, np.float32).copy()
    keep = ~np.any(weights == 0, axis=1)
    uvw, vis, weights = uvw[keep], vis[keep], weights[keep]
    # xweights = 1/(|M|^2 * 1/|w|) with M = I  ->  |w| up to two roundings
    with np.errstate(divide='ignore'):
        weights = (np.float32(1) / (np.float32(1) / np.abs(weights))).astype(np.float32)
    u = uvw[:, 0].copy()
    v = uvw[:, 1].copy()
    w = uvw[:, 2].copy()
    flip = w < 0
    u[flip] = -u[flip]
    v[flip] = -v[flip]
    w[flip] = -w[flip]
    vis[flip] = np.conj(vis[flip])
    vis = (vis * weights).astype(np.complex64)
    bad = ~(np.isfinite(vis.real) & np.isfinite(vis.imag))
    vis[bad] = 0
    weights[bad] = 0
    uv_scale = np.float32(1.0) / np.float32(cell_size)
    w_scale = np.float32((np.float32(w_slices) - np.float32(0.5)) * np.float32(w_planes)
                         / np.float32(max_w))
    u = (u * uv_scale).astype(np.float32)
    v = (v * uv_scale).astype(np.float32)
    wq = np.trunc((w * w_scale + np.float32(w_planes) * np.float32(0.5)).astype(np.float32))
    w_slice_plane = np.minimum(wq.astype(np.int64), w_slices * w_planes - 1)
    pu, su = subpixel_coord(u, oversample)
    pv, sv = subpixel_coord(v, oversample)
    return dict(
        uv=np.stack([pu, pv], axis=1), sub_uv=np.stack([su, sv], axis=1),
        w_plane=(w_slice_plane % w_planes).astype(np.int16),
        w_slice=(w_slice_plane // w_planes).astype(np.int16),
        vis=vis, weights=weights)


def compress(rec):
    """Adjacent-merge compression (preprocess.cpp:334-372): consecutive records
    with identical (uv, sub_uv, w_plane, w_slice) are summed."""
    key = np.concatenate([rec['uv'], rec['sub_uv'], rec['w_plane'][:, None],
                          rec['w_slice'][:, None]], axis=1)
    n = len(key)
    if n == 0:
        return rec
    new = np.ones(n, bool)
    new[1:] = np.any(key[1:] != key[:-1], axis=1)
    group = np.cumsum(new) - 1
    ng = group[-1] + 1
    out = {k: rec[k][new] for k in ('uv', 'sub_uv', 'w_plane', 'w_slice')}
    vis = np.zeros((ng,) + rec['vis'].shape[1:], np.complex64)
    wts = np.zeros((ng,) + rec['weights'].shape[1:], np.float32)
    # sequential float32 accumulation in arrival order, as the C++ loop does
    for i in range(n):
        vis[group[i]] += rec['vis'][i]
        wts[group[i]] += rec['weights'][i]
    out['vis'] = vis
    out['weights'] = wts
    return out


# --------------------------------------------------------------------------
# C restatement loader (oracle/kimg_oracle.c)
# --------------------------------------------------------------------------
_clib = None


def build_c(force=False):
    """Compile oracle/kimg_oracle.c -> oracle/libkimg_oracle.so with gcc."""
    src = os.path.join(_HERE, 'kimg_oracle.c')
    out = os.path.join(_HERE, 'libkimg_oracle.so')
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(
            ['gcc', '-O3', '-ffp-contract=off', '-fno-fast-math', '-march=x86-64-v3',
             '-shared', '-fPIC', '-o', out, src, '-lm'])
    return out


def clib():
    global _clib
    if _clib is None:
        _clib = ctypes.CDLL(build_c())
        _clib.oracle_version.restype = ctypes.c_int
    return _clib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype)
    return a


# --------------------------------------------------------------------------
# preprocess.cpp:390-513 + preprocess.py:73-310   visibility collector
# --------------------------------------------------------------------------
PP_CONFIG_DTYPE = np.dtype([('max_w', 'f4'), ('w_slices', 'i4'), ('w_planes', 'i4'),
                            ('oversample', 'i4'), ('cell_size', 'f4')])


def preprocess_convert(uvw, weights, vis, feed_angle1, feed_angle2, mueller_stokes, mueller_circular,
                       config, num_polarizations):
    """One channel of add_impl2 (preprocess.cpp:435-507) through the C restatement.
    Returns (key int16 [n][6] = u, v, sub_u, sub_v, w_plane, w_slice; weights [n][P]; vis [n][P])."""
    P = num_polarizations
    uvw = _c(uvw, np.float32)
    weights = _c(weights, np.float32)
    vis = _c(vis, np.complex64)
    n, Q = vis.shape
    stokes = _c(np.asarray(mueller_stokes), np.complex64)
    conf = np.zeros(1, PP_CONFIG_DTYPE)
    for k in PP_CONFIG_DTYPE.names:
        conf[k] = config[k]
    key = np.zeros((n, 6), np.int16)
    out_w = np.zeros((n, P), np.float32)
    out_vis = np.zeros((n, P), np.complex64)
    if feed_angle1 is None:
        assert stokes.shape == (P, Q)
        fa1 = fa2 = circ = None
    else:
        fa1 = _c(feed_angle1, np.float32)
        fa2 = _c(feed_angle2, np.float32)
        circ = _c(np.asarray(mueller_circular), np.complex64)
        assert stokes.shape == (P, 4) and circ.shape == (4, Q)
    clib().oracle_pp_convert(
        ctypes.c_int(P), ctypes.c_int(Q), ctypes.c_long(n), _p(uvw), _p(weights), _p(vis),
        _p(fa1) if fa1 is not None else None, _p(fa2) if fa2 is not None else None,
        _p(stokes), _p(circ) if circ is not None else None, _p(conf), _p(key), _p(out_w), _p(out_vis))
    return key, out_w, out_vis


def preprocess_compress(key, weights, vis, w_slices):
    """compress (preprocess.cpp:334-372): returns (key, weights, vis, counts per w_slice), the
    records ordered by w_slice with arrival order kept inside a slice."""
    n, P = weights.shape
    out_key = np.zeros_like(key)
    out_w = np.zeros_like(weights)
    out_vis = np.zeros_like(vis)
    tmp = (np.zeros_like(key), np.zeros_like(weights), np.zeros_like(vis))
    counts = np.zeros(max(w_slices, 1), np.int64)
    lib = clib()
    lib.oracle_pp_compress.restype = ctypes.c_long
    m = lib.oracle_pp_compress(
        ctypes.c_int(P), ctypes.c_long(n), ctypes.c_int(w_slices), _p(key), _p(weights), _p(vis),
        _p(out_key), _p(out_w), _p(out_vis), _p(counts), _p(tmp[0]), _p(tmp[1]), _p(tmp[2]))
    return out_key[:m], out_w[:m], out_vis[:m], counts


class VisibilityCollector:
    """preprocess.py:73-150 + VisibilityCollectorMem (:257-274): per channel and per buffer of
    `buffer_size` inputs, convert -> compress -> append each w_slice run to datasets[channel][slice].

    `configs` is a sequence of dicts/records with the fields of PP_CONFIG_DTYPE."""

    def __init__(self, configs, num_polarizations, buffer_size):
        self.configs = list(configs)
        self.P = num_polarizations
        self.buffer_size = buffer_size
        self.num_input = 0
        self.num_output = 0
        self.datasets = [[[] for _ in range(int(c['w_slices']))] for c in self.configs]

    def add(self, uvw, weights, vis, feed_angle1, feed_angle2, mueller_stokes, mueller_circular):
        N = len(uvw)
        for ch, conf in enumerate(self.configs):
            for i0 in range(0, N, self.buffer_size):
                i1 = min(N, i0 + self.buffer_size)
                fa1 = None if feed_angle1 is None else feed_angle1[i0:i1]
                fa2 = None if feed_angle2 is None else feed_angle2[i0:i1]
                key, w, v = preprocess_convert(uvw[i0:i1], weights[ch, i0:i1], vis[ch, i0:i1], fa1, fa2,
                                               mueller_stokes, mueller_circular, conf, self.P)
                key, w, v, counts = preprocess_compress(key, w, v, int(conf['w_slices']))
                pos = 0
                for s, c in enumerate(counts):
                    if c:
                        sl = slice(pos, pos + c)
                        self.datasets[ch][s].append(dict(
                            uv=key[sl, 0:2].copy(), sub_uv=key[sl, 2:4].copy(),
                            w_plane=key[sl, 4].copy(), weights=w[sl].copy(), vis=v[sl].copy()))
                        pos += c
                self.num_output += pos
        self.num_input += len(self.configs) * N

    def slice_arrays(self, channel, w_slice):
        """All records of one (channel, w_slice) concatenated, as a dict of arrays."""
        parts = self.datasets[channel][w_slice]
        names = ('uv', 'sub_uv', 'w_plane', 'weights', 'vis')
        if not parts:
            return dict(uv=np.zeros((0, 2), np.int16), sub_uv=np.zeros((0, 2), np.int16),
                        w_plane=np.zeros(0, np.int16), weights=np.zeros((0, self.P), np.float32),
                        vis=np.zeros((0, self.P), np.complex64))
        return {k: np.concatenate([p[k] for p in parts]) for k in names}


# --------------------------------------------------------------------------
# grid.py:1032-1052 / 1138-1154   gridding and degridding
# --------------------------------------------------------------------------
def grid_py(kernel, grid, weights_grid, uv, sub_uv, w_plane, vis):
    """Pure-numpy transcription of _grid (grid.py:1032-1052); small cases only."""
    ksize = kernel.shape[2]
    uv_bias = (ksize - 1) // 2 - grid.shape[2] // 2
    ctype = grid.dtype.type
    for row in range(uv.shape[0]):
        u0 = int(uv[row, 0]) - uv_bias
        v0 = int(uv[row, 1]) - uv_bias
        sub_u, sub_v = int(sub_uv[row, 0]), int(sub_uv[row, 1])
        wu = int(uv[row, 0]) + weights_grid.shape[2] // 2
        wv = int(uv[row, 1]) + weights_grid.shape[1] // 2
        wp = int(w_plane[row])
        for pol in range(grid.shape[0]):
            sample = ctype(vis[row, pol] * weights_grid[pol, wv, wu])
            kv = kernel[wp, sub_v, :]
            ku = kernel[wp, sub_u, :]
            weight = np.conj(kv[:, None] * ku[None, :])          # complex64 products
            grid[pol, v0:v0 + ksize, u0:u0 + ksize] += (sample * weight).astype(grid.dtype)


def grid(kernel, grid_, weights_grid, uv, sub_uv, w_plane, vis):
    """_grid (grid.py:1032-1052) via the C restatement.  grid_ is complex64
    [P][G][G] (modified in place) or complex128."""
    lib = clib()
    kernel = _c(kernel, np.complex64)
    weights_grid = _c(weights_grid, np.float32)
    uv = _c(uv, np.int16)
    sub_uv = _c(sub_uv, np.int16)
    w_plane = _c(w_plane, np.int16)
    vis = _c(vis, np.complex64)
    assert grid_.flags.c_contiguous
    P, G = grid_.shape[0], grid_.shape[2]
    assert grid_.shape[1] == G and weights_grid.shape == grid_.shape
    fn = lib.oracle_grid_c64 if grid_.dtype == np.complex64 else lib.oracle_grid_c128
    fn(_p(kernel), ctypes.c_int(kernel.shape[1]), ctypes.c_int(kernel.shape[2]),
       _p(grid_), ctypes.c_int(P), ctypes.c_int(G), _p(weights_grid),
       _p(uv), _p(sub_uv), _p(w_plane), _p(vis), ctypes.c_long(uv.shape[0]))


def degrid_py(kernel, values, uv, sub_uv, w_plane, weights, vis):
    """Pure-numpy transcription of _degrid (grid.py:1138-1154)."""
    ksize = kernel.shape[2]
    uv_bias = (ksize - 1) // 2 - values.shape[2] // 2
    for row in range(uv.shape[0]):
        u0 = int(uv[row, 0]) - uv_bias
        v0 = int(uv[row, 1]) - uv_bias
        wp = int(w_plane[row])
        kv = kernel[wp, int(sub_uv[row, 1]), :]
        ku = kernel[wp, int(sub_uv[row, 0]), :]
        weight = kv[:, None] * ku[None, :]
        for pol in range(values.shape[0]):
            fp = values[pol, v0:v0 + ksize, u0:u0 + ksize]
            sample = values.dtype.type(0)
            for j in range(ksize):
                for k in range(ksize):
                    sample += weight[j, k] * fp[j, k]
            vis[row, pol] -= weights[row, pol] * sample


def degrid(kernel, values, uv, sub_uv, w_plane, weights, vis):
    """_degrid (grid.py:1138-1154) via the C restatement.  vis modified in place."""
    lib = clib()
    kernel = _c(kernel, np.complex64)
    uv = _c(uv, np.int16)
    sub_uv = _c(sub_uv, np.int16)
    w_plane = _c(w_plane, np.int16)
    weights = _c(weights, np.float32)
    assert vis.dtype == np.complex64 and vis.flags.c_contiguous
    assert values.flags.c_contiguous
    P, G = values.shape[0], values.shape[2]
    fn = lib.oracle_degrid_c64 if values.dtype == np.complex64 else lib.oracle_degrid_c128
    fn(_p(kernel), ctypes.c_int(kernel.shape[1]), ctypes.c_int(kernel.shape[2]),
       _p(values), ctypes.c_int(P), ctypes.c_int(G),
       _p(uv), _p(sub_uv), _p(w_plane), _p(weights), _p(vis), ctypes.c_long(uv.shape[0]))


# --------------------------------------------------------------------------
# predict.py:73-149, 419-438
# --------------------------------------------------------------------------
def extract_sky_image(pixels, pixel_size, image_size, oversample, components, dtype=np.float32):
    """_extract_sky_image (predict.py:73-119).  components: {(y,x): array[P]}."""
    N = len(components)
    pols = len(next(iter(components.values()))) if N else 0
    lmn = np.empty((N, 3), np.float32)
    flux = np.empty((N, pols), dtype)
    x = np.array([pos[1] for pos in components])
    y = np.array([pos[0] for pos in components])
    l = (x - 0.5 * pixels) * pixel_size
    m = (y - 0.5 * pixels) * pixel_size
    n1 = np.sqrt(1.0 - (np.square(l) + np.square(m))) - 1.0
    lmn[:, 0] = l
    lmn[:, 1] = m
    lmn[:, 2] = n1
    if N:
        flux[:] = list(components.values())
    taper_scale = float(image_size * oversample)
    tap = np.sinc(l / taper_scale) * np.sinc(m / taper_scale)
    flux *= tap[:, np.newaxis]
    return lmn, flux


def uvw_scale_bias(cell_size, wavelength, max_w, w_slices, w_planes, oversample):
    """_uvw_scale_bias (predict.py:122-149)."""
    uv_scale = float((cell_size / oversample) / wavelength)
    w_scale = float((max_w / ((w_slices - 0.5) * w_planes)) / wavelength)
    w_bias = (0.5 - 0.5 * w_planes) * w_scale
    return uv_scale, w_scale, w_bias


def predict(vis, uv, sub_uv, w_plane, weights, lmn, flux, oversample, uv_scale, w_scale, w_bias):
    """_predict_host (predict.py:419-438) via the C restatement; vis in place."""
    lib = clib()
    assert vis.dtype == np.complex64 and vis.flags.c_contiguous
    uv = _c(uv, np.int16)
    sub_uv = _c(sub_uv, np.int16)
    w_plane = _c(w_plane, np.int16)
    weights = _c(weights, np.float32)
    lmn = _c(lmn, np.float32)
    flux = _c(flux, np.float32)
    lib.oracle_predict(_p(vis), _p(uv), _p(sub_uv), _p(w_plane), _p(weights), _p(lmn), _p(flux),
                       ctypes.c_long(vis.shape[0]), ctypes.c_int(lmn.shape[0]),
                       ctypes.c_int(vis.shape[1]),
                       ctypes.c_float(oversample), ctypes.c_float(uv_scale),
                       ctypes.c_float(w_scale), ctypes.c_float(w_bias))


# --------------------------------------------------------------------------
# image.py:743-848   grid <-> image
# --------------------------------------------------------------------------
def grid_to_image(grid_, image, kernel1d, lm_scale, lm_bias, w):
    """GridToImageHost.__call__ (image.py:781-799): image += ...; returns layer."""
    layer = np.fft.ifft2(np.fft.ifftshift(grid_, axes=(1, 2)), axes=(1, 2)).astype(grid_.dtype)
    scale = layer.shape[1] * layer.shape[2]
    lm = np.arange(image.shape[1]).astype(image.dtype) * lm_scale + lm_bias
    lm = np.fft.ifftshift(lm)
    lm2 = lm * lm
    n = np.sqrt(1 - (lm2[:, np.newaxis] + lm2[np.newaxis, :]))
    w_correct = expj2pi(w * (n - 1))
    layer *= w_correct
    img = layer.real.copy()
    img *= scale
    img *= n[np.newaxis, ...]
    img = np.fft.fftshift(img, axes=(1, 2))
    img /= np.outer(kernel1d, kernel1d)[np.newaxis, ...]
    image += img
    return layer


def image_to_grid(image, kernel1d, lm_scale, lm_bias, w, complex_dtype=np.complex64):
    """ImageToGridHost.__call__ (image.py:836-848): returns (grid, layer)."""
    lm = np.arange(image.shape[1]).astype(image.dtype) * lm_scale + lm_bias
    lm2 = lm * lm
    n = np.sqrt(1 - (lm2[:, np.newaxis] + lm2[np.newaxis, :]))[np.newaxis, ...]
    w_correct = expj2pi(-w * (n - 1))
    kernel = np.outer(kernel1d, kernel1d)[np.newaxis, ...]
    layer = (image / (kernel * n) * w_correct).astype(complex_dtype)
    grid_ = np.fft.fftshift(
        np.fft.fft2(np.fft.ifftshift(layer, axes=(1, 2)), axes=(1, 2)), axes=(1, 2))
    return grid_.astype(complex_dtype), layer


# --------------------------------------------------------------------------
# weight.py:541-605
# --------------------------------------------------------------------------
def weights_grid_add(weights_grid, uv, weights):
    """WeightsHost.grid (weight.py:567-572).  NB the reference biases `uv` in
    place; this restatement leaves the caller's array untouched."""
    shape = weights_grid.shape
    uu = uv[:, 0].astype(np.int64) + shape[2] // 2
    vv = uv[:, 1].astype(np.int64) + shape[1] // 2
    for i in range(len(uv)):
        weights_grid[:, vv[i], uu[i]] += weights[i, :]


def weights_finalize(weight_type, weights_grid, robustness=0.0):
    """WeightsHost.finalize (weight.py:574-605): in place; returns (rms, normalized_rms)."""
    if weight_type == NATURAL:
        weights_grid.fill(1)
        return None, 1.0
    if weight_type == UNIFORM:
        sum_w = np.sum(weights_grid[0])
        sum_dw = np.count_nonzero(weights_grid[0])
        weights_grid[weights_grid == 0] = np.inf
        np.reciprocal(weights_grid, out=weights_grid)
        sum_d2w = np.sum(weights_grid[0])
        rms = np.sqrt(sum_d2w) / sum_dw
        return rms, rms * np.sqrt(sum_w)
    if weight_type == ROBUST:
        sum_sq = np.dot(weights_grid[0].flat, weights_grid[0].flat)
        sum_ = np.sum(weights_grid[0])
        mean_weight = sum_sq / sum_
        S2 = (5 * 10 ** (-robustness)) ** 2 / mean_weight
        old0 = weights_grid[0].copy()
        weights_grid[weights_grid == 0] = np.inf
        np.reciprocal(weights_grid * S2 + 1, out=weights_grid)
        sum_w = np.sum(old0)
        sum_dw = np.sum(weights_grid[0] * old0)
        sum_d2w = np.sum(weights_grid[0] ** 2 * old0)
        rms = np.sqrt(sum_d2w) / sum_dw
        return rms, rms * np.sqrt(sum_w)
    raise ValueError('Unknown weight_type {}'.format(weight_type))


# --------------------------------------------------------------------------
# clean.py:894-1075
# --------------------------------------------------------------------------
def psf_patch(psf, threshold, limit=None):
    """psf_patch_host (clean.py:894-935)."""
    if limit is not None:
        hlimit = (round(limit * min(psf.shape[1], psf.shape[2])) - 1) // 2
        mid_x = psf.shape[2] // 2
        mid_y = psf.shape[1] // 2
        min_x = max(0, mid_x - hlimit)
        min_y = max(0, mid_y - hlimit)
        max_x = min(psf.shape[2] - 1, mid_x + hlimit)
        max_y = min(psf.shape[1] - 1, mid_y + hlimit)
        psf = psf[:, min_y:max_y + 1, min_x:max_x + 1]
    nz = np.nonzero(np.abs(psf) >= threshold)
    if len(nz[0]) == 0:
        return (psf.shape[0], 1, 1)
    y_dist = np.max(np.abs(nz[1] - psf.shape[1] // 2))
    x_dist = np.max(np.abs(nz[2] - psf.shape[2] // 2))
    return (psf.shape[0], int(min(psf.shape[1], 2 * y_dist + 1)),
            int(min(psf.shape[2], 2 * x_dist + 1)))


def noise_est(image, border):
    """noise_est_host (clean.py:938-943)."""
    bp = round(border * min(image.shape[1], image.shape[2]))
    image = image[:, bp:-bp, bp:-bp]
    return np.median(np.abs(image)) * MEDIAN_TO_RMS


def metric_to_power(mode, metric):
    """clean.py:166-174."""
    return metric if mode == CLEAN_I else math.sqrt(metric)


def power_to_metric(mode, power):
    """clean.py:177-184."""
    return power if mode == CLEAN_I else power * power


def noise_threshold_scale(mode, threshold, num_polarizations):
    """clean.py:187-203."""
    if mode == CLEAN_I:
        return threshold
    import scipy.stats
    p = 2 * scipy.stats.norm.sf(threshold)
    return np.sqrt(scipy.stats.chi2.isf(p, num_polarizations))


class Clean:
    """CleanHost (clean.py:971-1075): tiled Hogbom CLEAN on float32 arrays.

    The tile scan (_tile_peak, clean.py:946-968) runs in the C restatement
    (first strict maximum in row-major order; all-zero tile keeps the
    reference's (x0, y0) initial position quirk, clean.py:950)."""

    def __init__(self, pixels, border, loop_gain, mode, image, psf, model):
        self.loop_gain = loop_gain
        self.mode = mode
        self.image = image
        self.psf = psf
        self.model = model
        self.tile_size = 32
        self.border_pixels = round(pixels * border)
        tiles_x = -(-(image.shape[2] - 2 * self.border_pixels) // self.tile_size)
        tiles_y = -(-(image.shape[1] - 2 * self.border_pixels) // self.tile_size)
        self._tile_max = np.zeros((tiles_y, tiles_x), image.dtype)
        self._tile_pos = np.empty((tiles_y, tiles_x, 2), np.int32)
        assert image.dtype == np.float32 and image.flags.c_contiguous

    def _update_tiles(self, ty0, tx0, ty1, tx1):
        clib().oracle_update_tiles(
            _p(self.image), ctypes.c_int(self.image.shape[0]), ctypes.c_int(self.image.shape[1]),
            ctypes.c_int(self.image.shape[2]), ctypes.c_int(self.border_pixels),
            ctypes.c_int(self.tile_size), ctypes.c_int(self.mode),
            _p(self._tile_max), _p(self._tile_pos), ctypes.c_int(self._tile_max.shape[1]),
            ctypes.c_int(ty0), ctypes.c_int(tx0), ctypes.c_int(ty1), ctypes.c_int(tx1))

    def reset(self):
        self._update_tiles(0, 0, self._tile_max.shape[0], self._tile_max.shape[1])

    def _subtract_psf(self, y, x, psf_patch_):
        """clean.py:1014-1048."""
        px, py = psf_patch_[2], psf_patch_[1]
        sx, sy = self.image.shape[2], self.image.shape[1]
        psf_x = self.psf.shape[2] // 2
        psf_y = self.psf.shape[1] // 2
        x0 = x - px // 2
        x1 = x0 + px
        y0 = y - py // 2
        y1 = y0 + py
        psf_x0 = psf_x - px // 2
        psf_y0 = psf_y - py // 2
        psf_x1 = psf_x0 + px
        psf_y1 = psf_y0 + py
        if x0 < 0:
            psf_x0 -= x0
            x0 = 0
        if y0 < 0:
            psf_y0 -= y0
            y0 = 0
        if x1 > sx:
            psf_x1 -= (x1 - sx)
            x1 = sx
        if y1 > sy:
            psf_y1 -= (y1 - sy)
            y1 = sy
        scale = (np.float32(self.loop_gain) * self.image[:, y, x]).astype(self.image.dtype)
        self.image[..., y0:y1, x0:x1] -= (
            scale[:, np.newaxis, np.newaxis] * self.psf[..., psf_y0:psf_y1, psf_x0:psf_x1])
        self.model[..., y, x] += scale
        return (y0, x0, y1, x1), scale

    def __call__(self, psf_patch_, threshold=0.0):
        """clean.py:1060-1075.

        Reference quirk, reproduced: ``peak_pos`` is a *view* of ``_tile_pos`` (clean.py:1063),
        so the position RETURNED at :1075 is read after ``_update_tile`` has rewritten that
        tile -- it is the tile's new best pixel, not the pixel that was subtracted.  The
        subtraction itself (and the model image) use the true peak, kept here in
        ``self.last_pos``.  The reference's GPU path returns the true peak
        (clean.py:881-891); the HIP path follows that."""
        peak_tile = np.unravel_index(np.argmax(self._tile_max), self._tile_max.shape)
        peak_pos = self._tile_pos[peak_tile]
        peak_value = self._tile_max[peak_tile]
        if peak_value < threshold:
            return None, None, None
        self.last_pos = (int(peak_pos[0]), int(peak_pos[1]))
        (y0, x0, y1, x1), model_pixel = self._subtract_psf(int(peak_pos[0]), int(peak_pos[1]),
                                                           psf_patch_)
        ts, bp = self.tile_size, self.border_pixels
        ty0 = max((y0 - bp) // ts, 0)
        tx0 = max((x0 - bp) // ts, 0)
        ty1 = min(-(-(y1 - bp) // ts), self._tile_max.shape[0])
        tx1 = min(-(-(x1 - bp) // ts), self._tile_max.shape[1])
        self._update_tiles(ty0, tx0, ty1, tx1)
        return peak_value, tuple(int(v) for v in peak_pos), model_pixel


# --------------------------------------------------------------------------
# beam.py:158-201   restoring-beam convolution
# --------------------------------------------------------------------------
def beam_covariance_sqrt(x_stddev, y_stddev, theta):
    """beam.py:158-168: M = R diag(sx, sy) R^T."""
    c, s = np.cos(theta), np.sin(theta)
    Q = np.array([[c, -s], [s, c]])
    return Q @ np.diag([x_stddev, y_stddev]) @ Q.T


def convolve_beam(model, amplitude, x_stddev, y_stddev, theta):
    """beam.py:172-201: multiply the FFT of every polarization plane by the analytic transform
    2 pi A |det M| exp(-2 pi^2 |M k|^2) of the Gaussian beam and transform back (wraps)."""
    model = np.asarray(model)
    out = np.empty_like(model)
    model_ft = np.fft.fftn(model, axes=[1, 2])
    M = beam_covariance_sqrt(x_stddev, y_stddev, theta)
    amp = 2 * np.pi * amplitude * np.abs(np.linalg.det(M))
    u = np.fft.fftfreq(model.shape[1])
    v = np.fft.fftfreq(model.shape[2])
    coords = np.stack(np.meshgrid(u, v, indexing='ij'), axis=-1)
    rotated = np.inner(coords, M)
    beam_ft = amp * np.exp(-2.0 * np.pi ** 2 * np.sum(rotated ** 2, axis=-1))
    out[:] = np.fft.ifftn(model_ft * beam_ft[np.newaxis, ...], axes=[1, 2]).real
    return out


# --------------------------------------------------------------------------
# frontend.py:171-209   output statistics of the restore step
# --------------------------------------------------------------------------
def find_peak(image, pbeam, noise):
    """frontend.find_peak (frontend.py:171-194), vectorised: the loop keeps the running maximum of
    |v| over the pixels with |v| * pbeam > 7.5 noise, i.e. the maximum over that set."""
    v = np.abs(image)
    with np.errstate(invalid='ignore'):
        ok = v * pbeam[np.newaxis] > 7.5 * noise
    if not np.any(ok):
        return np.nan
    return v[ok].max()


def get_totals(image, beam_major, beam_minor):
    """frontend.get_totals (frontend.py:197-209) without the Stokes names."""
    sums = np.nansum(image, axis=(1, 2), dtype=np.float64)
    beam_area = 2 * math.pi * beam_major * beam_minor / (8 * math.log(2))
    return sums / beam_area


# --------------------------------------------------------------------------
# io.py:18-270 -- FITS output.  Parity status of THIS section: UNPINNED by a reference file
# (astropy is absent, so the reference's writer cannot be run here and it has no test of its
# own); the functions below restate, card by card, what io.py hands to astropy.io.fits, and
# the test compares the product's header cards and raw data block with them.
# --------------------------------------------------------------------------
# io.py:21-34 (IQUV 1..4, RR LL RL LR -1..-4, and X / Y swapped relative to the IEEE
# enumeration: YY XX YX XY = -5..-8); keys are the reference's polarization constants
# (polarization.py:14-25: I Q U V = 1..4, RR RL LR LL = 5..8, XX XY YX YY = 9..12)
FITS_POLARIZATIONS = {1: 1, 2: 2, 3: 3, 4: 4, 5: -1, 8: -2, 6: -3, 7: -4, 12: -5, 9: -6, 11: -7,
                      10: -8}


def fits_polarization_cards(axis, polarizations):
    """io.py:38-85: ([(key, value)], permutation) of the STOKES axis."""
    codes = np.array([FITS_POLARIZATIONS[p] for p in polarizations])
    permute = np.argsort(codes) if codes[0] >= 0 else np.argsort(-codes)     # :66-70
    codes = codes[permute]
    ref = codes[0]
    delta = codes[1] - codes[0] if len(codes) > 1 else 1                     # :73-76
    if np.any(codes != np.arange(len(codes)) * delta + ref):                 # :77-78
        raise ValueError('Polarizations do not form a linear sequence in FITS enumeration')
    a = str(axis)
    return [('CTYPE' + a, 'STOKES'), ('CRPIX' + a, 1.0), ('CRVAL' + a, float(ref)),
            ('CDELT' + a, float(delta))], permute


def fits_image(image, pixel_size, wavelength, polarizations, phase_centre_rad, beam=None,
               bunit='Jy/beam', extra_fits_headers=None):
    """io.py:126-204: the header cards in insertion order (DATE left out: it is the wall
    clock) and the array handed to the HDU.  ``beam`` = (major, minor, theta) in pixels /
    radians.  ORIGIN / HISTORY carry the reference's package name (io.py:129-130)."""
    cards = []
    if bunit is not None:
        cards.append(('BUNIT', bunit))                                       # :127-128
    cards += [('ORIGIN', 'katsdpimager'), ('HISTORY', 'Created by katsdpimager'),
              ('TIMESYS', 'UTC')]
    cards += [('CRPIX1', image.shape[2] * 0.5), ('CRPIX2', image.shape[1] * 0.5 + 1.0),
              ('CRPIX4', 1.0)]                                               # :140-142
    delt = float(np.degrees(np.arcsin(pixel_size)))                          # :144
    cards += [('CDELT1', -delt), ('CDELT2', delt), ('CDELT4', 1.0)]
    cards += [('EQUINOX', 2000.0), ('RADESYS', 'FK5'), ('CUNIT1', 'deg'), ('CUNIT2', 'deg'),
              ('CUNIT4', 'Hz'), ('CTYPE1', 'RA---SIN'), ('CTYPE2', 'DEC--SIN'), ('CTYPE4', 'FREQ'),
              ('CRVAL1', float(np.degrees(phase_centre_rad[0]))),
              ('CRVAL2', float(np.degrees(phase_centre_rad[1]))),
              ('CRVAL4', 299792458.0 / wavelength)]                          # :153-165
    if beam is not None:                                                     # :166-171
        cards += [('BMAJ', float(np.degrees(beam[0] * pixel_size))),
                  ('BMIN', float(np.degrees(beam[1] * pixel_size))),
                  ('BPA', float(np.degrees(beam[2])))]
    cards += fits_polarization_cards(3, polarizations)[0]                    # :172
    datamin = float(np.fmin.reduce(image, axis=None))                        # :176-180
    datamax = float(np.fmax.reduce(image, axis=None))
    if not np.isnan(datamin):
        cards += [('DATAMIN', datamin), ('DATAMAX', datamax)]
    if extra_fits_headers:                                                   # :182-184 (dict update)
        d = dict(cards)
        d.update(extra_fits_headers)
        keys = [k for k, _ in cards] + [k for k in extra_fits_headers if k not in dict(cards)]
        cards = [(k, d[k]) for k in keys]
    return cards, image[np.newaxis, :, :, ::-1]                              # :191


def fits_grid(grid_, cell_size, polarizations, real_dtype=np.float32):
    """io.py:246-270: cards and the [complex][polarization][v][u] array of write_fits_grid."""
    parts = np.ascontiguousarray(grid_).view(real_dtype).reshape(grid_.shape + (2,))   # :246
    parts = parts.transpose(3, 0, 1, 2)                                                 # :247
    cards = [('BUNIT', 'Jy'), ('ORIGIN', 'katsdpimager'),
             ('CUNIT1', 'm'), ('CRPIX1', parts.shape[3] // 2 + 1.0), ('CRVAL1', 0.0),
             ('CDELT1', float(cell_size)),
             ('CUNIT2', 'm'), ('CRPIX2', parts.shape[2] // 2 + 1.0), ('CRVAL2', 0.0),
             ('CDELT2', float(cell_size))]
    pol_cards, permute = fits_polarization_cards(3, polarizations)
    cards += pol_cards
    cards += [('CTYPE4', 'COMPLEX'), ('CRPIX4', 1.0), ('CRVAL4', 1.0), ('CDELT4', 1.0)]
    return cards, parts[:, permute, :, :]                                    # :269


# --------------------------------------------------------------------------
# Resident-store order (katsdpimager_amd/csrc/store.hip).  NOT a restatement of the reference --
# it has no such step -- but of this package's own definition, kept here so that the device
# kernels are checked against independent numpy code.  What IS pinned to the reference is the
# result of gridding / degridding the re-ordered store: equal (to float rounding) to `grid` /
# `degrid` above applied to the stream in arrival order.
# --------------------------------------------------------------------------
def store_strip_width(kernel_width):
    taps = kernel_width if kernel_width <= 32 else (kernel_width + 1) // 2
    return 32 - taps + 1


def _bits_for(values):
    b = 0
    while (1 << b) < values:
        b += 1
    return b


STORE_MAX_RUN = 4096


def store_reorder(uv4, w_plane, weights, vis, kernel_width, oversample, w_planes, merge):
    """uv4 int16 [N][4] = (u, v, sub_u, sub_v).  Returns (uv4, w_plane, weights, vis) in store
    order; with ``merge`` runs of equal coordinates are summed left to right in float32 (a run
    also ends at every multiple of STORE_MAX_RUN sorted positions: csrc/store.hip)."""
    width = store_strip_width(kernel_width)
    u = uv4[:, 0].astype(np.int64) + 32768
    v = uv4[:, 1].astype(np.int64) + 32768
    strip = u // width
    vv = np.where(strip & 1, 65535 - v, v)
    key = (strip << 16) | vv
    if merge:
        sub_bits, wp_bits, in_bits = _bits_for(oversample), _bits_for(w_planes), _bits_for(width)
        key = (key << in_bits) | (u - strip * width)
        key = (key << sub_bits) | (uv4[:, 3].astype(np.int64) & ((1 << sub_bits) - 1))
        key = (key << sub_bits) | (uv4[:, 2].astype(np.int64) & ((1 << sub_bits) - 1))
        key = (key << wp_bits) | (w_plane.astype(np.int64) & ((1 << wp_bits) - 1))
    order = np.argsort(key.astype(np.uint64), kind='stable')
    uv_s, wp_s, w_s, vis_s = uv4[order], w_plane[order], weights[order], vis[order]
    if not merge or len(order) == 0:
        return uv_s, wp_s, w_s, vis_s
    head = np.ones(len(order), bool)
    head[1:] = np.any(uv_s[1:] != uv_s[:-1], axis=1) | (wp_s[1:] != wp_s[:-1])
    head[::STORE_MAX_RUN] = True
    starts = np.flatnonzero(head)
    lengths = np.diff(np.append(starts, len(order)))
    out_w = w_s[starts].copy()
    out_v = vis_s[starts].copy()
    # k-th member of every run that has one, added in turn: left to right within each run
    for k in range(1, int(lengths.max())):
        sel = lengths > k
        idx = starts[sel] + k
        out_w[sel] = out_w[sel] + w_s[idx]
        out_v[sel] = (out_v[sel].real + vis_s[idx].real) + 1j * (out_v[sel].imag + vis_s[idx].imag)
    return uv_s[starts], wp_s[starts], out_w, out_v.astype(np.complex64)
