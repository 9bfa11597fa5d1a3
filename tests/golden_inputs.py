"""Seeded inputs and configurations shared by tools/gen_golden.py (which feeds
them to the *reference* host classes) and the parity tests (which feed them to
the oracle and to the HIP path).  Test infrastructure only.

Recipes follow the reference's own unit tests where they exist
(katsdpimager/test/test_grid.py:24-135, test_clean.py:66-73,174-192,
test_image.py:13-45, test_weight.py:10-118, test_predict.py:54-92).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class RandomState(np.random.RandomState):
    """katsdpimager/test/utils.py:8-23 (complex distributions)."""
    def complex_normal(self, loc=0.0j, scale=1.0, size=None):
        return self.normal(np.real(loc), scale, size) + 1j * self.normal(np.imag(loc), scale, size)

    def complex_uniform(self, low=0.0, high=1.0, size=None):
        if not np.iscomplexobj(low):
            low = np.asarray(low) * (1 + 1j)
        if not np.iscomplexobj(high):
            high = np.asarray(high) * (1 + 1j)
        return self.uniform(np.real(low), np.real(high), size) \
            + 1j * self.uniform(np.imag(low), np.imag(high), size)


def middle(array, shape):
    """Central view of `array` with size `shape` (test_grid.py:13-21)."""
    index = []
    for a, s in zip(array.shape, shape):
        assert a >= s and (a - s) % 2 == 0
        pad = (a - s) // 2
        index.append(np.s_[pad:a - pad])
    return array[tuple(index)]


def make_config(pixels, pixel_size, wavelength, P, kernel_width, w_planes, w_slices=1,
                max_w=5.0, oversample=8, antialias_width=7.0, image_oversample=4,
                real_dtype='float32', **extra):
    """Plain-float equivalent of parameters.ImageParameters/GridParameters
    (parameters.py:52-132,208-256): image_size = pixel_size*pixels,
    cell_size = wavelength/image_size."""
    image_size = pixel_size * pixels
    c = dict(pixels=pixels, pixel_size=pixel_size, wavelength=wavelength, P=P,
             image_size=image_size, cell_size=wavelength / image_size,
             kernel_width=kernel_width, w_planes=w_planes, w_slices=w_slices, max_w=max_w,
             oversample=oversample, antialias_width=antialias_width,
             image_oversample=image_oversample, real_dtype=real_dtype,
             complex_dtype='complex64' if real_dtype == 'float32' else 'complex128')
    c.update(extra)
    return c


# ---- G1 ---------------------------------------------------------------------
KERNEL_CONFIGS = {
    # test_grid.py:26-64
    'testgrid': make_config(256, 0.0001, 0.01, 4, 28, 32),
    # C1-like: 1024 px, 8 planes
    'c1': make_config(1024, 2.0e-5, 0.21, 1, 28, 8, w_slices=2, max_w=800.0),
    # narrow kernel, 2 slices
    'k8': make_config(256, 4.0e-5, 0.2, 1, 8, 8, w_slices=2, max_w=60.0),
    # wide kernel (CLI default width 60, frontend.py:325), few planes to stay small
    'k60': make_config(2048, 1.0e-5, 0.21, 1, 60, 4, w_slices=3, max_w=2000.0),
}

# ---- G2/G3 ------------------------------------------------------------------
GRID_CONFIGS = {
    'p4_f32': make_config(128, 0.0002, 0.01, 4, 28, 32, grid_cover=90, n_vis=1000),
    'p1_f64': make_config(128, 0.0002, 0.01, 1, 28, 32, real_dtype='float64',
                          grid_cover=90, n_vis=600),
    'p1_k8': make_config(96, 0.0002, 0.01, 1, 8, 8, grid_cover=80, n_vis=1500),
    # the CLI's default kernel width (frontend.py:325), two polarizations
    'p2_k60': make_config(192, 0.0002, 0.01, 2, 60, 8, grid_cover=110, n_vis=400),
}


def grid_track(c, seed=1, vis_seed=2):
    """Random-walk track with occasional jumps: test_grid.py:66-90, plus the
    visibilities of do_grid (test_grid.py:94-95)."""
    n_vis, grid_cover = c['n_vis'], c['grid_cover']
    oversample, w_planes = c['oversample'], c['w_planes']
    assert grid_cover + c['kernel_width'] < c['pixels']
    rs = np.random.RandomState(seed=seed)
    uv = np.empty((n_vis, 2), dtype=np.int16)
    sub_uv = np.empty((n_vis, 2), dtype=np.int16)
    w_plane = np.empty((n_vis,), dtype=np.int16)
    for i in range(n_vis):
        if i % 73 == 0:
            uv[i, :] = rs.randint(0, grid_cover, (2,))
            sub_uv[i, :] = rs.randint(0, oversample, (2,))
            w_plane[i] = rs.randint(0, w_planes)
        else:
            for j in range(2):
                uv[i, j] = (uv[i - 1, j] + rs.randint(-1, 2)) % grid_cover
                sub_uv[i, j] = (sub_uv[i - 1, j] + rs.randint(-1, 2)) % oversample
            w_plane[i] = (w_plane[i - 1] + rs.randint(-1, 2)) % w_planes
    uv -= grid_cover // 2
    weights_grid = rs.uniform(size=(c['P'], grid_cover, grid_cover)).astype(np.float32)
    rs2 = RandomState(seed=vis_seed)
    vis = rs2.complex_uniform(-1, 1, size=(n_vis, c['P'])).astype(np.complex64)
    return dict(uv=uv, sub_uv=sub_uv, w_plane=w_plane, weights_grid=weights_grid, vis=vis)


def degrid_inputs(c, seed=2):
    """test_grid.py:114-122."""
    rs = RandomState(seed=seed)
    shape = (c['P'], c['pixels'], c['pixels'])
    grid = rs.complex_uniform(-1, 1, size=shape).astype(c['complex_dtype'])
    vis = rs.complex_uniform(-1, 1, size=(c['n_vis'], c['P'])).astype(np.complex64)
    weights = rs.uniform(0.5, 1.5, size=(c['n_vis'], c['P'])).astype(np.float32)
    return dict(grid=grid, vis=vis, weights=weights)


# ---- G4 ---------------------------------------------------------------------
PREDICT_CONFIG = make_config(4096, 0.00001, 0.2, 3, 7, 100, w_slices=10, max_w=5.0, n_vis=301)


def predict_inputs(c, seed=1):
    """Random quantised coordinates as test_predict.py:57-63 with a synthetic
    component list in place of the katpoint catalogue."""
    rs = RandomState(seed=seed)
    n = c['n_vis']
    uv = rs.randint(-2048, 2049, size=(n, 2)).astype(np.int16)
    sub_uv = rs.randint(0, c['oversample'], size=(n, 2)).astype(np.int16)
    w_plane = rs.randint(0, c['w_planes'], size=n).astype(np.int16)
    weights = rs.uniform(size=(n, c['P'])).astype(np.float32)
    vis = rs.complex_normal(size=(n, c['P'])).astype(np.complex64)
    components = {}
    for i in range(23):
        pos = (int(rs.randint(100, 3996)), int(rs.randint(100, 3996)))
        components[pos] = rs.uniform(-1, 2, size=c['P']).astype(np.float32)
    # corners, as test_predict.py:107-113
    components[(0, 4095)] = np.array([4.0, 0.0, 0.0], np.float32)
    components[(2048, 2048)] = np.array([1.0, 2.0, 3.0], np.float32)
    return dict(uv=uv, sub_uv=sub_uv, w_plane=w_plane, weights=weights, vis=vis,
                components=components, w=1.2)


def predict_mild_components():
    """The reference's own predict test (test_predict.py:45-52) puts three sources within 2.5
    arcmin of the phase centre, so that |phase| stays below ~40 turns and float32 carries it to
    ~1e-5: three components at the pixel offsets of those sources from the centre of the 4096^2,
    1e-5 rad/pixel image ((-16, -8), (-31, +10), (-70, -1) pixels in (l, m)), fluxes (I, Q, V) of
    the same order as the catalogue's at 1.5 GHz."""
    return {
        (2048 - 8, 2048 - 16): np.array([14.8, 1.48, 0.0], np.float32),
        (2048 + 10, 2048 - 31): np.array([6.2, 1.24, 1.24], np.float32),
        (2048 - 1, 2048 - 70): np.array([1.9, 0.19, 1.9], np.float32),
    }


# ---- G12: parameter formulas (parameters.py:17-26, :135-183) -------------------------------
W_SLICES_CASES = [
    # pixel_size, pixels, wavelength, max_w, eps_w, kernel_width, antialias_width
    (1e-5, 4096, 0.2, 5.0, 0.001, 7, 7.0),
    (1e-5, 4096, 0.2, 8000.0, 0.001, 28, 7.0),
    (2.6e-5, 4800, 0.21, 1000.0, 0.001, 60, 7.0),
    (2.6e-5, 4800, 0.21, 8000.0, 0.001, 60, 7.0),
    (2.6e-5, 4800, 0.21, 8000.0, 0.01, 64, 7.0),
    (4e-5, 1024, 0.3, 100.0, 0.001, 16, 7.0),
    (4e-5, 8192, 0.05, 7700.0, 0.003, 33, 5.0),
    (1.3e-5, 6144, 0.21, 7700.0, 0.001, 16, 7.0),
    (1.3e-5, 6144, 0.21, 0.5, 0.001, 64, 0.0),
]
IS_SMOOTH_RANGE = 20000


# ---- G5 ---------------------------------------------------------------------
IMAGE_CONFIGS = {
    # off-centre lm_bias and w as test_image.py:17-21
    'offcentre': dict(P=2, size=102, lm_scale=0.1 / 102, lm_bias=-(0.1 / 102) * 102 / 3,
                      ws=[0.0, 12.3], real_dtype='float32', complex_dtype='complex64'),
    # centred as imaging.py:90-91
    'centred': dict(P=1, size=64, lm_scale=0.002, lm_bias=-0.5 * 64 * 0.002,
                    ws=[0.0, 150.7], real_dtype='float32', complex_dtype='complex64'),
}


def image_inputs(c, seed=1):
    rs = RandomState(seed=seed)
    shape = (c['P'], c['size'], c['size'])
    grid = rs.complex_uniform(-1.0, 1.0, shape).astype(c['complex_dtype'])
    kernel1d = rs.uniform(1.0, 2.0, c['size']).astype(c['real_dtype'])
    model = rs.uniform(-1.0, 1.0, shape).astype(c['real_dtype'])
    return dict(grid=grid, kernel1d=kernel1d, model=model, image_shape=shape)


# ---- G6 ---------------------------------------------------------------------
def weights_inputs(seed=3):
    rs = np.random.RandomState(seed)
    shape = (2, 64, 96)
    n = 3000
    # concentrated towards the centre so that cells collect several visibilities
    uv = np.stack([np.clip(np.rint(rs.normal(0, 12, n)), -48, 47),
                   np.clip(np.rint(rs.normal(0, 8, n)), -32, 31)], axis=1).astype(np.int16)
    weights = rs.uniform(0.1, 2.0, size=(n, shape[0])).astype(np.float32)
    return dict(shape=shape, uv=uv, weights=weights, robustness=-0.5)


# ---- G7 ---------------------------------------------------------------------
CLEAN_CONFIGS = {
    'i': dict(pixels=256, P=1, mode=0, loop_gain=0.1, border=0.02, cycles=250, threshold=0.0,
              psf_patch=(1, 65, 65)),
    'sumsq': dict(pixels=192, P=3, mode=1, loop_gain=0.25, border=0.05, cycles=120,
                  threshold=0.0, psf_patch=(3, 47, 33)),
    # big patch -> subtracts are clipped at the image edges (test_clean.py:146-172)
    'clipped': dict(pixels=128, P=1, mode=0, loop_gain=0.2, border=0.0, cycles=60,
                    threshold=0.0, psf_patch=(1, 101, 127)),
    # threshold stop (clean.py:879-880)
    'threshold': dict(pixels=128, P=1, mode=0, loop_gain=0.3, border=0.1, cycles=200,
                      threshold=2.5, psf_patch=(1, 31, 31)),
}


def gaussian(size, std):
    x = np.arange(size) - (size - 1) / 2.0
    return np.exp(-0.5 * (x / std) ** 2)


def clean_inputs(c, seed=1):
    """Noisy point sources + Gaussian PSF with noise (test_clean.py:66-73 recipe)."""
    rs = np.random.RandomState(seed)
    G, P = c['pixels'], c['P']
    g1 = gaussian(G, G / 40.0)
    psf = np.empty((P, G, G), np.float32)
    for p in range(P):
        psf[p] = np.outer(g1, g1) + 0.01 * rs.standard_normal((G, G))
    # peak of the PSF normalised to 1 at (G//2, G//2) as frontend.py:514-529
    psf /= psf[:, G // 2, G // 2][:, None, None]
    dirty = (0.3 * rs.standard_normal((P, G, G))).astype(np.float32)
    for i in range(12):
        y, x = rs.randint(0, G, 2)
        amp = rs.uniform(2.0, 10.0, P) * rs.choice([-1, 1])
        for p in range(P):
            y0, y1 = max(0, y - G // 2), min(G, y + G - G // 2)
            x0, x1 = max(0, x - G // 2), min(G, x + G - G // 2)
            dirty[p, y0:y1, x0:x1] += (amp[p] * psf[p, y0 - y + G // 2:y1 - y + G // 2,
                                                    x0 - x + G // 2:x1 - x + G // 2])
    return dict(dirty=dirty.astype(np.float32), psf=psf.astype(np.float32),
                psf_patch=tuple(c['psf_patch']))


# ---- G8 ---------------------------------------------------------------------
def psf_patch_cases():
    """The five cases of test_clean.py:13-37 on a (4, 206, 304) PSF."""
    def base():
        psf = np.zeros((4, 206, 304), np.float32)
        psf[:, 103, 152] = 1.0
        return psf
    cases = []
    cases.append((base(), 0.01, None))
    p = base(); p[0, 0, 0] = 0.1; cases.append((p, 0.01, None))
    p = base(); p[3, 205, 303] = -0.2; cases.append((p, 0.01, None))
    p = base(); p[1, 0, :152] = np.arange(152); cases.append((p, 50.5, None))
    p = base(); p[0, 0, 0] = 0.4; p[3, 205, 303] = 0.3; p[1, 110, 150] = 0.2
    cases.append((p, 0.01, 50 / 206))
    return cases


def noise_cases():
    """test_clean.py:174-192 recipe (std 3.2 and 0.0), reduced size."""
    out = []
    for std in (3.2, 0.0):
        rs = np.random.RandomState(seed=1)
        shape = (2, 200, 272)
        bp = 23
        border = bp / shape[1]
        dirty = rs.standard_normal(shape).astype(np.float32)
        dirty[:, bp:-bp, bp:-bp] *= std
        dirty.flat[rs.choice(dirty.size, 300, replace=False)] += 1e6
        out.append((dirty, border))
    return out


# ---- G9 ---------------------------------------------------------------------
_E2E = dict(pixels=256, pixel_size=4.0e-4, wavelength=0.2, P=1, kernel_width=16, w_planes=8,
            w_slices=2, max_w=30.0, weight_type=2, robustness=0.0, loop_gain=0.1, mode=0,
            border=0.02, psf_cutoff=0.01, psf_limit=0.5, major=2, minor=25, major_gain=0.85,
            threshold=5.0, n_vis=1800, vis_block=700, longest_baseline=150.0)
E2E_CONFIGS = {
    'degrid': make_config(**dict(_E2E, degrid=True)),
    'predict': make_config(**dict(_E2E, degrid=False)),
    # full Stokes, CLEAN_SUMSQ peak metric (clean.py:28-31), uniform weights
    'stokes': make_config(**dict(_E2E, degrid=True, P=4, mode=1, weight_type=1, n_vis=1200)),
}


def make_records(P, uv, sub_uv, w_plane, weights, vis):
    """Record layout of the preprocessor output (preprocess.cpp:39-52,
    preprocess.py:42-56): uv i2[2], sub_uv i2[2] contiguous, w_plane i2,
    weights f4[P], vis c8[P]."""
    dtype = np.dtype([('uv', 'i2', (2,)), ('sub_uv', 'i2', (2,)), ('w_plane', 'i2'),
                      ('w_slice', 'i2'), ('weights', 'f4', (P,)), ('vis', 'c8', (P,))])
    rec = np.zeros(len(uv), dtype)
    rec['uv'] = uv
    rec['sub_uv'] = sub_uv
    rec['w_plane'] = w_plane
    rec['weights'] = weights
    rec['vis'] = vis
    return rec.view(np.recarray)


def e2e_raw(c, seed=11):
    """Synthetic observation before preprocessing: random-walk uvw tracks (metres, float32
    [N][3]), point-source visibilities (complex128 [N]) and weights (float32 [N][1])."""
    rs = np.random.RandomState(seed)
    n = c['n_vis']
    ntracks = 60
    per = n // ntracks
    L = c['longest_baseline']
    uvw = []
    for t in range(ntracks):
        r = 0.9 * L * np.sqrt(rs.uniform())
        th = rs.uniform(0, 2 * np.pi)
        start = np.array([r * np.cos(th), r * np.sin(th), rs.uniform(-0.9, 0.9) * c['max_w']])
        heading = rs.uniform(0, 2 * np.pi)
        step = np.zeros((per, 3))
        step[:, 0] = 0.45 * c['cell_size'] * np.cos(heading) + rs.normal(0, 0.05, per) * c['cell_size']
        step[:, 1] = 0.45 * c['cell_size'] * np.sin(heading) + rs.normal(0, 0.05, per) * c['cell_size']
        step[:, 2] = rs.normal(0, 0.01, per) * c['max_w']
        uvw.append(start + np.cumsum(step, axis=0))
    uvw = np.concatenate(uvw).astype(np.float32)
    r = np.hypot(uvw[:, 0], uvw[:, 1])
    uvw[:, :2] *= np.minimum(1.0, 0.95 * L / np.maximum(r, 1e-9))[:, None]
    uvw[:, 2] = np.clip(uvw[:, 2], -0.95 * c['max_w'], 0.95 * c['max_w'])
    # three point sources
    src_lm = np.array([[20, -33], [-41, 12], [5, 60]]) * c['pixel_size']
    src_flux = np.array([1.0, 0.6, 0.35])
    uvw_wl = uvw.astype(np.float64) / c['wavelength']
    vis = np.zeros(len(uvw), np.complex128)
    for (l, m), f in zip(src_lm, src_flux):
        nn = np.sqrt(1 - l * l - m * m)
        vis += f / nn * np.exp(-2j * np.pi * (uvw_wl[:, 0] * l + uvw_wl[:, 1] * m
                                              + uvw_wl[:, 2] * (nn - 1)))
    vis += 0.02 * (rs.standard_normal(len(uvw)) + 1j * rs.standard_normal(len(uvw)))
    weights = rs.uniform(0.5, 1.5, (len(uvw), 1)).astype(np.float32)
    P = c['P']
    if P > 1:
        # further polarizations: scaled copies of the first with their own noise and weights
        # (drawn after everything above, so that the single-polarization inputs are unchanged)
        pol_scale = np.array([1.0, 0.25, -0.15, 0.05])[:P]
        extra = 0.02 * (rs.standard_normal((len(uvw), P)) + 1j * rs.standard_normal((len(uvw), P)))
        extra[:, 0] = 0
        vis = vis[:, None] * pol_scale[None, :] + extra
        weights = np.concatenate(
            [weights, rs.uniform(0.5, 1.5, (len(uvw), P - 1)).astype(np.float32)], axis=1)
    return uvw, vis, weights


def e2e_inputs(c, seed=11):
    """:func:`e2e_raw` quantised and compressed with the restated preprocessor rules."""
    from oracle import kimg_oracle as orc
    uvw, vis, weights = e2e_raw(c, seed)
    vis = vis[:, None] if vis.ndim == 1 else vis
    rec = orc.quantise_uvw(uvw, vis.astype(np.complex64), weights, c['cell_size'],
                           c['max_w'], c['w_slices'], c['w_planes'], c['oversample'])
    rec = orc.compress(rec)
    slices = []
    for s in range(c['w_slices']):
        sel = rec['w_slice'] == s
        slices.append(make_records(c['P'], rec['uv'][sel], rec['sub_uv'][sel],
                                   rec['w_plane'][sel], rec['weights'][sel], rec['vis'][sel]))
    return dict(slices=slices)


def run_major_cycle(im, c, data, host=False, conv=None):
    """Drive an Imaging-shaped object through the per-channel loop of
    frontend.process_channel (frontend.py:465-585; make_weights :86-106,
    make_dirty :110-142).  Works on the reference's ImagingHost and on
    katsdpimager_amd.imaging.Imaging alike; returns arrays for comparison."""
    if conv is None:
        from oracle import kimg_oracle as conv      # same formulas as katsdpimager.clean
    G = c['pixels']
    slices = data['slices']
    vb = c['vis_block']
    out = {}
    slice_w_step = float(c['max_w'] / c['wavelength'] / (c['w_slices'] - 0.5))
    mid_w = np.arange(c['w_slices']) * slice_w_step

    def chunks(rec):
        for i in range(0, len(rec), vb):
            yield rec[i:i + vb].copy().view(np.recarray)

    def make_dirty(field, full_cycle, capture=None):
        im.clear_dirty()
        if full_cycle and not c['degrid']:
            if true_components is not None:
                im._model_components = dict(true_components)
            im.model_to_predict()
        for s, rec in enumerate(slices):
            if len(rec) == 0:
                continue
            if full_cycle and c['degrid']:
                im.model_to_grid(mid_w[s])
            im.clear_grid()
            for ci, chunk in enumerate(chunks(rec)):
                im.num_vis = len(chunk)
                im.set_coordinates(chunk)
                v = np.ascontiguousarray(chunk[field])
                if field == 'weights':
                    v = v.astype(np.complex64)
                im.set_vis(v)
                if full_cycle:
                    im.set_weights(np.ascontiguousarray(chunk.weights))
                    im.predict(mid_w[s])
                    if capture is not None and s == 0 and ci == 0:
                        capture['residual_vis'] = (v.copy() if host else
                                                   np.array(im.get_buffer('vis')[:len(chunk)]))
                im.grid()
            im.grid_to_image(mid_w[s])

    true_components = None
    if host:
        # The reference's CleanHost returns an aliased position (clean.py:1063,1075: a view of
        # _tile_pos that _update_tile rewrites), so ImagingHost._model_components can be keyed
        # by the wrong pixel.  Keep the positions actually subtracted and hand those to
        # model_to_predict, which is what the reference's GPU path does (clean.py:881-891).
        true_components = {}
        inner = im._clean._subtract_psf

        def recording_subtract(y, x, psf_patch):
            res = inner(y, x, psf_patch)
            key = (int(y), int(x))
            true_components[key] = true_components.get(key, 0) + res[1]
            return res
        im._clean._subtract_psf = recording_subtract

    im.clear_model()
    im.clear_weights()
    if c['weight_type'] != 0:
        for rec in slices:
            for chunk in chunks(rec):
                im.grid_weights(np.ascontiguousarray(chunk.uv), np.ascontiguousarray(chunk.weights))
    rms, nrms = im.finalize_weights()
    out['weights_rms'] = np.float64(np.nan if rms is None else rms)
    out['weights_nrms'] = np.float64(nrms)
    out['weights_grid'] = np.array(im.get_buffer('weights_grid'))

    make_dirty('weights', False)
    dirty = np.array(im.get_buffer('dirty'))
    psf_peak = dirty[..., G // 2, G // 2]
    out['psf_peak'] = psf_peak.copy()
    scale = np.reciprocal(psf_peak)
    im.scale_dirty(scale)
    im.dirty_to_psf()
    psf_patch = im.psf_patch()
    out['psf_patch'] = np.array(psf_patch, np.int64)
    out['psf_core'] = np.array(im.get_buffer('psf'))[:, G // 2 - 32:G // 2 + 32,
                                                    G // 2 - 32:G // 2 + 32]
    comps_pos, comps_val, n_minor = [], [], []
    for major in range(c['major']):
        cap = {} if major == 1 else None
        make_dirty('vis', major != 0, cap)
        if cap:
            out.update(cap)
        im.scale_dirty(scale)
        out['dirty%d' % major] = np.array(im.get_buffer('dirty'))
        noise = im.noise_est()
        out['noise%d' % major] = np.float64(noise)
        im.clean_reset()
        peak_value = im.clean_cycle(psf_patch)
        vals = [peak_value]
        # frontend.py:566-576 with the conversions of clean.py:166-203
        peak_power = conv.metric_to_power(c['mode'], peak_value)
        noise_threshold = noise * conv.noise_threshold_scale(c['mode'], c['threshold'], c['P'])
        mgain_threshold = (1.0 - c['major_gain']) * peak_power
        threshold = max(noise_threshold, mgain_threshold)
        if peak_power > threshold:
            threshold = conv.power_to_metric(c['mode'], threshold)
            for j in range(c['minor'] - 1):
                value = im.clean_cycle(psf_patch, threshold)
                if value is None:
                    break
                vals.append(value)
        comps_val.append(np.array(vals, np.float32))
        n_minor.append(len(vals))
    out['n_minor'] = np.array(n_minor, np.int64)
    out['peak_values'] = np.concatenate(comps_val)
    comps = true_components if true_components is not None else im._model_components
    keys = sorted(comps.keys())
    out['component_pos'] = np.array(keys, np.int64).reshape(-1, 2)
    out['component_flux'] = np.array([np.asarray(comps[k]) for k in keys], np.float32)
    out['dirty_final'] = np.array(im.get_buffer('dirty'))
    out['model_final'] = np.array(im.get_buffer('model'))
    return out


# --------------------------------------------------------------------------
# G10: restoring-beam convolution (beam.py:172-201)
# --------------------------------------------------------------------------
def beam_cases():
    """(name, model float32 [P][H][W], beam dict) -- the first is test_beam.py:34-43."""
    cases = []
    model = np.zeros((4, 128, 128), np.float32)
    model[0, 32, 80] = 1.0
    model[0, 100, 40] = 2.0
    model[1, 50, 60] = 3.0
    model[2, 64, 64] = 4.0
    model[2, 80, 64] = 3.0
    cases.append(('testbeam', model, dict(amplitude=3.5, x_stddev=2.0, y_stddev=5.0, theta=1.0)))
    rs = np.random.RandomState(21)
    model = np.zeros((2, 96, 160), np.float32)
    for _ in range(40):
        model[rs.randint(2), rs.randint(96), rs.randint(160)] += rs.uniform(-1, 2)
    cases.append(('rect', model, dict(amplitude=1.0, x_stddev=3.0, y_stddev=1.5, theta=-0.4)))
    model = rs.standard_normal((1, 256, 256)).astype(np.float32)
    cases.append(('dense', model, dict(amplitude=0.7, x_stddev=1.2, y_stddev=4.1, theta=2.5)))
    return cases


# --------------------------------------------------------------------------
# G11: Mueller matrices between polarization bases (polarization.py:69-132)
# --------------------------------------------------------------------------
def polarization_cases():
    """(outputs, inputs) in the CASA enumeration: 1-4 IQUV, 5-8 RR RL LR LL, 9-12 XX XY YX YY.
    Includes combinations that have no solution and redundant inputs."""
    linear, circular = [9, 10, 11, 12], [5, 6, 7, 8]
    inputs = [linear, circular, [9, 12], [5, 8], [12, 11, 10, 9], [1, 2, 3, 4], [9, 10, 11, 12, 5],
              [10, 11]]
    outputs = [[1], [1, 2], [1, 2, 3, 4], [1, 4], [2, 3], [4], circular, linear]
    return [(o, i) for o in outputs for i in inputs]
