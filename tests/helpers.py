"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np


def relerr(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def make_params(c, longest_baseline=None, degrid=None):
    """golden_inputs config dict -> katsdpimager_amd parameter objects."""
    from katsdpimager_amd import parameters
    fixed_i = parameters.FixedImageParameters(list(range(c['P'])), np.float32)
    ip = parameters.ImageParameters(fixed_i, q_fov=1.0, image_oversample=None,
                                    wavelength=c['wavelength'], array=None,
                                    pixel_size=c['pixel_size'], pixels=c['pixels'])
    fixed_g = parameters.FixedGridParameters(
        c['antialias_width'], c['oversample'], c['image_oversample'], c['max_w'],
        c['kernel_width'], degrid=c.get('degrid', False) if degrid is None else degrid)
    gp = parameters.GridParameters(fixed_g, c['w_slices'], c['w_planes'])
    if longest_baseline is None:
        if 'grid_cover' in c:
            longest_baseline = ip.cell_size * (c['grid_cover'] // 2)
        else:
            longest_baseline = c.get('longest_baseline', 0.0)
    ap = parameters.ArrayParameters(13.5, longest_baseline)
    return ip, gp, ap


_ctx = None


def context_queue():
    global _ctx
    from katsdpimager_amd import accel
    if _ctx is None:
        ctx = accel.create_some_context()
        _ctx = (ctx, ctx.create_command_queue())
    return _ctx


def tapered_relerr(a, b, taper1d):
    """Norm-wise error of two dirty images compared BEFORE the division by the image-plane
    taper: a*T vs b*T with T = outer(taper, taper).  The taper (grid.py:404-423) falls to
    ~3e-3 of its peak at the image edge (1e-5 in the corners), so dividing by it amplifies
    the float32 rounding noise of *any* FFT there by up to 1e5; two correct float32
    implementations (pocketfft vs rocFFT) therefore differ by O(1e-3) of the image maximum in
    the corners while agreeing to 1e-6 where the image is meaningful.  The weighted form is
    the quantity the FFT actually computes."""
    t2 = np.outer(taper1d, taper1d).astype(np.float64)
    return relerr(np.asarray(a, np.float64) * t2, np.asarray(b, np.float64) * t2)


def kernel_taper(c):
    from oracle import kimg_oracle as orc
    return orc.taper(c['pixels'], c['antialias_width'], orc.kernel_beta(c['antialias_width']),
                     c['oversample'])


def grid_to_image_truth(grid, kernel1d, lm_scale, lm_bias, w):
    """float64 evaluation of GridToImageHost (image.py:781-799) for a [P][G][G] grid: the
    oracle's formulas, which are dtype-generic, run on complex128 / float64 copies."""
    from oracle import kimg_oracle as orc
    grid = np.asarray(grid, np.complex128)
    image = np.zeros(grid.shape, np.float64)
    orc.grid_to_image(grid, image, np.asarray(kernel1d, np.float64), float(lm_scale), float(lm_bias),
                      float(w))
    return image


def grid_truth_numpy(kernel, uv, sub_uv, w_plane, vis, weights_grid):
    """float64 evaluation of the gridding sum (grid.py:1032-1052) with numpy (small inputs)."""
    kern = np.asarray(kernel, np.complex128)
    P, G = weights_grid.shape[0], weights_grid.shape[-1]
    K = kern.shape[-1]
    half = G // 2
    bias = (K - 1) // 2 - half
    out = np.zeros((P, G * G), np.complex128)
    u = uv[:, 0].astype(np.int64)
    v = uv[:, 1].astype(np.int64)
    taps = np.arange(K)
    kv = np.conj(kern[w_plane, sub_uv[:, 1]])
    ku = np.conj(kern[w_plane, sub_uv[:, 0]])
    idx = ((v - bias)[:, None, None] + taps[None, :, None]) * G + ((u - bias)[:, None, None] + taps[None, None, :])
    for p in range(P):
        smp = vis[:, p].astype(np.complex128) * weights_grid[p][v + half, u + half].astype(np.float64)
        vals = smp[:, None, None] * kv[:, :, None] * ku[:, None, :]
        np.add.at(out[p], idx.reshape(-1), vals.reshape(-1))
    return out.reshape(P, G, G)


def taper_zones(taper1d, threshold=1e-2):
    """(well-conditioned zone, the rest) of an image divided by outer(taper, taper): where the
    product is at least `threshold` of its peak the division amplifies a float32 FFT's rounding by
    at most 1 / threshold."""
    t2 = np.outer(taper1d, taper1d).astype(np.float64)
    good = t2 >= threshold * t2.max()
    return good, ~good, t2
