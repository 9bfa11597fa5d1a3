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
