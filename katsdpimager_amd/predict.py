"""Direct (DFT) prediction of model visibilities and subtraction from the data.

Operator surface of the reference's ``katsdpimager.predict`` (PredictTemplate /
Predict; predict.py:152-416) on libkimg.so.  Sky models are supplied as plain
arrays: ``set_sky_image`` takes CLEAN components (as the reference does), and
``set_sky_arrays`` takes ready-made lmn / flux arrays in place of the
reference's katpoint-catalogue ``set_sky_model`` (catalogue parsing is out of
scope for the hot path).
"""
import numpy as np

from . import accel, grid
from ._lib import lib, check


def extract_sky_image(image_parameters, grid_parameters, components):
    """CLEAN components {(y, x): flux[P]} -> (lmn float32 [N][3] with n-1, flux [N][P])
    with the sub-cell quantisation taper removed (predict.py:73-119)."""
    pols = len(image_parameters.fixed.polarizations)
    n = len(components)
    pos = np.array(list(components.keys()), np.float64).reshape(n, 2)
    pix = float(image_parameters.pixel_size)
    l = (pos[:, 1] - 0.5 * image_parameters.pixels) * pix
    m = (pos[:, 0] - 0.5 * image_parameters.pixels) * pix
    lmn = np.empty((n, 3), np.float32)
    lmn[:, 0] = l
    lmn[:, 1] = m
    lmn[:, 2] = np.sqrt(1.0 - (l * l + m * m)) - 1.0
    flux = np.empty((n, pols), image_parameters.fixed.real_dtype)
    if n:
        flux[:] = list(components.values())
    scale = float(image_parameters.image_size * grid_parameters.fixed.oversample)
    flux *= (np.sinc(l / scale) * np.sinc(m / scale))[:, np.newaxis]
    return lmn, flux


def extract_sky_model(image_parameters, grid_parameters, model, phase_centre):
    """Sky model -> (lmn float32 [N][3] with n-1, flux float32 [N][P]) (predict.py:30-70).

    ``model`` is any object with the two methods of the reference's ``sky_model.SkyModel``:
    ``lmn(phase_centre)`` -> [N][3] direction cosines and ``flux_density(wavelength)`` -> [N][4]
    Stokes IQUV in Jy.  The sub-cell quantisation taper is removed as for CLEAN components and
    the image's polarizations are picked from IQUV."""
    from . import polarization
    lmn = np.array(model.lmn(phase_centre), np.float64).reshape(-1, 3)
    lmn[:, 2] -= 1.0
    flux = np.array(model.flux_density(image_parameters.wavelength), np.float64).reshape(-1, 4)
    scale = float(image_parameters.image_size * grid_parameters.fixed.oversample)
    flux = flux * (np.sinc(lmn[:, 0] / scale) * np.sinc(lmn[:, 1] / scale))[:, np.newaxis]
    index = [polarization.STOKES_IQUV.index(pol) for pol in image_parameters.fixed.polarizations]
    return lmn.astype(np.float32), flux[:, index].astype(np.float32)


def uvw_scale_bias(image_parameters, grid_parameters):
    """Factors turning quantised (cell, sub-cell, plane) indices back into wavelengths
    (predict.py:122-149): uv = uv_scale*(oversample*g + s + 0.5), w = w0 + w_scale*p + w_bias."""
    ip, gp = image_parameters, grid_parameters
    uv_scale = float(ip.cell_size / gp.fixed.oversample / ip.wavelength)
    w_scale = float(gp.fixed.max_w / ((gp.w_slices - 0.5) * gp.w_planes) / ip.wavelength)
    w_bias = (0.5 - 0.5 * gp.w_planes) * w_scale
    return uv_scale, w_scale, w_bias


class PredictTemplate:
    """predict.py:152-287 (no autotuning: one thread per visibility, 256-wide groups)."""

    def __init__(self, context, real_dtype, num_polarizations, tuning=None):
        lib()
        self.context = context
        self.real_dtype = np.dtype(real_dtype)
        self.num_polarizations = num_polarizations
        self.wgs = 256

    def instantiate(self, *args, **kwargs):
        return Predict(self, *args, **kwargs)


class Predict(grid.VisOperation):
    """Instantiation of :class:`PredictTemplate` (predict.py:289-416).

    Slots beyond :class:`~.grid.VisOperation`: **lmn** float32 [sources][3],
    **flux** float32 [sources][pols], **weights** float32 [max_vis][pols].
    """

    def __init__(self, template, command_queue, image_parameters, grid_parameters,
                 max_vis, max_sources, allocator=None):
        if len(image_parameters.fixed.polarizations) != template.num_polarizations:
            raise ValueError('Mismatch in number of polarizations')
        super().__init__(command_queue, template.num_polarizations, max_vis, allocator)
        self.template = template
        pol_dim = accel.Dimension(template.num_polarizations, exact=True)
        sources_dim = max(1, max_sources)
        self.slots['lmn'] = accel.IOSlot((sources_dim, accel.Dimension(3, exact=True)), np.float32)
        self.slots['flux'] = accel.IOSlot((sources_dim, pol_dim), np.float32)
        self.slots['weights'] = accel.IOSlot((max_vis, pol_dim), np.float32)
        self._num_sources = 0
        self.max_sources = max_sources
        self.image_parameters = image_parameters
        self.grid_parameters = grid_parameters
        self._w = 0.0

    def set_sky_arrays(self, lmn, flux):
        """Upload a sky model given as arrays (l, m, n-1) and per-polarization flux."""
        n = len(lmn)
        if n > self.max_sources:
            raise ValueError('too many sources ({} > {})'.format(n, self.max_sources))
        self.ensure_all_bound()
        self._num_sources = n
        if n:
            self.buffer('lmn').set_region(self.command_queue, np.asarray(lmn, np.float32),
                                          np.s_[:n], np.s_[:])
            self.buffer('flux').set_region(self.command_queue, np.asarray(flux, np.float32),
                                           np.s_[:n], np.s_[:])

    def set_sky_model(self, model, phase_centre):
        """predict.py:332-348: sources of a sky model (see :func:`extract_sky_model`)."""
        if len(model) > self.max_sources:
            raise ValueError('too many sources ({} > {})'.format(len(model), self.max_sources))
        lmn, flux = extract_sky_model(self.image_parameters, self.grid_parameters, model,
                                      phase_centre)
        self.set_sky_arrays(lmn, flux)

    def set_sky_image(self, components):
        """predict.py:351-370."""
        lmn, flux = extract_sky_image(self.image_parameters, self.grid_parameters, components)
        if len(lmn) > self.max_sources:
            raise ValueError('too many components ({} > {})'.format(len(lmn), self.max_sources))
        self.set_sky_arrays(lmn, flux)

    @property
    def num_sources(self):
        return self._num_sources

    def set_w(self, w):
        """Centre of the W slice in wavelengths (predict.py:376-384)."""
        self._w = w

    def _run(self):
        if self.num_vis == 0 or self.num_sources == 0:
            return
        uv_scale, w_scale, w_bias = uvw_scale_bias(self.image_parameters, self.grid_parameters)
        w_bias += self._w
        rc = lib().kimg_predict(
            self.buffer('vis').ptr, self.buffer('uv').ptr, self.buffer('w_plane').ptr,
            self.buffer('weights').ptr, self.buffer('lmn').ptr, self.buffer('flux').ptr,
            self.num_vis, self.num_sources, self.template.num_polarizations,
            self.grid_parameters.fixed.oversample, uv_scale, w_scale, w_bias,
            self.command_queue.handle)
        check(rc, 'kimg_predict')
