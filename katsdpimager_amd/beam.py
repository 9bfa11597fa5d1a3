"""Restoring-beam convolution on the device.

Mirror of the device half of ``katsdpimager.beam`` (beam.py:204-398): ``FourierBeamTemplate``
/ ``FourierBeam`` multiply the half-complex transform of an image by the analytic transform of
a Gaussian beam, ``ConvolveBeamTemplate`` / ``ConvolveBeam`` wrap it in an R2C and a C2R rocFFT.
The beam *fit* (beam.py:91-155) is a least-squares problem with three unknowns on a ~100-pixel
patch and stays on the host: :func:`fit_beam` solves it with scipy where the reference uses
astropy's Levenberg-Marquardt fitter; :class:`Beam` carries the fitted Gaussian's parameters.
"""
import ctypes
import math

import numpy as np

from . import accel, types
from ._lib import lib, check


class _Param(float):
    """A float with the ``.value`` attribute of an astropy model parameter."""

    @property
    def value(self):
        return float(self)


class _Model:
    def __init__(self, amplitude, x_stddev, y_stddev, theta):
        self.amplitude = _Param(amplitude)
        self.x_stddev = _Param(x_stddev)
        self.y_stddev = _Param(y_stddev)
        self.theta = _Param(theta)


class Beam:
    """Gaussian synthesised beam (beam.py:49-88).  Built from the parameters of the fitted
    ``Gaussian2D`` (standard deviations in pixels along the two model axes, angle in
    radians); ``major`` / ``minor`` are FWHMs and ``theta`` is normalised as in the reference.
    ``model`` exposes ``amplitude``, ``x_stddev.value``, ``y_stddev.value``, ``theta`` like the
    astropy model the reference keeps."""

    def __init__(self, amplitude, x_stddev, y_stddev, theta):
        self.model = _Model(amplitude, x_stddev, y_stddev, theta)
        scale = math.sqrt(8 * math.log(2))
        self.major = x_stddev * scale
        self.minor = y_stddev * scale
        theta = float(theta)
        if self.major < self.minor:
            self.minor, self.major = self.major, self.minor
            theta += math.pi / 2
        self.theta = theta % math.pi

    def __repr__(self):
        return 'Beam({0.major!r}, {0.minor!r}, {0.theta!r})'.format(self)


def _gaussian2d(x, y, x_stddev, y_stddev, theta):
    """Unit-amplitude elliptical Gaussian centred on the origin, in the parametrisation of the
    model the reference fits (astropy ``Gaussian2D``): ``theta`` rotates the x axis towards y."""
    c, s = math.cos(theta), math.sin(theta)
    xr = (c * x + s * y) / x_stddev
    yr = (-s * x + c * y) / y_stddev
    return np.exp(-0.5 * (xr * xr + yr * yr))


def fit_beam(psf, step=1.0, threshold=0.01, init_threshold=0.5):
    """Fit a 2-D Gaussian (unit amplitude, centred on pixel ``shape // 2``) to a PSF patch and
    return it as a :class:`Beam` (beam.py:91-155).  ``x`` is axis 0 of ``psf``.

    As in the reference the starting point is the second moment about the origin of the samples
    above ``init_threshold``, corrected for the truncation of a Gaussian at that level
    (beam.py:132-145); the fit then uses all samples above ``threshold``.  The minimiser is
    ``scipy.optimize.least_squares`` (Levenberg-Marquardt) instead of astropy's wrapper around
    the same MINPACK routine."""
    import scipy.optimize
    psf = np.asarray(psf, np.float64)
    if psf.ndim != 2:
        raise ValueError('psf must be 2D')

    def samples(level):
        i0, i1 = np.nonzero(psf > level)
        return (psf[i0, i1], (i0 - psf.shape[0] // 2) * float(step),
                (i1 - psf.shape[1] // 2) * float(step))

    value, x, y = samples(init_threshold)
    total = value.sum()
    cov = np.array([[np.sum(value * x * x), np.sum(value * x * y)],
                    [np.sum(value * x * y), np.sum(value * y * y)]]) / total
    r2 = -2.0 * math.log(init_threshold)
    cov /= 1.0 - (1.0 + 0.5 * r2) * math.exp(-0.5 * r2)
    eigenvalues, eigenvectors = np.linalg.eigh(cov)
    start = [math.sqrt(max(eigenvalues[1], 1e-12)), math.sqrt(max(eigenvalues[0], 1e-12)),
             math.atan2(eigenvectors[1, 1], eigenvectors[0, 1])]

    value, x, y = samples(threshold)
    fit = scipy.optimize.least_squares(
        lambda p: _gaussian2d(x, y, p[0], p[1], p[2]) - value, start, method='lm',
        xtol=1e-12, ftol=1e-12, gtol=1e-12)
    return Beam(1.0, abs(float(fit.x[0])), abs(float(fit.x[1])), float(fit.x[2]))


def beam_covariance_sqrt(beam):
    """beam.py:158-168: M = R diag(sigma_x, sigma_y) R^T."""
    model = beam.model
    c, s = math.cos(model.theta), math.sin(model.theta)
    Q = np.array([[c, -s], [s, c]])
    D = np.diag([model.x_stddev.value, model.y_stddev.value])
    return Q @ D @ Q.T


class FourierBeamTemplate:
    """beam.py:204-233."""

    def __init__(self, context, dtype, tuning=None):
        types.require_float32(dtype, 'FourierBeamTemplate')
        self.context = context
        self.dtype = np.dtype(dtype)

    def instantiate(self, *args, **kwargs):
        return FourierBeam(self, *args, **kwargs)


class FourierBeam(accel.Operation):
    """beam.py:236-311.  Slot **data**: complex (height, width // 2 + 1), transformed in place.
    ``beam`` must be set before the operation is run."""

    def __init__(self, template, command_queue, image_shape, allocator=None):
        if len(image_shape) != 2:
            raise ValueError('image_shape must be 2D')
        super().__init__(command_queue, allocator=allocator)
        self.template = template
        self.image_shape = tuple(image_shape)
        self.slots['data'] = accel.IOSlot((image_shape[0], image_shape[1] // 2 + 1), np.complex64)
        self.beam = None

    def coefficients(self):
        """(amplitude, a, b, c) handed to the kernel, beam.py:283-299."""
        M = beam_covariance_sqrt(self.beam)
        amplitude = 2 * np.pi * self.beam.model.amplitude * np.abs(np.linalg.det(M))
        # the inverse transform is not normalised: fold 1/(H W) into the amplitude
        amplitude /= self.image_shape[0] * self.image_shape[1]
        # integer coordinates -> [-1, 1): folded into the matrix
        M = M @ np.diag([1.0 / self.image_shape[0], 1.0 / self.image_shape[1]])
        C = -2 * np.pi ** 2 * M.T @ M
        return float(amplitude), float(C[0, 0]), float(2 * C[0, 1]), float(C[1, 1])

    def _run(self):
        if self.beam is None:
            raise ValueError('Must set beam')
        amplitude, a, b, c = self.coefficients()
        data = self.buffer('data')
        check(lib().kimg_fourier_beam(data.ptr, data.shape[1], data.shape[1], data.shape[0],
                                      amplitude, a, b, c, self.command_queue.handle),
              'kimg_fourier_beam')


class _RfftPlan:
    def __init__(self, shape):
        self.shape = tuple(shape)
        handle = ctypes.c_void_p()
        check(lib().kimg_rfft_plan_create(ctypes.byref(handle), self.shape[0], self.shape[1]),
              'kimg_rfft_plan_create')
        self._handle = handle

    def execute(self, command_queue, image, fourier, inverse):
        check(lib().kimg_rfft_exec(self._handle, image.ptr, fourier.ptr, 1 if inverse else -1,
                                   command_queue.handle), 'kimg_rfft_exec')

    def __del__(self):
        try:
            if self._handle:
                lib().kimg_rfft_plan_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


class ConvolveBeamTemplate:
    """beam.py:314-349: convolution of one polarization plane with the restoring beam."""

    def __init__(self, context, shape, dtype, padded_shape_image=None, padded_shape_fourier=None,
                 tuning=None):
        if len(shape) != 2:
            raise ValueError('wrong number of dimensions')
        types.require_float32(dtype, 'ConvolveBeamTemplate')
        self.context = context
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape)
        tuning = dict(tuning or {})
        #: square images of a size the library's own transforms take (even, at most 8192, no prime
        #: factor above 7) are convolved by ``kimg_convolve_beam`` (three launches) unless
        #: ``tuning={'own_transform': False}``
        self.own_transform = bool(tuning.pop('own_transform', True)) and self.shape[0] == self.shape[1] \
            and bool(lib().kimg_grid_image_real_supported(self.shape[0], self.shape[0]))
        self._fft = None
        self.fourier_beam = FourierBeamTemplate(context, dtype, tuning)

    @property
    def fft(self):
        """The FFT library's real <-> half-complex plans (made when first needed: plan creation
        costs milliseconds)."""
        if self._fft is None:
            self._fft = _RfftPlan(self.shape)
        return self._fft

    def instantiate(self, *args, **kwargs):
        return ConvolveBeam(self, *args, **kwargs)


class ConvolveBeam(accel.Operation):
    """beam.py:351-398.  Slots **image** (height, width) real — input and output — and
    **fourier** (height, width // 2 + 1) complex scratch."""

    def __init__(self, template, command_queue, allocator=None):
        super().__init__(command_queue, allocator=allocator)
        self.template = template
        self._fourier_beam = template.fourier_beam.instantiate(command_queue, template.shape,
                                                               allocator=allocator)
        self.slots['image'] = accel.IOSlot(template.shape, np.float32)
        self.slots['fourier'] = self._fourier_beam.slots['data']

    @property
    def beam(self):
        return self._fourier_beam.beam

    @beam.setter
    def beam(self, value):
        self._fourier_beam.beam = value

    def _run(self):
        if self.beam is None:
            raise ValueError('Must set beam')
        image, fourier = self.buffer('image'), self.buffer('fourier')
        if self.template.own_transform:
            amplitude, a, b, c = self._fourier_beam.coefficients()
            check(lib().kimg_convolve_beam(
                image.ptr, image.shape[1], image.shape[0], amplitude, a, b, c, fourier.ptr,
                fourier.tensor.numel() * fourier.tensor.element_size(), self.command_queue.handle),
                'kimg_convolve_beam')
            return
        self.template.fft.execute(self.command_queue, image, fourier, inverse=False)
        self._fourier_beam()
        self.template.fft.execute(self.command_queue, image, fourier, inverse=True)


def restore(imager, beam, convolve=None):
    """The restore step of frontend.process_channel (frontend.py:623-641): convolve every
    polarization of the model with the restoring beam (in place) and add the residuals, leaving
    the restored image in ``dirty``.  ``convolve`` may carry a :class:`ConvolveBeam` to reuse."""
    queue = imager.command_queue
    model = imager.buffer('model')
    if convolve is None:
        convolve = ConvolveBeamTemplate(queue.context, model.shape[1:], model.dtype).instantiate(queue)
    convolve.beam = beam
    convolve.ensure_all_bound()
    image = convolve.buffer('image')
    for pol in range(model.shape[0]):
        model.copy_region(queue, image, np.s_[pol], ())
        convolve()
        image.copy_region(queue, model, (), np.s_[pol])
    imager.add_model_to_dirty()
    return convolve
