"""Plain-float parameter objects with the attribute names of the reference's
``katsdpimager.parameters`` (parameters.py:28-298).

The reference carries astropy Quantities; here every length is a float in
metres and every angle-like quantity is dimensionless (l/m direction cosines),
which is what the operators consume after ``float(...)`` anyway.
"""
import math

import numpy as np

from . import types

# clean.py:28-31
CLEAN_I = 0
CLEAN_SUMSQ = 1


def is_smooth(x):
    """Is ``x`` an image size the FFT likes: a multiple of 8 with no prime factor above 7
    (same answers as parameters.py:17-25 for every positive size)."""
    x = int(x)
    if x <= 0 or x & 7:
        return False
    # divide out everything 2, 3, 5 and 7 contribute, whole bunches at a time
    while True:
        common = math.gcd(x, 2 * 2 * 2 * 3 * 5 * 7)
        if common == 1:
            return x == 1
        x //= common


class ArrayParameters:
    """parameters.py:28-34 (metres)."""
    def __init__(self, antenna_diameter, longest_baseline):
        self.antenna_diameter = float(antenna_diameter)
        self.longest_baseline = float(longest_baseline)


class FixedImageParameters:
    """parameters.py:37-49."""
    def __init__(self, polarizations, dtype):
        self.polarizations = list(polarizations)
        self.real_dtype = np.dtype(dtype)
        self.complex_dtype = types.real_to_complex(dtype)

    def __eq__(self, other):
        return (isinstance(other, FixedImageParameters)
                and self.polarizations == other.polarizations
                and self.real_dtype == other.real_dtype)


class ImageParameters:
    """parameters.py:52-118.  ``wavelength`` in metres; ``pixel_size`` is the l/m step."""
    def __init__(self, fixed, q_fov, image_oversample, wavelength, array,
                 pixel_size=None, pixels=None):
        self.fixed = fixed
        self.wavelength = float(wavelength)
        if pixel_size is None:
            if image_oversample < 3.0:
                raise ValueError('image_oversample is too small '
                                 'to capture all visibilities in the UV plane')
            uv_size = (2.0 / 3.0 * image_oversample) * array.longest_baseline
            self.pixel_size = self.wavelength / uv_size
        else:
            self.pixel_size = float(pixel_size)
        if pixels is None:
            cell_size = array.antenna_diameter * (math.pi / (7.6634 * q_fov))
            image_size = self.wavelength / cell_size
            pixels = int(0.98 * image_size / self.pixel_size)
            while not is_smooth(pixels):
                pixels += 1
        elif not is_smooth(pixels):
            recommended = pixels
            while not is_smooth(recommended):
                recommended += 1
            raise ValueError("Image size {} not supported - try {}".format(pixels, recommended))
        assert pixels % 2 == 0
        self.pixels = int(pixels)
        self.image_size = self.pixel_size * self.pixels
        self.cell_size = self.wavelength / self.image_size


def w_kernel_width(image_parameters, w, eps_w, antialias_width=0):
    """parameters.py:135-158 (w in metres)."""
    fov = image_parameters.image_size
    wl = float(w / image_parameters.wavelength)
    wk2 = 4 * fov**2 * ((wl * image_parameters.image_size / 2)**2
                        + wl**1.5 * fov / (2 * math.pi * eps_w))
    return np.sqrt(wk2 + antialias_width**2)


def w_slices(image_parameters, max_w, eps_w, kernel_width, antialias_width=0):
    """Fewest W slices for which the combined W + anti-aliasing kernel fits ``kernel_width``
    taps (parameters.py:161-183; ``max_w`` in metres).

    A slice is corrected to its centre, so only half of its thickness remains as residual w; the
    first slice is half as thick as the others, hence ``count - 0.5`` slices share ``max_w``.
    The kernel width falls monotonically with the slice count: double the count until the kernel
    fits, then bisect between the last count known to fail and the first known to fit (the
    reference's two comparisons, ``>`` while growing and ``<`` while bisecting, are kept so that
    a kernel of exactly ``kernel_width`` is judged the same way)."""
    residual_w = 0.5 * max_w

    def width_with(count):
        return w_kernel_width(image_parameters, residual_w / (count - 0.5), eps_w, antialias_width)

    fits = 1
    while width_with(fits) > kernel_width:
        fits *= 2
    fails = 0
    while fits - fails > 1:
        probe = (fits + fails) // 2
        if width_with(probe) < kernel_width:
            fits = probe
        else:
            fails = probe
    return fits


class WeightParameters:
    """parameters.py:186-205."""
    def __init__(self, weight_type, robustness=0.0):
        self.weight_type = weight_type
        self.robustness = robustness


class FixedGridParameters:
    """parameters.py:208-238 (max_w in metres)."""
    def __init__(self, antialias_width, oversample, image_oversample,
                 max_w, kernel_width, degrid=False, beams=None):
        self.antialias_width = antialias_width
        self.oversample = int(oversample)
        self.image_oversample = image_oversample
        self.max_w = float(max_w)
        self.kernel_width = int(kernel_width)
        self.degrid = degrid
        self.beams = beams

    def __eq__(self, other):
        return isinstance(other, FixedGridParameters) and self.__dict__ == other.__dict__


class GridParameters:
    """parameters.py:241-271."""
    def __init__(self, fixed, w_slices, w_planes):
        self.fixed = fixed
        self.w_slices = int(w_slices)
        self.w_planes = int(w_planes)


class CleanParameters:
    """parameters.py:274-298."""
    def __init__(self, minor, loop_gain, major_gain, threshold, mode,
                 psf_cutoff, psf_limit, border):
        self.minor = minor
        self.loop_gain = loop_gain
        self.major_gain = major_gain
        self.threshold = threshold
        self.mode = mode
        self.psf_cutoff = psf_cutoff
        if self.psf_cutoff >= 1.0:
            raise ValueError('PSF cutoff must be less than 1')
        self.psf_limit = psf_limit
        self.border = border
