"""Convolutional gridding / degridding operators on MI355X.

Operator surface of the reference's ``katsdpimager.grid`` (GridderTemplate /
Gridder, DegridderTemplate / Degridder, ConvolutionKernel[Device];
grid.py:344-463, 549-1029), re-implemented on top of libkimg.so.  The device
kernels are not Romein-style thread-per-cell kernels: see
``csrc/grid_mfma.hip`` for the MFMA moving-window design.

The combined anti-aliasing x W-projection kernel table is computed on the
host in float64 exactly as the reference does for both its CPU and GPU paths
(grid.py:235-334: the device classes also build the table with numpy and
upload it).
"""
import collections
import math

import numpy as np

from . import accel, types
from ._lib import lib, check


# --------------------------------------------------------------------------
# Kernel table
# --------------------------------------------------------------------------
def _i0_ratio(x, beta):
    return np.i0(x) / np.i0(beta)


def kaiser_bessel(x, width, beta):
    """Kaiser-Bessel window with support [-width/2, width/2] (grid.py:136-155)."""
    x = np.asarray(x, np.float64)
    t = 1.0 - np.square(2.0 * x / width)
    inside = t >= 0
    out = np.zeros_like(t)
    out[inside] = _i0_ratio(beta * np.sqrt(t[inside]), beta)
    return out


def kaiser_bessel_fourier(f, width, beta):
    """Continuous Fourier transform of :func:`kaiser_bessel` (grid.py:158-184):
    width/I0(beta) * sinc(sqrt((width f)^2 - (beta/pi)^2)), analytically continued."""
    f = np.asarray(f, np.float64)
    q = np.square(width * f) - (beta / math.pi) ** 2
    root = np.sqrt(q.astype(np.complex128))
    return (width / np.i0(beta)) * np.sinc(root).real


def antialias_beta(antialias_width):
    """Shape parameter: first null of the taper just outside the image (grid.py:373-378)."""
    return 1.2 * math.pi * math.sqrt(0.25 * antialias_width ** 2 - 1.0)


def make_kernel_table(cell_wavelengths, ws, width, oversample, antialias_width,
                      image_oversample, beta):
    """Separable AA x W kernel table, complex128 [len(ws)][oversample][width].

    Same construction as the reference's ``antialias_w_kernel`` (grid.py:235-334):
    sample aa(l) * exp(2 pi i (w (l^2/2 + 5 l^4/24) + shift l)) on an
    ``image_oversample``-times finer image grid, DFT to uv space, crop to
    ``oversample*width`` samples and deinterleave the sub-pixel phases (reversed,
    because the sub-pixel index is that of the visibility, not of the tap).
    """
    ws = np.asarray(ws, np.float64)
    n_out = oversample * width
    if n_out % 2:
        raise ValueError('oversample * kernel_width must be even')
    n_img = n_out * image_oversample
    step = 1.0 / (width * cell_wavelengths * image_oversample)
    l = (np.arange(n_img) - n_img // 2) * step
    aa = cell_wavelengths * kaiser_bessel_fourier(l * cell_wavelengths, antialias_width, beta)
    half_subcell = -0.5 * cell_wavelengths / oversample
    l2 = l * l
    phase = np.outer(-ws, -0.5 * l2 - (5.0 / 24.0) * l2 * l2) + half_subcell * l
    phase -= np.rint(phase)
    img = aa * np.exp(2j * np.pi * phase)
    uv = np.fft.fft(np.fft.ifftshift(img, axes=-1), axis=-1) * step
    uv = np.concatenate([uv[..., -(n_out // 2):], uv[..., :n_out // 2]], axis=-1)
    table = uv.reshape(ws.shape + (width, oversample))[..., ::-1]
    return np.ascontiguousarray(np.swapaxes(table, -1, -2))


def antialias_w_kernel(cell_wavelengths, w, width, oversample, antialias_width,
                       image_oversample, beta, out=None):
    """The reference's name and signature for :func:`make_kernel_table` (grid.py:235-334):
    ``w`` is an array of w values in wavelengths; the result is [len(w)][oversample][width]."""
    table = make_kernel_table(cell_wavelengths, w, width, oversample, antialias_width,
                              image_oversample, beta)
    if out is None:
        return table
    out[:] = table
    return out


def subpixel_coord(x, oversample):
    """(pixel, sub-pixel) index of a real-valued grid coordinate (grid.py:338-341)."""
    fine = int(np.floor(x * oversample))
    return fine // oversample, fine % oversample


class ConvolutionKernel:
    """Separable convolution kernel with metadata (grid.py:344-423)."""

    def __init__(self, image_parameters, grid_parameters, data=None):
        self._set_parameters(image_parameters, grid_parameters)
        fixed = grid_parameters.fixed
        shape = (grid_parameters.w_planes, fixed.oversample, fixed.kernel_width)
        self.data = np.empty(shape, np.complex64) if data is None else data
        self.data[:] = make_kernel_table(self.cell_wavelengths, self.ws, fixed.kernel_width,
                                         fixed.oversample, fixed.antialias_width,
                                         fixed.image_oversample, self.beta)

    def _set_parameters(self, image_parameters, grid_parameters):
        """beta and the w of every plane (grid.py:374-383)."""
        self.grid_parameters = grid_parameters
        fixed = grid_parameters.fixed
        self.cell_wavelengths = float(image_parameters.cell_size / image_parameters.wavelength)
        slice_wl = float(fixed.max_w / (grid_parameters.w_slices * image_parameters.wavelength))
        plane_wl = slice_wl / grid_parameters.w_planes
        self.beta = antialias_beta(fixed.antialias_width)
        w_edge = 0.5 * (slice_wl - plane_wl)
        self.ws = np.linspace(-w_edge, w_edge, grid_parameters.w_planes)

    def taper(self, N, out=None):
        """Image-plane correction for an N-pixel image (grid.py:404-423)."""
        x = np.arange(N) / N - 0.5
        fixed = self.grid_parameters.fixed
        values = kaiser_bessel_fourier(x, fixed.antialias_width, self.beta)
        values = values * np.sinc(x / fixed.oversample)
        if out is None:
            return values
        out[:] = values
        return out


class ConvolutionKernelDevice(ConvolutionKernel):
    """:class:`ConvolutionKernel` whose table is generated on the device
    (``kimg_kernel_table``) and stays there (grid.py:426-463).  ``data``, the host copy the
    reference keeps, is downloaded on first use.  The table is stored unpadded: the HIP kernels
    pad rows to the window width when staging to LDS.

    Tables too large for the device generator (``oversample * kernel_width * image_oversample``
    > 5120, i.e. kernel widths beyond 160) are built on the host exactly as the reference
    builds all of its tables, and uploaded."""

    def __init__(self, context, image_parameters, grid_parameters, pad=0, allocator=None):
        self._set_parameters(image_parameters, grid_parameters)
        fixed = grid_parameters.fixed
        if allocator is None:
            allocator = accel.DeviceAllocator(context)
        shape = (grid_parameters.w_planes, fixed.oversample, fixed.kernel_width)
        if (fixed.oversample * fixed.kernel_width) % 2:
            raise ValueError('oversample * kernel_width must be even')
        image_oversample = int(fixed.image_oversample)
        if image_oversample != fixed.image_oversample:
            raise ValueError('image_oversample must be an integer')
        self.padded_data = allocator.allocate(shape, np.complex64)
        self._queue = queue = context.create_command_queue()
        self._data = None
        if fixed.oversample * fixed.kernel_width * image_oversample <= 5120:
            ws = allocator.allocate((len(self.ws),), np.float64)
            ws.set(queue, self.ws)
            check(lib().kimg_kernel_table(
                self.padded_data.ptr, ws.ptr, shape[0], shape[2], shape[1], image_oversample,
                self.cell_wavelengths, float(fixed.antialias_width), self.beta, queue.handle),
                'kimg_kernel_table')
        else:
            self._data = make_kernel_table(
                self.cell_wavelengths, self.ws, fixed.kernel_width, fixed.oversample,
                fixed.antialias_width, image_oversample, self.beta).astype(np.complex64)
            self.padded_data.set(queue, self._data)
        queue.finish()
        self.pad = 0

    @property
    def data(self):
        if self._data is None:
            self._data = self.padded_data.get(self._queue)
        return self._data

    @property
    def bin_size(self):
        return self.padded_data.shape[-1]


_kernel_cache = collections.OrderedDict()
_KERNEL_CACHE_SIZE = 8


def shared_kernel_device(context, image_parameters, grid_parameters, pad=0):
    """The (read-only) device kernel table for these parameters.  A channel's gridder and
    degridder use the same table, and generating it on the host costs several milliseconds —
    comparable with imaging a whole channel — so the most recent few are shared."""
    fixed = grid_parameters.fixed
    key = (id(context), float(image_parameters.cell_size), float(image_parameters.wavelength),
           float(fixed.max_w), grid_parameters.w_slices, grid_parameters.w_planes, fixed.oversample,
           float(fixed.image_oversample), fixed.kernel_width, float(fixed.antialias_width), pad)
    kernel = _kernel_cache.get(key)
    if kernel is None:
        kernel = ConvolutionKernelDevice(context, image_parameters, grid_parameters, pad)
        _kernel_cache[key] = kernel
        while len(_kernel_cache) > _KERNEL_CACHE_SIZE:
            _kernel_cache.popitem(last=False)
    else:
        _kernel_cache.move_to_end(key)
    return kernel


# --------------------------------------------------------------------------
# Operators
# --------------------------------------------------------------------------
GRID_VARIANTS = {'auto': 0, 'generic': 1, 'mfma': 2, 'binned': 3}           # KIMG_VARIANT_*
#: Arithmetic of the matrix instructions (KIMG_ARITH_*): ``fp32`` = v_mfma_f32_32x32x2_f32, every
#: product and sum in float32 like the reference (grid.py:1049-1052), the default; ``split_fp16`` =
#: operands as fp16 hi/lo pairs with float32 accumulation (faster, 22-bit operands; opt-in).
GRID_ARITH = {'fp32': 0, 'split_fp16': 1}


#: `auto` variant: calls smaller than this go straight to the window kernel ...
AUTO_MIN_VIS = 65536
#: ... larger ones are binned when this fraction of their records would force a window flush (a
#: flush costs about what 25 visibilities cost; sorting costs about what gridding twice costs)
AUTO_JUMP_FRACTION = 0.05


def _tuning(tuning):
    tuning = tuning or {}
    unknown = set(tuning) - {'variant', 'arith'}
    if unknown:
        raise ValueError('unknown tuning keys: {}'.format(sorted(unknown)))
    try:
        return GRID_VARIANTS[tuning.get('variant', 'auto')], GRID_ARITH[tuning.get('arith', 'fp32')]
    except KeyError as e:
        raise ValueError('unknown tuning value {}'.format(e)) from None


class GridderTemplate:
    """grid.py:549-653.  ``tuning`` may hold ``{'variant': 'auto'|'generic'|'mfma'|'binned',
    'arith': 'fp32'|'split_fp16'}`` (the reference's tuning dict carries its autotuned work-group
    shape; there is no autotuner here -- the kernel geometry is fixed by the MFMA tile shape).
    Both are per template, passed to the C ABI on every call: nothing is read from the environment.

    ``mfma`` is the window kernel on the stream as it comes; ``binned`` first sorts the visibilities
    by grid tile on the device (for streams without locality: time order, shuffled).  ``auto``
    measures the stream (``kimg_grid_jumps``: records that would force a whole-window flush, one
    4-byte read-back per call of at least :data:`AUTO_MIN_VIS` visibilities) and takes ``binned``
    when more than :data:`AUTO_JUMP_FRACTION` of the records jump, unless the caller already knows
    (``Gridder.locality_hint``)."""

    def __init__(self, context, fixed_image_parameters, fixed_grid_parameters, tuning=None):
        types.require_float32(fixed_image_parameters.real_dtype, 'GridderTemplate')
        lib()
        self.context = context
        self.fixed_image_parameters = fixed_image_parameters
        self.fixed_grid_parameters = fixed_grid_parameters
        self.variant, self.arith = _tuning(tuning)
        self.kernel_pad = 0

    def instantiate(self, *args, **kwargs):
        return Gridder(self, *args, **kwargs)


class VisOperation(accel.Operation):
    """Operations that hold visibilities in device buffers (grid.py:656-703).

    Slots: **uv** int16 [max_vis][4] (u, v, sub_u, sub_v); **w_plane** int16 [max_vis];
    **vis** complex64 [max_vis][pols] (pre-multiplied by statistical weights).
    """

    def __init__(self, command_queue, num_polarizations, max_vis, allocator=None):
        super().__init__(command_queue, allocator)
        self.max_vis = max_vis
        self.slots['uv'] = accel.IOSlot((max_vis, accel.Dimension(4, exact=True)), np.int16)
        self.slots['w_plane'] = accel.IOSlot((max_vis,), np.int16)
        self.slots['vis'] = accel.IOSlot(
            (max_vis, accel.Dimension(num_polarizations, exact=True)), np.complex64)
        self._num_vis = 0

    @property
    def num_vis(self):
        return self._num_vis

    @num_vis.setter
    def num_vis(self, n):
        if n < 0 or n > self.max_vis:
            raise ValueError('Number of visibilities {} is out of range 0..{}'.format(
                n, self.max_vis))
        self._num_vis = n


class GridDegrid(VisOperation):
    """Common part of :class:`Gridder` and :class:`Degridder` (grid.py:706-773)."""

    def __init__(self, template, command_queue, array_parameters,
                 image_parameters, grid_parameters, max_vis, allocator=None):
        assert image_parameters.fixed == template.fixed_image_parameters
        assert grid_parameters.fixed == template.fixed_grid_parameters
        num_polarizations = len(image_parameters.fixed.polarizations)
        super().__init__(command_queue, num_polarizations, max_vis, allocator)
        self.convolve_kernel = shared_kernel_device(
            template.context, image_parameters, grid_parameters, template.kernel_pad)
        # Longest baseline must leave the whole footprint inside the grid (grid.py:753-761)
        max_uv_src = float(array_parameters.longest_baseline / image_parameters.cell_size)
        kernel_size = self.convolve_kernel.padded_data.shape[-1]
        grid_pixels = 2 * (int(max_uv_src) + kernel_size // 2 + 1)
        if grid_pixels > image_parameters.pixels:
            raise ValueError('image_oversample is too small '
                             'to capture all visibilities in the UV plane')
        self.template = template
        self.image_parameters = image_parameters
        self.grid_parameters = grid_parameters
        self.slots['grid'] = accel.IOSlot(
            (num_polarizations, grid_pixels, grid_pixels), image_parameters.fixed.complex_dtype)

    def parameters(self):
        return {'grid_parameters': self.grid_parameters,
                'image_parameters': self.image_parameters}

    def _kernel_args(self):
        table = self.convolve_kernel.padded_data
        return (table.ptr, table.shape[0], table.shape[1], table.shape[2])

    # ---- stream order: window kernel as is, or tile-binned first (see GridderTemplate) ----------
    def _init_locality(self, binned_bytes):
        self._binned_bytes = binned_bytes
        self._jumps = None
        #: What the caller knows about the bound visibilities' order, for the `auto` variant: True =
        #: consecutive records stay close (the window kernel as is), False = no locality (bin first),
        #: None = measure on every call.  Reset it when other data is bound.
        self.locality_hint = None
        #: variant the last call took ('mfma', 'binned' or 'generic'), for tests and reports
        self.last_variant = None
        #: CUs (of 256) the window kernel of THIS operation fills with its resident workgroups;
        #: 0 = the process default (all).  It travels with every call (KIMG_WINDOW_CUS): an imager
        #: that shares its GPU with other channels in flight leaves room for their CLEAN launches.
        self.window_cus = 0

    def _binned_workspace(self):
        """The scratch of the binned variant, allocated on first use."""
        if not self._binned_bytes:
            raise ValueError('the binned variant needs the MFMA window kernel (kernel width <= 64)')
        if self._workspace_bytes < self._binned_bytes:
            self._workspace = accel.DeviceArray(self.command_queue.context, (self._binned_bytes,),
                                                np.uint8, queue=self.command_queue)
            self._workspace_bytes = self._binned_bytes

    def _note_variant(self, variant):
        self.last_variant = {1: 'generic', 3: 'binned'}.get(variant, 'mfma' if self._binned_bytes
                                                             else 'generic')

    def jump_fraction(self):
        """Fraction of the bound visibilities that would force a whole-window flush in the window
        kernel (``kimg_grid_jumps``); synchronises with the queue."""
        if self.num_vis < 2:
            return 0.0
        if self._jumps is None:
            self._jumps = accel.DeviceArray(self.command_queue.context, (1,), np.uint32,
                                            queue=self.command_queue)
        check(lib().kimg_grid_jumps(self.buffer('uv').ptr, self.num_vis, self._kernel_args()[3],
                                    self._jumps.ptr, self.command_queue.handle), 'kimg_grid_jumps')
        return int(self._jumps.get(self.command_queue)[0]) / self.num_vis

    def measure_locality(self):
        """Set :attr:`locality_hint` from one measurement of the bound visibilities (what the
        resident store does once per slice); returns the jump fraction."""
        f = self.jump_fraction()
        self.locality_hint = f <= AUTO_JUMP_FRACTION
        return f

    def _choose_variant(self):
        variant = self.template.variant
        if variant != GRID_VARIANTS['auto'] or not self._binned_bytes:
            return variant
        if self.locality_hint is not None:
            return GRID_VARIANTS['mfma' if self.locality_hint else 'binned']
        if self.num_vis < AUTO_MIN_VIS:
            return variant
        return GRID_VARIANTS['binned' if self.jump_fraction() > AUTO_JUMP_FRACTION else 'mfma']



class Gridder(GridDegrid):
    """Instantiation of :class:`GridderTemplate` (grid.py:776-867).

    Extra slot **weights_grid** float32 [pols][G][G]: density weights looked up per
    visibility.  ``__call__`` adds ``num_vis`` visibilities to **grid**.
    """

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.slots['weights_grid'] = accel.IOSlot(self.slots['grid'].shape, np.float32)
        num_pols = self.slots['grid'].shape[0]
        table = self.convolve_kernel.padded_data
        nbytes = lib().kimg_grid_workspace_bytes(self.max_vis, num_pols, table.shape[0],
                                                 table.shape[1], table.shape[2])
        self._workspace = None
        self._workspace_bytes = nbytes
        if nbytes:
            self._workspace = accel.DeviceArray(self.command_queue.context, (nbytes,), np.uint8,
                                                queue=self.command_queue)
        binned = lib().kimg_grid_binned_workspace_bytes(
            self.max_vis, num_pols, table.shape[0], table.shape[1], table.shape[2])
        # the window gridder packs first-tap coordinates into 16 bits and addresses a polarization
        # plane with 32-bit byte offsets (kimg_grid: grid_size <= 32000, grid_size * row_stride * 8
        # < 2^32); wider grids go to the generic kernel whatever the stream's order, so `auto`
        # must not ask for the binned variant there
        Gg = self.slots['grid'].shape[1]
        if Gg > 32000 or Gg * Gg * 8 >= 1 << 32:
            binned = 0
        self._init_locality(binned)

    def _run(self):
        grid = self.buffer('grid')
        wg = self.buffer('weights_grid')
        P, G = grid.shape[0], grid.shape[1]
        table, W, OV, K = self._kernel_args()
        variant = self._choose_variant()
        if variant == GRID_VARIANTS['binned']:
            self._binned_workspace()        # 34 + 8 P bytes per visibility of max_vis
        rc = lib().kimg_grid(
            grid.ptr, G, G * G, G, P,
            wg.ptr, G, G * G,
            self.buffer('uv').ptr, self.buffer('w_plane').ptr, self.buffer('vis').ptr,
            self.num_vis, table, W, OV, K,
            self._workspace.ptr if self._workspace is not None else None,
            self._workspace_bytes, variant | int(self.window_cus) << 8, self.template.arith,
            self.command_queue.handle)
        check(rc, 'kimg_grid')
        self._note_variant(variant)


class DegridderTemplate:
    """grid.py:870-970.  ``tuning`` as for :class:`GridderTemplate` (the binned variant also sorts
    the statistical weights and scatters the residual visibilities back into the caller's order)."""

    def __init__(self, context, fixed_image_parameters, fixed_grid_parameters, tuning=None):
        types.require_float32(fixed_image_parameters.real_dtype, 'DegridderTemplate')
        lib()
        self.context = context
        self.fixed_image_parameters = fixed_image_parameters
        self.fixed_grid_parameters = fixed_grid_parameters
        self.variant, self.arith = _tuning(tuning)
        self.kernel_pad = 0

    def instantiate(self, *args, **kwargs):
        return Degridder(self, *args, **kwargs)


class Degridder(GridDegrid):
    """Instantiation of :class:`DegridderTemplate` (grid.py:973-1029).

    Extra slot **weights** float32 [max_vis][pols] (statistical weights).  ``__call__``
    subtracts the visibilities predicted from **grid** from **vis** in place.
    """

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        num_pols = self.slots['grid'].shape[0]
        self.slots['weights'] = accel.IOSlot(
            (self.max_vis, accel.Dimension(num_pols, exact=True)), np.float32)
        table = self.convolve_kernel.padded_data
        nbytes = lib().kimg_degrid_workspace_bytes(num_pols, table.shape[0], table.shape[1],
                                                   table.shape[2])
        self._workspace = None
        self._workspace_bytes = nbytes
        if nbytes:
            self._workspace = accel.DeviceArray(self.command_queue.context, (nbytes,), np.uint8,
                                                queue=self.command_queue)
        self._init_locality(lib().kimg_degrid_binned_workspace_bytes(
            self.max_vis, num_pols, table.shape[0], table.shape[1], table.shape[2]))

    def _run(self):
        grid = self.buffer('grid')
        P, G = grid.shape[0], grid.shape[1]
        table, W, OV, K = self._kernel_args()
        variant = self._choose_variant()
        if variant == GRID_VARIANTS['binned']:
            self._binned_workspace()        # 38 + 12 P bytes per visibility of max_vis
        rc = lib().kimg_degrid(
            grid.ptr, G, G * G, G, P,
            self.buffer('uv').ptr, self.buffer('w_plane').ptr, self.buffer('weights').ptr,
            self.buffer('vis').ptr, self.num_vis, table, W, OV, K,
            self._workspace.ptr if self._workspace is not None else None, self._workspace_bytes,
            variant | int(self.window_cus) << 8, self.template.arith, self.command_queue.handle)
        check(rc, 'kimg_degrid')
        self._note_variant(variant)
