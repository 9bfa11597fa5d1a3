"""Image-domain operators and grid<->image conversion on MI355X.

Operator surface of the reference's ``katsdpimager.image`` (LayerToImage,
ImageToLayer, Scale, AddImage, ApplyPrimaryBeam, GridImageTemplate,
GridToImage, ImageToGrid; image.py:15-740) on libkimg.so + rocFFT.
"""
import ctypes
import threading

import numpy as np

from . import accel, types
from ._lib import lib, check


def _pol_ptr(array, pol):
    """Device address of polarization `pol` of a [P][H][W] array."""
    return array.ptr + pol * array.shape[1] * array.shape[2] * array.dtype.itemsize


class _LayerImageTemplate:
    """image.py:15-86."""
    def __init__(self, context, real_dtype, tuning=None):
        types.require_float32(real_dtype, type(self).__name__)
        lib()
        self.context = context
        self.real_dtype = np.dtype(real_dtype)


class _LayerImage(accel.Operation):
    """Conversion between a "layer" (raw FFT of one uv plane, DC in the corner) and the
    stacked image (image.py:89-180).  Slots: **layer** complex [G][G], **image** real
    [P][G][G], **kernel1d** real [G]."""

    def __init__(self, template, command_queue, shape, lm_scale, lm_bias, allocator=None):
        if len(shape) != 3 or shape[-1] != shape[-2]:
            raise ValueError('shape must be square, not {}'.format(shape))
        if shape[-1] % 2 != 0:
            raise ValueError('image size must be even, not {}'.format(shape[-1]))
        super().__init__(command_queue, allocator)
        self.template = template
        complex_dtype = types.real_to_complex(template.real_dtype)
        self.slots['layer'] = accel.IOSlot(shape[-2:], complex_dtype)
        self.slots['image'] = accel.IOSlot(shape, template.real_dtype)
        self.slots['kernel1d'] = accel.IOSlot((shape[-1],), template.real_dtype)
        self.lm_scale = lm_scale
        self.lm_bias = lm_bias
        self.w = 0
        self.polarization = 0

    def set_w(self, w):
        self.w = w

    def set_polarization(self, polarization):
        if polarization < 0 or polarization >= self.slots['image'].shape[0]:
            raise IndexError('polarization index out of range')
        self.polarization = polarization


class LayerToImageTemplate(_LayerImageTemplate):
    def instantiate(self, *args, **kwargs):
        return LayerToImage(self, *args, **kwargs)


class LayerToImage(_LayerImage):
    """image += Re(layer e^{2 pi i w (n-1)}) n / taper, with fftshift (image.py:203-229)."""
    def _run(self):
        image = self.buffer('image')
        size = image.shape[-1]
        rc = lib().kimg_layer_to_image(
            _pol_ptr(image, self.polarization), size, self.buffer('layer').ptr, size,
            self.buffer('kernel1d').ptr, self.lm_scale, self.lm_bias, self.w,
            self.command_queue.handle)
        check(rc, 'kimg_layer_to_image')


class ImageToLayerTemplate(_LayerImageTemplate):
    def instantiate(self, *args, **kwargs):
        return ImageToLayer(self, *args, **kwargs)


class ImageToLayer(_LayerImage):
    """layer = image / (taper n) e^{-2 pi i w (n-1)} (image.py:252-278)."""
    def _run(self):
        image = self.buffer('image')
        size = image.shape[-1]
        rc = lib().kimg_image_to_layer(
            self.buffer('layer').ptr, _pol_ptr(image, self.polarization), size, size,
            self.buffer('kernel1d').ptr, self.lm_scale, self.lm_bias, self.w,
            self.command_queue.handle)
        check(rc, 'kimg_image_to_layer')


class _ImageTemplate:
    def __init__(self, context, dtype, num_polarizations, tuning=None):
        types.require_float32(dtype, type(self).__name__)
        lib()
        self.context = context
        self.dtype = np.dtype(dtype)
        self.num_polarizations = num_polarizations


def _check_image_shape(template, shape):
    if len(shape) != 3:
        raise ValueError('Wrong number of dimensions in shape')
    if shape[0] != template.num_polarizations:
        raise ValueError('Mismatch in number of polarizations')


class ScaleTemplate(_ImageTemplate):
    """image.py:281-314."""
    def instantiate(self, *args, **kwargs):
        return Scale(self, *args, **kwargs)


class Scale(accel.Operation):
    """data[p] *= scale_factor[p] (image.py:317-367)."""
    def __init__(self, template, command_queue, shape, allocator=None):
        super().__init__(command_queue, allocator)
        self.template = template
        _check_image_shape(template, shape)
        self.slots['data'] = accel.IOSlot(shape, template.dtype)
        self.scale_factor = np.zeros((shape[0],), template.dtype)

    def set_scale_factor(self, scale_factor):
        self.scale_factor[:] = scale_factor

    def _run(self):
        data = self.buffer('data')
        P, H, W = data.shape
        sf = np.ascontiguousarray(self.scale_factor, np.float32)
        rc = lib().kimg_scale(data.ptr, W, H * W, W, H, P,
                              sf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                              self.command_queue.handle)
        check(rc, 'kimg_scale')


class AddImageTemplate(_ImageTemplate):
    """image.py:370-403."""
    def instantiate(self, *args, **kwargs):
        return AddImage(self, *args, **kwargs)


class AddImage(accel.Operation):
    """dest += src (image.py:406-458)."""
    def __init__(self, template, command_queue, shape, allocator=None):
        super().__init__(command_queue, allocator)
        self.template = template
        _check_image_shape(template, shape)
        self.slots['src'] = accel.IOSlot(shape, template.dtype)
        self.slots['dest'] = accel.IOSlot(shape, template.dtype)

    def _run(self):
        src, dest = self.buffer('src'), self.buffer('dest')
        P, H, W = src.shape
        rc = lib().kimg_add_image(dest.ptr, W, H * W, src.ptr, W, H * W, W, H, P,
                                  self.command_queue.handle)
        check(rc, 'kimg_add_image')


class ApplyPrimaryBeamTemplate(_ImageTemplate):
    """image.py:461-494."""
    def instantiate(self, *args, **kwargs):
        return ApplyPrimaryBeam(self, *args, **kwargs)


class ApplyPrimaryBeam(accel.Operation):
    """data = beam < threshold ? replacement : data / beam (image.py:497-558)."""
    def __init__(self, template, command_queue, shape, threshold, replacement, allocator=None):
        super().__init__(command_queue, allocator)
        self.template = template
        _check_image_shape(template, shape)
        self.slots['data'] = accel.IOSlot(shape, template.dtype)
        self.slots['beam_power'] = accel.IOSlot(shape[1:], template.dtype)
        self.threshold = threshold
        self.replacement = replacement

    def _run(self):
        data = self.buffer('data')
        P, H, W = data.shape
        rc = lib().kimg_apply_primary_beam(
            data.ptr, W, H * W, self.buffer('beam_power').ptr, W, W, H, P,
            self.threshold, self.replacement, self.command_queue.handle)
        check(rc, 'kimg_apply_primary_beam')


class _PlanPool:
    """Idle rocFFT plans by shape.  Creating and destroying a 2-D plan costs ~7 + ~11 ms of host
    time, more than the device work of a small channel, so a plan released by one channel's
    ``Imaging`` is handed to the next one instead of being destroyed.  A plan is owned by one
    :class:`FftPlan` at a time (a plan carries ONE execution info: set-stream + execute on a shared plan is not thread safe)."""
    MAX_IDLE = 4

    def __init__(self):
        self._lock = threading.Lock()
        self._idle = {}

    def take(self, shape):
        with self._lock:
            handles = self._idle.get(shape)
            return handles.pop() if handles else None

    def give(self, shape, handle):
        with self._lock:
            handles = self._idle.setdefault(shape, [])
            if len(handles) < self.MAX_IDLE:
                handles.append(handle)
                return True
        return False


_plan_pool = _PlanPool()


class FftPlan:
    """rocFFT 2-D C2C plan (stands in for katsdpsigproc.fft.FftTemplate, image.py:599)."""
    def __init__(self, shape):
        self.shape = tuple(int(x) for x in shape)
        self.dtype_src = self.dtype_dest = np.dtype(np.complex64)
        handle = _plan_pool.take(self.shape)
        if handle is None:
            handle = ctypes.c_void_p()
            check(lib().kimg_fft_plan_create(ctypes.byref(handle), self.shape[0], self.shape[1]),
                  'kimg_fft_plan_create')
        self._handle = handle
        self._queue = None
        self._real = None

    def execute(self, command_queue, layer, inverse):
        self._queue = command_queue
        check(lib().kimg_fft_exec(self._handle, layer.ptr, 1 if inverse else -1,
                                  command_queue.handle), 'kimg_fft_exec')

    def real_plan(self):
        """The real <-> half-complex plan of the same size for the w = 0 routes of GridToImage /
        ImageToGrid, made on first use and shared by the operators that share this plan (and its
        `layer` buffer)."""
        if self._real is None:
            self._real = RealFftPlan(self.shape)
        return self._real

    def __del__(self):
        try:
            handle, self._handle = self._handle, None
            if handle and self._queue is not None:
                self._queue.finish()      # the next owner may run the plan on another stream
            if handle and not _plan_pool.give(self.shape, handle):
                lib().kimg_fft_plan_destroy(handle)
        except Exception:
            pass


class RealFftPlan:
    """rocFFT 2-D complex-to-real plan (with its real-to-complex twin, unused here) for the w = 0
    layers of :class:`GridToImage`; pooled like :class:`FftPlan`."""
    def __init__(self, shape):
        self.shape = tuple(int(x) for x in shape)
        key = ('real',) + self.shape
        handle = _plan_pool.take(key)
        if handle is None:
            handle = ctypes.c_void_p()
            check(lib().kimg_rfft_plan_create(ctypes.byref(handle), self.shape[0], self.shape[1]),
                  'kimg_rfft_plan_create')
        self._handle = handle
        self._queue = None

    def execute_in_place(self, command_queue, half_layer, inverse=True):
        """Half-complex [H][W/2+1] <-> real rows of W + 2 floats, in the same memory."""
        self._queue = command_queue
        check(lib().kimg_rfft_exec(self._handle, half_layer.ptr, half_layer.ptr, 1 if inverse else -1,
                                   command_queue.handle), 'kimg_rfft_exec')

    def __del__(self):
        try:
            handle, self._handle = self._handle, None
            if handle and self._queue is not None:
                self._queue.finish()
            if handle and not _plan_pool.give(('real',) + self.shape, handle):
                lib().kimg_rfft_plan_destroy(handle)
        except Exception:
            pass


class GridImageTemplate:
    """Grid <-> image through a complex-to-complex transform keeping the real part
    (image.py:561-606).  The grid need not be Hermitian.

    ``tuning={'real_transform': False}`` switches off the complex-to-real route that grid -> image
    takes for w = 0 (see :class:`GridToImage`); ``{'own_transform': False}`` keeps that route on
    the library's 2-D plans where it would otherwise run the two-launch transforms of
    ``kimg_grid_to_image_real`` / ``kimg_image_to_grid_real`` and their any-w counterparts (even
    layer sizes up to 8192 without a prime factor above 7)."""

    def __init__(self, context, real_dtype, tuning=None):
        types.require_float32(real_dtype, 'GridImageTemplate')
        tuning = tuning or {}
        if set(tuning) - {'real_transform', 'own_transform'}:
            raise ValueError('bad GridImageTemplate tuning {}'.format(tuning))
        self.real_transform = bool(tuning.get('real_transform', True))
        self.own_transform = bool(tuning.get('own_transform', True))
        self.context = context
        self.real_dtype = np.dtype(real_dtype)
        self.layer_to_image = LayerToImageTemplate(context, real_dtype)
        self.image_to_layer = ImageToLayerTemplate(context, real_dtype)

    def make_fft_plan(self, shape_layer, padded_shape_layer=None):
        plan = FftPlan(shape_layer)
        if self.real_transform:
            plan.real_plan()        # (plan creation costs milliseconds of host time: not in the loop)
        return plan

    def instantiate_grid_to_image(self, *args, **kwargs):
        return GridToImage(self, *args, **kwargs)

    def instantiate_image_to_grid(self, *args, **kwargs):
        return ImageToGrid(self, *args, **kwargs)


class _GridImage(accel.Operation):
    def __init__(self, template, command_queue, shape_grid, lm_scale, lm_bias, fft_plan,
                 layer_image_template, allocator=None):
        super().__init__(command_queue, allocator)
        self.template = template
        self._fft = fft_plan
        shape_image = (shape_grid[0],) + tuple(fft_plan.shape)
        self._layer_image = layer_image_template.instantiate(
            command_queue, shape_image, lm_scale, lm_bias, allocator)
        for name in ('layer', 'image', 'kernel1d'):
            self.slots[name] = self._layer_image.slots[name]
        self.slots['grid'] = accel.IOSlot(shape_grid, np.complex64)
        if shape_grid[1] != shape_grid[2] or shape_grid[1] % 2:
            raise ValueError('grid must be square with even size')     # image.py:655-656

    def set_w(self, w):
        self._layer_image.set_w(w)

    def _own_transform(self):
        """Whether the w = 0 route runs on the library's own transforms."""
        G = self.buffer('layer').shape[0]
        Gg = self.buffer('grid').shape[1]
        return (self.template.real_transform and self.template.own_transform
                and bool(lib().kimg_grid_image_real_supported(G, Gg)))


class GridToImage(_GridImage):
    """Per polarization: centred grid -> corner-DC layer (zero padded), inverse FFT,
    layer_to_image accumulate (image.py:609-673)."""
    def __init__(self, template, command_queue, shape_grid, lm_scale, lm_bias, fft_plan,
                 allocator=None):
        super().__init__(template, command_queue, shape_grid, lm_scale, lm_bias, fft_plan,
                         template.layer_to_image, allocator)
        self._real_plan = None
        #: set by a caller that knows the image to be all zero (``Imaging.clear_dirty``): the next
        #: call writes the image instead of adding to it, where the route can (:meth:`can_overwrite`)
        self.overwrite_next = False

    def can_overwrite(self):
        """Whether the next call can write the image instead of accumulating into it."""
        return self._own_transform()

    def _run(self):
        grid, layer = self.buffer('grid'), self.buffer('layer')
        P, Gg, _ = grid.shape
        G = layer.shape[0]
        q = self.command_queue
        if self._layer_image.w == 0 and self.template.real_transform:
            # w = 0: the phase factor is 1 and only the real part of the transform is used, which
            # is the transform of the grid's Hermitian part: half the layer, a complex-to-real
            # transform in place (real rows of G + 2 floats), half the traffic all the way
            image = self.buffer('image')
            li = self._layer_image
            if self._own_transform():
                overwrite, self.overwrite_next = self.overwrite_next, False
                for pol in range(P):
                    check(lib().kimg_grid_to_image_real(
                        _pol_ptr(image, pol), G, G, _pol_ptr(grid, pol), Gg, Gg,
                        self.buffer('kernel1d').ptr, li.lm_scale, li.lm_bias,
                        0 if overwrite else 1, layer.ptr,
                        layer.tensor.numel() * layer.tensor.element_size(), q.handle),
                        'kimg_grid_to_image_real')
                return
            self._real_plan = self._fft.real_plan()
            for pol in range(P):
                check(lib().kimg_grid_to_half_layer(layer.ptr, G, _pol_ptr(grid, pol), Gg, Gg,
                                                    q.handle), 'kimg_grid_to_half_layer')
                self._real_plan.execute_in_place(q, layer)
                check(lib().kimg_real_layer_to_image(
                    _pol_ptr(image, pol), G, layer.ptr, G + 2, G, self.buffer('kernel1d').ptr,
                    li.lm_scale, li.lm_bias, q.handle), 'kimg_real_layer_to_image')
            return
        if self._own_transform():
            overwrite, self.overwrite_next = self.overwrite_next, False
            image = self.buffer('image')
            li = self._layer_image
            for pol in range(P):
                check(lib().kimg_grid_to_image_w(
                    _pol_ptr(image, pol), G, G, _pol_ptr(grid, pol), Gg, Gg,
                    self.buffer('kernel1d').ptr, li.lm_scale, li.lm_bias, float(li.w),
                    0 if overwrite else 1, layer.ptr,
                    layer.tensor.numel() * layer.tensor.element_size(), q.handle),
                    'kimg_grid_to_image_w')
            return
        for pol in range(P):
            check(lib().kimg_grid_to_layer(layer.ptr, G, _pol_ptr(grid, pol), Gg, Gg,
                                           q.handle), 'kimg_grid_to_layer')
            self._fft.execute(q, layer, inverse=True)
            self._layer_image.set_polarization(pol)
            self._layer_image()


class ImageToGrid(_GridImage):
    """Per polarization: image_to_layer, forward FFT, layer corners -> centred grid
    (image.py:676-740)."""
    def __init__(self, template, command_queue, shape_grid, lm_scale, lm_bias, fft_plan,
                 allocator=None):
        super().__init__(template, command_queue, shape_grid, lm_scale, lm_bias, fft_plan,
                         template.image_to_layer, allocator)
        self._real_plan = None

    def _run(self):
        grid, layer = self.buffer('grid'), self.buffer('layer')
        P, Gg, _ = grid.shape
        G = layer.shape[0]
        if self._layer_image.w == 0 and self.template.real_transform:
            # w = 0: the layer is real; real-to-complex transform in place, the grid's other half
            # from F(-k) = conj F(k)
            q = self.command_queue
            image = self.buffer('image')
            li = self._layer_image
            if self._own_transform():
                for pol in range(P):
                    check(lib().kimg_image_to_grid_real(
                        _pol_ptr(grid, pol), Gg, Gg, _pol_ptr(image, pol), G, G,
                        self.buffer('kernel1d').ptr, li.lm_scale, li.lm_bias, layer.ptr,
                        layer.tensor.numel() * layer.tensor.element_size(), q.handle),
                        'kimg_image_to_grid_real')
                return
            self._real_plan = self._fft.real_plan()
            for pol in range(P):
                check(lib().kimg_image_to_real_layer(
                    layer.ptr, G + 2, _pol_ptr(image, pol), G, G, self.buffer('kernel1d').ptr,
                    li.lm_scale, li.lm_bias, q.handle), 'kimg_image_to_real_layer')
                self._real_plan.execute_in_place(q, layer, inverse=False)
                check(lib().kimg_half_layer_to_grid(_pol_ptr(grid, pol), Gg, Gg, layer.ptr, G,
                                                    q.handle), 'kimg_half_layer_to_grid')
            return
        if self._own_transform():
            q = self.command_queue
            image = self.buffer('image')
            li = self._layer_image
            for pol in range(P):
                check(lib().kimg_image_to_grid_w(
                    _pol_ptr(grid, pol), Gg, Gg, _pol_ptr(image, pol), G, G,
                    self.buffer('kernel1d').ptr, li.lm_scale, li.lm_bias, float(li.w), layer.ptr,
                    layer.tensor.numel() * layer.tensor.element_size(), q.handle),
                    'kimg_image_to_grid_w')
            return
        for pol in range(P):
            self._layer_image.set_polarization(pol)
            self._layer_image()
            self._fft.execute(self.command_queue, layer, inverse=False)
            check(lib().kimg_layer_to_grid(_pol_ptr(grid, pol), Gg, Gg, layer.ptr, G,
                                           self.command_queue.handle), 'kimg_layer_to_grid')
