"""Top-level imaging facade: the drop-in boundary of the hot path.

Same method surface as the reference's ``katsdpimager.imaging.Imaging`` /
``ImagingHost`` (imaging.py:81-588), so that the per-channel driver
(``frontend.process_channel``, frontend.py:465-658, and ``make_weights`` /
``make_dirty``, frontend.py:86-142) runs on it unchanged.  Everything executes
on one HIP stream; all buffers live in HBM for the whole channel.

Additions over the reference surface (all optional for a caller):
``clean_cycles`` runs a batch of minor cycles on the device without host
round trips, ``set_sky_arrays`` feeds the continuum predictor from arrays, and
``streams=2`` alternates consecutive chunks over two HIP streams (see
:class:`_SidePipeline`).
"""
import functools

import numpy as np

from . import accel, clean, grid, image, predict, weight


class ImagingTemplate:
    """Holds all operator templates (imaging.py:11-51).  ``tuning`` maps 'gridder', 'degridder'
    and 'clean' to the tuning dicts of those templates (the reference's per-template autotuning
    results travel the same way, imaging.py:24-29)."""

    def __init__(self, context, array_parameters, fixed_image_parameters,
                 weight_parameters, fixed_grid_parameters, clean_parameters, tuning=None):
        self.context = context
        self.array_parameters = array_parameters
        self.fixed_image_parameters = fixed_image_parameters
        self.weight_parameters = weight_parameters
        self.fixed_grid_parameters = fixed_grid_parameters
        self.clean_parameters = clean_parameters
        dtype = fixed_image_parameters.real_dtype
        num_pols = len(fixed_image_parameters.polarizations)
        tuning = tuning or {}
        self.weights = weight.WeightsTemplate(context, weight_parameters.weight_type, num_pols)
        self.gridder = grid.GridderTemplate(context, fixed_image_parameters,
                                            fixed_grid_parameters, tuning.get('gridder'))
        self.predict = predict.PredictTemplate(context, dtype, num_pols)
        self.grid_image = image.GridImageTemplate(context, dtype, tuning.get('grid_image'))
        self.psf_patch = clean.PsfPatchTemplate(context, dtype, num_pols)
        self.noise_est = clean.NoiseEstTemplate(context, dtype, num_pols)
        self.clean = clean.CleanTemplate(context, clean_parameters, dtype, num_pols,
                                         tuning.get('clean'))
        self.scale = image.ScaleTemplate(context, dtype, num_pols)
        self.add_image = image.AddImageTemplate(context, dtype, num_pols)
        self.apply_primary_beam = image.ApplyPrimaryBeamTemplate(context, dtype, num_pols)
        self.degridder = grid.DegridderTemplate(
            context, fixed_image_parameters, fixed_grid_parameters, tuning.get('degridder')) \
            if fixed_grid_parameters.degrid else None

    def instantiate(self, *args, **kwargs):
        return Imaging(self, *args, **kwargs)


def _get_uv(coords):
    """(N, 4) int16 view of the contiguous ``uv`` and ``sub_uv`` fields (imaging.py:63-78)."""
    dt = coords.dtype
    if dt['uv'] != np.dtype(('i2', (2,))) or dt['sub_uv'] != np.dtype(('i2', (2,))):
        raise TypeError('uv and sub_uv must be pairs of int16')
    if dt.fields['sub_uv'][1] != dt.fields['uv'][1] + 4:
        raise TypeError('uv and sub_uv must be adjacent in the record')
    alias = np.dtype(dict(names=['uv_sub_uv'], formats=[('i2', (4,))],
                          offsets=[dt.fields['uv'][1]], itemsize=dt.itemsize))
    return np.asarray(coords).view(alias)['uv_sub_uv']


def _serial(method):
    """A façade method that is not part of the per-chunk sequence: it must see everything the
    side pipeline has done, and the side pipeline must see what it does."""
    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        self._join()
        try:
            return method(self, *args, **kwargs)
        finally:
            self._fence()
    return wrapper


class _SidePipeline:
    """A second copy of the per-chunk operators (gridder, predictor, continuum predictor) with
    its own chunk buffers on its own HIP stream.

    The gridder and degridder launches of consecutive chunks are separated by a device-wide
    barrier when they share a stream: every launch ends with all 3 072 windows flushing at once
    and the next one starts by staging its kernel table.  Alternating chunks between two streams
    (both accumulate into the same grid with atomics, both read the same model grid) lets the
    tail of one launch overlap the head of the next at workgroup granularity: +18 % gridding and
    +15 % degridding rate with vis_block-sized chunks.
    """

    def __init__(self, main, image_parameters, grid_parameters, max_vis, max_sources, major):
        t = main.template
        context = t.context
        self.main = main
        self.queue = context.create_command_queue()
        q = self.queue
        degrid = grid_parameters.fixed.degrid
        self.gridder = t.gridder.instantiate(q, t.array_parameters, image_parameters,
                                             grid_parameters, max_vis)
        self.continuum = t.predict.instantiate(q, image_parameters, grid_parameters, max_vis,
                                               max_sources)
        if degrid:
            self.predict = t.degridder.instantiate(q, t.array_parameters, image_parameters,
                                                   grid_parameters, max_vis)
        else:
            cp = t.clean_parameters
            max_components = min(image_parameters.pixels ** 2, (major - 1) * cp.minor)
            self.predict = t.predict.instantiate(q, image_parameters, grid_parameters, max_vis,
                                                 max_components)
        self.degrid = degrid
        P = len(image_parameters.fixed.polarizations)
        self.own = dict(
            uv=accel.DeviceArray(context, (max_vis, 4), np.int16),
            w_plane=accel.DeviceArray(context, (max_vis,), np.int16),
            vis=accel.DeviceArray(context, (max_vis, P), np.complex64),
            weights=accel.DeviceArray(context, (max_vis, P), np.float32))
        self.bind_chunk(**self.own)
        self.attach()
        for op in self.ops():
            op.ensure_all_bound()

    def ops(self):
        return (self.gridder, self.predict, self.continuum)

    def bind_chunk(self, **buffers):
        for op in self.ops():
            op.bind(**{k: v for k, v in buffers.items() if k in op.slots})

    def attach(self):
        """(Re)bind the buffers shared with the main pipeline."""
        main = self.main
        self.gridder.bind(grid=main.buffer('grid'), weights_grid=main.buffer('weights_grid'))
        if self.degrid:
            self.predict.bind(grid=main.buffer('degrid'))

    def set_num_vis(self, value):
        for op in self.ops():
            op.num_vis = value

    def buffer(self, name):
        return self.gridder.buffer(name) if name in self.gridder.slots else self.predict.buffer(name)


class Imaging(accel.OperationSequence):
    """All operations and buffers for imaging one channel (imaging.py:81-419)."""

    #: what frontend.process_channel may use beyond the reference's calls (results the same):
    #: :meth:`clean_major_cycles`, and :meth:`scale_dirty_by_centre` / :meth:`psf_patch_start`
    one_call_major_cycles = True
    device_psf_stage = True

    def __init__(self, template, command_queue, image_parameters, grid_parameters,
                 max_vis, max_sources, major, allocator=None, streams=1):
        if streams not in (1, 2):
            raise ValueError('streams must be 1 or 2')
        self._side_args = (image_parameters, grid_parameters, max_vis, max_sources, major) \
            if streams == 2 else None
        self._side = None
        self._cur = 0               # pipeline of the current chunk: 0 main, 1 side
        self._side_dirty = False    # the side stream has work the main stream has not waited for
        self._side_synced = True    # the side stream has waited for the latest main-stream fence
        self._prologue = None
        self._side_num_vis = 0
        assert image_parameters.fixed == template.fixed_image_parameters
        assert grid_parameters.fixed == template.fixed_grid_parameters
        self.template = template
        #: what this imager was made for (frontend.process_channel refuses a channel whose own
        #: parameters differ: the kernel table, the taper and the image geometry follow from them)
        self.image_parameters = image_parameters
        self.grid_parameters = grid_parameters
        lm_scale = float(image_parameters.pixel_size)
        lm_bias = -0.5 * image_parameters.pixels * lm_scale
        num_pols = len(image_parameters.fixed.polarizations)
        pixels = image_parameters.pixels
        image_shape = (num_pols, pixels, pixels)
        fft_plan = template.grid_image.make_fft_plan(image_shape[1:], image_shape[1:])
        cp = template.clean_parameters
        degrid = grid_parameters.fixed.degrid

        self._gridder = template.gridder.instantiate(
            command_queue, template.array_parameters, image_parameters, grid_parameters,
            max_vis, allocator)
        self._continuum_predict = template.predict.instantiate(
            command_queue, image_parameters, grid_parameters, max_vis, max_sources, allocator)
        grid_shape = self._gridder.slots['grid'].shape
        self._weights = template.weights.instantiate(command_queue, grid_shape, max_vis, allocator)
        self._weights.robustness = template.weight_parameters.robustness
        self._grid_to_image = template.grid_image.instantiate_grid_to_image(
            command_queue, grid_shape, lm_scale, lm_bias, fft_plan, allocator)
        self._psf_patch = template.psf_patch.instantiate(command_queue, image_shape, allocator)
        self._noise_est = template.noise_est.instantiate(
            command_queue, image_shape, cp.border, allocator)
        self._clean = template.clean.instantiate(command_queue, image_parameters, allocator)
        self._scale = template.scale.instantiate(command_queue, image_shape, allocator)
        self._add_image = template.add_image.instantiate(command_queue, image_shape, allocator)
        self._apply_primary_beam_model = template.apply_primary_beam.instantiate(
            command_queue, image_shape, 0.0, 0.0, allocator)
        self._apply_primary_beam_dirty = template.apply_primary_beam.instantiate(
            command_queue, image_shape, 0.0, np.nan, allocator)

        context = template.context
        taper1d = accel.DeviceArray(context, (pixels,), image_parameters.fixed.real_dtype)
        taper1d.set(command_queue, self._gridder.convolve_kernel.taper(pixels))
        self._grid_to_image.bind(kernel1d=taper1d)
        self._image_to_grid = None
        if degrid:
            self._predict = template.degridder.instantiate(
                command_queue, template.array_parameters, image_parameters, grid_parameters,
                max_vis, allocator)
            untaper1d = accel.DeviceArray(context, (pixels,), image_parameters.fixed.real_dtype)
            untaper1d.set(command_queue, self._predict.convolve_kernel.taper(pixels))
            self._image_to_grid = template.grid_image.instantiate_image_to_grid(
                command_queue, self._predict.slots['grid'].shape, lm_scale, lm_bias, fft_plan,
                allocator)
            self._image_to_grid.bind(kernel1d=untaper1d)
        else:
            max_components = min(pixels**2, (major - 1) * cp.minor)
            self._predict = template.predict.instantiate(
                command_queue, image_parameters, grid_parameters, max_vis, max_components,
                allocator)
        self._components = {}
        self._pending_components = []
        self._kept_scale = None         # device float32 [P]: 1 / the PSF's central pixel
        self._small_stream = None       # (small read-backs next to the queue's work)
        self._dirty_cleared = False
        operations = [
            ('weights', self._weights), ('gridder', self._gridder), ('predict', self._predict),
            ('continuum_predict', self._continuum_predict),
            ('grid_to_image', self._grid_to_image), ('psf_patch', self._psf_patch),
            ('noise_est', self._noise_est), ('clean', self._clean), ('scale', self._scale),
            ('add_image', self._add_image),
            ('apply_primary_beam_model', self._apply_primary_beam_model),
            ('apply_primary_beam_dirty', self._apply_primary_beam_dirty)]
        # buffer names of imaging.py:185-209
        compounds = {
            'weights': ['predict:weights', 'continuum_predict:weights'],
            'weights_grid': ['weights:grid', 'gridder:weights_grid'],
            'uv': ['gridder:uv', 'predict:uv', 'continuum_predict:uv'],
            'w_plane': ['gridder:w_plane', 'predict:w_plane', 'continuum_predict:w_plane'],
            'vis': ['gridder:vis', 'predict:vis', 'continuum_predict:vis'],
            'grid': ['gridder:grid', 'grid_to_image:grid'],
            'layer': ['grid_to_image:layer'],
            'dirty': ['grid_to_image:image', 'noise_est:dirty', 'clean:dirty', 'scale:data',
                      'add_image:dest', 'apply_primary_beam_dirty:data'],
            'model': ['clean:model', 'apply_primary_beam_model:data', 'add_image:src'],
            'psf': ['clean:psf', 'psf_patch:psf'],
            'tile_max': ['clean:tile_max'], 'tile_pos': ['clean:tile_pos'],
            'peak_value': ['clean:peak_value'], 'peak_pos': ['clean:peak_pos'],
            'peak_pixel': ['clean:peak_pixel'],
            'beam_power': ['apply_primary_beam_model:beam_power',
                           'apply_primary_beam_dirty:beam_power'],
        }
        if 'uv' in self._weights.slots:
            compounds['weights'].insert(0, 'weights:weights')
            compounds['uv'].insert(0, 'weights:uv')
        if degrid:
            operations.append(('image_to_grid', self._image_to_grid))
            compounds['degrid'] = ['predict:grid', 'image_to_grid:grid']
            compounds['layer'].append('image_to_grid:layer')
            compounds['model'].append('image_to_grid:image')
        super().__init__(command_queue, operations, compounds, allocator=allocator)
        self._bound = False
        self._own = {}

    def __call__(self, **kwargs):
        raise NotImplementedError()

    def ensure_all_bound(self):
        super().ensure_all_bound()
        self._bound = True
        if self._side_args is not None and self._side is None:
            self._side = _SidePipeline(self, *self._side_args)
            self.set_window_cus(getattr(self, '_window_cus', 0))
            self._fence()

    # ---- two-stream bookkeeping ---------------------------------------------------------
    def _join(self):
        """Before a non-chunk operation: the main stream waits for the side stream."""
        if self._side is not None and self._side_dirty:
            self.command_queue.enqueue_wait_for_events([self._side.queue.enqueue_marker()])
            self._side_dirty = False
        self._cur = 0

    def _fence(self):
        """After a non-chunk operation: the side stream must not run ahead of it."""
        if self._side is not None:
            self._prologue = self.command_queue.enqueue_marker()
            self._side_synced = False

    def set_window_cus(self, cus):
        """CUs (of 256; 0 = all) the gridder and degridder launches of THIS imager fill
        (``Gridder.window_cus``): ``frontend.process_channel_stream`` sets 192 on the imagers it keeps
        in flight together, so that the other channels' CLEAN launches find room."""
        self._window_cus = int(cus)
        ops = [self._gridder, self._predict]
        if self._side is not None:
            ops += [self._side.gridder, self._side.predict]
        for op in ops:
            if hasattr(op, 'window_cus'):
                op.window_cus = self._window_cus

    def _use_side(self):
        """True if the current chunk belongs to the side pipeline (and make it ready)."""
        if self._side is None or self._cur == 0:
            return False
        if not self._side_synced:
            self._side.queue.enqueue_wait_for_events([self._prologue])
            self._side_synced = True
            self._side.attach()
        self._side_dirty = True
        return True

    def _ready(self, keep_cleared=False):
        if not self._bound:
            self.ensure_all_bound()
        if self._dirty_cleared and not keep_cleared:
            self._dirty_cleared = False
            super().buffer('dirty').zero(self.command_queue)

    def buffer(self, name):
        if name == 'dirty' and self._dirty_cleared:
            self._ready()
        return super().buffer(name)

    def bind(self, **kwargs):
        """As the operations' ``bind``; a deferred :meth:`clear_dirty` belongs to the buffer that was
        bound when it was asked for (the reference zeroes at call time, imaging.py:258-261): it is
        carried out before 'dirty' is bound to another buffer."""
        if 'dirty' in kwargs and getattr(self, '_dirty_cleared', False) and getattr(self, '_bound', False):
            self._ready()
        super().bind(**kwargs)

    # ---- visibilities ---------------------------------------------------------------
    @property
    def num_vis(self):
        return self._gridder.num_vis

    @num_vis.setter
    def num_vis(self, value):
        if self._use_side():
            if value < 0 or value > self._gridder.max_vis:
                raise ValueError('Number of visibilities {} is out of range 0..{}'.format(
                    value, self._gridder.max_vis))
            self._side.set_num_vis(value)
            self._side_num_vis = value
            return
        self._gridder.num_vis = value
        self._predict.num_vis = value
        self._continuum_predict.num_vis = value

    def _chunk_num_vis(self):
        return self._side_num_vis if (self._side is not None and self._cur == 1) else self.num_vis

    def _set_buffer(self, name, N, data, columns=None):
        if len(data) != N:
            raise ValueError('Lengths do not match')
        self._ready()
        if name == 'uv':
            # new coordinates from the host: nothing is known about their order
            side_now = self._use_side()
            (self._side.gridder if side_now else self._gridder).locality_hint = None
            if self.template.fixed_grid_parameters.degrid:
                (self._side.predict if side_now else self._predict).locality_hint = None
        if name in ('uv', 'w_plane', 'vis', 'weights') and self._use_side():
            side = self._side
            if side.buffer(name) is not side.own[name]:
                side.bind_chunk(**{name: side.own[name]})
            device, queue = side.own[name], side.queue
        else:
            self._restore_own(name)
            device, queue = self.buffer(name), self.command_queue
        data = np.asarray(data)
        if columns is None:
            device.set_region(queue, data, np.s_[:N], np.s_[:], blocking=True)
        else:
            device.set_region(queue, data, (np.s_[:N], columns), np.s_[:], blocking=True)

    @_serial
    def clear_weights(self):
        self._ready(keep_cleared=True)     # (does not touch the dirty image)
        self._weights.clear()

    @_serial
    def grid_weights(self, uv, weights):
        if 'uv' not in self._weights.slots:
            return
        self._set_buffer('uv', len(uv), uv, np.s_[:2])
        self._set_buffer('weights', len(uv), weights)
        self._weights.grid(len(uv))

    @_serial
    def finalize_weights(self):
        self._ready(keep_cleared=True)     # (does not touch the dirty image)
        return self._weights.finalize()

    @_serial
    def clear_grid(self):
        self._ready(keep_cleared=True)     # (does not touch the dirty image)
        self.buffer('grid').zero(self.command_queue)

    @_serial
    def clear_dirty(self):
        """imaging.py:258-261.  The fill is deferred: when the next thing to touch the image is a
        :meth:`grid_to_image` that can write it instead of adding to it, neither the fill nor the
        read of the zeros happens; anything else (:meth:`_ready`, :meth:`buffer`) fills first."""
        self._ready()
        self._dirty_cleared = True

    @_serial
    def clear_model(self):
        self._ready(keep_cleared=True)     # (does not touch the dirty image)
        self.buffer('model').zero(self.command_queue)
        self._components.clear()
        del self._pending_components[:]

    def set_coordinates(self, coords):
        """``coords``: structured array with fields ``uv``, ``sub_uv`` (2 x int16 each,
        adjacent) and ``w_plane`` (imaging.py:294-305)."""
        N = self._chunk_num_vis()
        if len(coords) != N:
            raise ValueError('Lengths do not match')
        self._set_buffer('uv', N, _get_uv(coords))
        self._set_buffer('w_plane', N, np.ascontiguousarray(coords['w_plane']))

    def set_vis(self, vis):
        self._set_buffer('vis', self._chunk_num_vis(), vis)

    @_serial
    def bind_chunk(self, num_vis, uv, w_plane, vis, weights=None):
        """Use visibilities that are already resident in HBM (DeviceArrays of the slot shapes)
        instead of copying a host chunk: the zero-copy counterpart of ``num_vis = n;
        set_coordinates(); set_vis(); set_weights()``.  ``vis`` is modified in place by
        ``predict``; see :meth:`set_chunk_device` for a store that must stay intact."""
        self._ready()
        self._keep_own('uv', 'w_plane', 'vis', 'weights')
        self.num_vis = num_vis
        self.bind(uv=uv, w_plane=w_plane, vis=vis)
        self._gridder.locality_hint = None
        if self.template.fixed_grid_parameters.degrid:
            self._predict.locality_hint = None
        if weights is not None:
            self.bind(weights=weights)

    def _keep_own(self, *names):
        """Remember the façade's own buffers before external ones are bound over them."""
        for name in names:
            if name not in self._own:
                self._own[name] = self.buffer(name)

    def _restore_own(self, name):
        if name in self._own and self.buffer(name) is not self._own[name]:
            self.bind(**{name: self._own[name]})

    def set_chunk_device(self, chunk, field='vis'):
        """Device-resident counterpart of the per-chunk part of frontend.make_dirty
        (frontend.py:129-134): ``num_vis = len(chunk); set_coordinates(chunk);
        set_vis(chunk[field]); set_weights(chunk.weights)`` for a
        :class:`preprocess.DeviceChunk`.  Coordinates and weights are bound zero-copy; the
        visibilities are copied (device to device) into the façade's own buffer because
        ``predict`` subtracts the model in place and the store must keep the originals."""
        if field not in ('vis', 'weights'):
            raise ValueError('field must be vis or weights')
        self._ready(keep_cleared=True)     # (does not touch the dirty image)
        if self._use_side():
            side = self._side
            n = chunk.num_vis
            self.num_vis = n
            side.bind_chunk(uv=chunk.uv, w_plane=chunk.w_plane, weights=chunk.weights,
                            vis=side.own['vis'])
            side.gridder.locality_hint = getattr(chunk, 'locality', None)
            if self.template.fixed_grid_parameters.degrid:
                side.predict.locality_hint = getattr(chunk, 'locality', None)
            if field == 'vis':
                chunk.vis.copy_region(side.queue, side.own['vis'], np.s_[:n], np.s_[:n])
            else:
                from ._lib import lib, check
                check(lib().kimg_real_to_complex(side.own['vis'].ptr, chunk.weights.ptr,
                                                 n * side.own['vis'].shape[1], side.queue.handle),
                      'kimg_real_to_complex')
            return
        self._keep_own('uv', 'w_plane', 'weights')
        self._restore_own('vis')
        n = chunk.num_vis
        self.num_vis = n
        self.bind(uv=chunk.uv, w_plane=chunk.w_plane, weights=chunk.weights)
        self._gridder.locality_hint = getattr(chunk, 'locality', None)
        if self.template.fixed_grid_parameters.degrid:
            self._predict.locality_hint = getattr(chunk, 'locality', None)
        own_vis = self.buffer('vis')
        if field == 'vis':
            chunk.vis.copy_region(self.command_queue, own_vis, np.s_[:n], np.s_[:n])
        else:
            from ._lib import lib, check
            check(lib().kimg_real_to_complex(own_vis.ptr, chunk.weights.ptr, n * own_vis.shape[1],
                                             self.command_queue.handle), 'kimg_real_to_complex')

    @_serial
    def grid_weights_device(self, chunk):
        """``grid_weights(chunk.uv, chunk.weights)`` (frontend.py:96) without the host copy."""
        if 'uv' not in self._weights.slots:
            return
        self._ready()
        self._keep_own('uv', 'weights')
        self.bind(uv=chunk.uv, weights=chunk.weights)
        self._weights.grid(chunk.num_vis)

    def set_weights(self, weights):
        """Statistical weights for prediction."""
        self._set_buffer('weights', self._chunk_num_vis(), weights)

    # ---- operations -------------------------------------------------------------------
    def grid(self):
        """Grid the current chunk; with ``streams=2`` the next chunk goes to the other stream."""
        self._ready(keep_cleared=True)     # (does not touch the dirty image)
        if self._use_side():
            self._side.gridder()
        else:
            self._gridder()
        if self._side is not None:
            self._cur ^= 1

    def predict(self, w):
        self._ready(keep_cleared=True)     # (does not touch the dirty image)
        op = self._side.predict if self._use_side() else self._predict
        if not self.template.fixed_grid_parameters.degrid:
            op.set_w(w)
        op()

    def continuum_predict(self, w):
        self._ready(keep_cleared=True)     # (does not touch the dirty image)
        op = self._side.continuum if self._use_side() else self._continuum_predict
        op.set_w(w)
        op()

    @_serial
    def set_sky_arrays(self, lmn, flux):
        """Continuum model as arrays (l, m, n-1) / flux[P]; see predict.Predict."""
        self._ready(keep_cleared=True)     # (does not touch the dirty image)
        self._continuum_predict.set_sky_arrays(lmn, flux)
        if self._side is not None:
            self._side.continuum.set_sky_arrays(lmn, flux)

    @_serial
    def set_sky_model(self, sky_model, phase_centre):
        """imaging.py:331-332: continuum model from an object with the reference's SkyModel
        interface (``lmn(phase_centre)``, ``flux_density(wavelength)``, ``len``)."""
        self._ready(keep_cleared=True)     # (does not touch the dirty image)
        self._continuum_predict.set_sky_model(sky_model, phase_centre)
        if self._side is not None:
            self._side.continuum.set_sky_model(sky_model, phase_centre)

    @_serial
    def grid_to_image(self, w):
        self._ready(keep_cleared=True)
        self._grid_to_image.set_w(w)
        if self._dirty_cleared:
            if self._grid_to_image.can_overwrite():
                self._dirty_cleared = False
                self._grid_to_image.overwrite_next = True
            else:
                self._ready()
        self._grid_to_image()

    @_serial
    def model_to_grid(self, w):
        if not self._image_to_grid:
            raise RuntimeError('Can only use model_to_grid with degridding')
        self._ready(keep_cleared=True)     # (does not touch the dirty image)
        self._image_to_grid.set_w(w)
        self._image_to_grid()

    @_serial
    def model_to_predict(self):
        if self.template.fixed_grid_parameters.degrid:
            raise RuntimeError('Can only use model_to_predict with direct prediction')
        self._ready(keep_cleared=True)     # (does not touch the dirty image)
        self._predict.set_sky_image(self._model_components)
        if self._side is not None:
            self._side.predict.set_sky_image(self._model_components)

    @_serial
    def scale_dirty(self, scale_factor):
        self._ready()
        self._scale.set_scale_factor(scale_factor)
        self._scale()

    @_serial
    def add_model_to_dirty(self):
        self._ready()
        self._add_image()

    @_serial
    def apply_primary_beam(self, threshold):
        """Divide model and dirty images by the primary beam power (imaging.py:362-367)."""
        self._ready()
        self._apply_primary_beam_model.threshold = threshold
        self._apply_primary_beam_model()
        self._apply_primary_beam_dirty.threshold = threshold
        self._apply_primary_beam_dirty()

    @_serial
    def dirty_to_psf(self):
        """Swap the dirty and PSF buffers (imaging.py:370-373)."""
        self._ready()
        dirty = self.buffer('dirty')
        psf = self.buffer('psf')
        self.bind(dirty=psf, psf=dirty)

    @_serial
    def psf_patch(self):
        self._ready()
        cp = self.template.clean_parameters
        return self._psf_patch(cp.psf_cutoff, cp.psf_limit)

    # ---- the PSF stage without the host in between (frontend.py:541-548) -----------------
    # The reference reads the PSF's central pixel back, scales by its reciprocal, and reads the
    # patch back; both read-backs leave the device idle while the host reacts.  Here the
    # reciprocal stays on the device, and the small results travel on a stream of their own
    # while the next gridding runs.
    @_serial
    def scale_dirty_by_centre(self):
        """``scale_dirty(1 / dirty[:, centre, centre])`` with the factors made and kept on the
        device (:meth:`scale_dirty_by_kept` applies them again, :meth:`psf_patch_finish` hands
        them to the host)."""
        from ._lib import lib, check
        self._ready()
        dirty = self.buffer('dirty')
        P, H, W = dirty.shape
        if self._kept_scale is None:
            self._kept_scale = accel.DeviceArray(self.command_queue.context, (P,), np.float32,
                                                 queue=self.command_queue)
        centre = H // 2             # (frontend.py:541: one index for both axes)
        check(lib().kimg_pixel_reciprocal(dirty.ptr, W, H * W, W, H, P, centre, centre,
                                          self._kept_scale.ptr, self.command_queue.handle),
              'kimg_pixel_reciprocal')
        self.scale_dirty_by_kept()

    @_serial
    def scale_dirty_by_kept(self):
        from ._lib import lib, check
        self._ready()
        dirty = self.buffer('dirty')
        P, H, W = dirty.shape
        check(lib().kimg_scale_device(dirty.ptr, W, H * W, W, H, P, self._kept_scale.ptr,
                                      self.command_queue.handle), 'kimg_scale_device')

    @_serial
    def psf_patch_start(self):
        """:meth:`psf_patch` without its read-back: the search is enqueued, and its two bounds and
        the kept scale factors are copied to the host on a stream of their own behind it.
        :meth:`psf_patch_finish` returns (patch, scale factors)."""
        import torch
        self._ready()
        cp = self.template.clean_parameters
        self._psf_patch.enqueue(cp.psf_cutoff, cp.psf_limit)
        q = self.command_queue
        bound = self._psf_patch.buffer('bound')
        bound.used_on(q)
        self._kept_scale.used_on(q)
        if self._small_stream is None:
            self._small_stream = torch.cuda.Stream(device=q.stream.device)
        host_bound = torch.empty(2, dtype=torch.int32, pin_memory=True)
        host_scale = torch.empty(self._kept_scale.shape[0], dtype=torch.float32, pin_memory=True)
        ready = torch.cuda.Event()
        ready.record(q.stream)
        with torch.cuda.stream(self._small_stream):
            self._small_stream.wait_event(ready)
            host_bound.copy_(bound.tensor.reshape(-1)[:2].view(torch.int32), non_blocking=True)
            host_scale.copy_(self._kept_scale.tensor.reshape(-1), non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self._small_stream)
        return copied, host_bound, host_scale

    def psf_patch_finish(self, started):
        copied, host_bound, host_scale = started
        copied.synchronize()
        return self._psf_patch.finish(host_bound.numpy().copy()), host_scale.numpy().copy()

    @_serial
    def noise_est(self):
        self._ready()
        return self._noise_est()

    @_serial
    def clean_reset(self):
        self._ready()
        self._clean.reset()

    @property
    def _model_components(self):
        """{(y, x): flux per polarization} of the CLEAN components so far (imaging.py:257, :392-394).
        The components of the device-resident loops are kept as the arrays they come in and only
        folded into the dictionary when somebody asks for it (``model_to_predict``, tests): with
        degridding nobody does, and a thousand dictionary updates per major cycle are 0.5 ms of
        host time that the next channel's launches would wait for."""
        self._settle_components()
        for positions, pixels in self._pending_components:
            self._record_many([tuple(p) for p in positions.tolist()], pixels)
        del self._pending_components[:]
        return self._components

    @_model_components.setter
    def _model_components(self, value):
        del self._pending_components[:]
        self._components = value

    @_serial
    def clean_cycle(self, psf_patch, threshold=0.0):
        """One minor cycle; returns the peak metric or None (imaging.py:389-396)."""
        self._ready()
        self._settle_components()
        peak_value, peak_pos, model_pixel = self._clean(psf_patch, threshold)
        if peak_pos is not None:
            # (kept with the pending arrays: folding a thousand components of the last major
            # cycle into the dictionary here would stall the first cycle of this one)
            self._pending_components.append(
                (np.array([peak_pos], np.int32), np.asarray(model_pixel)[np.newaxis]))
        return peak_value

    @_serial
    def clean_cycles(self, psf_patch, threshold, max_cycles, batcher=None):
        """Up to `max_cycles` minor cycles without host round trips; returns the list of
        peak metrics (shorter than `max_cycles` iff the threshold was reached).  ``batcher``
        (a :class:`clean.CleanBatcher` shared by the channels imaged concurrently) lets the cycles
        of those channels share their launches; the results are the same."""
        self._ready()
        self._settle_components()
        if max_cycles <= 0:
            return []
        if batcher is not None:
            values, positions, pixels = batcher.run_cycles(self._clean, psf_patch, threshold,
                                                           max_cycles, arrays=True)
        else:
            self._clean.run_cycles(psf_patch, threshold, max_cycles, collect=False)
            values, positions, pixels = self._clean._collect_cycle_arrays()
        if len(values):
            self._pending_components.append((positions, pixels))
        return values.tolist()

    @_serial
    def clean_major_cycles(self, psf_patch, noise_threshold, left_for_next, max_cycles, batcher=None):
        """The minor cycles of one major cycle (the first one included) in one call, the threshold
        following from the first peak on the device (:meth:`clean.Clean.run_major_cycles`).  Returns
        (metric of the first cycle, cycles done) -- which the call has without reading the device
        back: the components themselves are on their way on a stream of their own and are looked at
        when somebody asks for them -- or None where that is not available (the caller then runs
        :meth:`clean_cycle` and :meth:`clean_cycles` as the reference does).  ``batcher``: the
        :class:`clean.CleanBatcher` of the channels in flight, which has to know."""
        self._ready()
        self._settle_components()       # (the last call's read-back, before its buffers are written again)
        if batcher is not None:
            got = batcher.run_major_cycles(self._clean, psf_patch, noise_threshold, left_for_next, max_cycles)
        else:
            got = self._clean.run_major_cycles(psf_patch, noise_threshold, left_for_next, max_cycles)
        if not got:
            return None
        count, first = got
        if count <= 0:
            return None                 # (nothing there to CLEAN: as the reference's steps find out)
        self._pending_components.append(self._clean)           # (its read-back in flight)
        return first, count

    def _settle_components(self):
        """Turn read-backs in flight among the pending components into the arrays they bring."""
        for i, item in enumerate(self._pending_components):
            if not isinstance(item, tuple):
                values, positions, pixels = item._collect_cycle_arrays()
                self._pending_components[i] = (positions, pixels)

    def _record_many(self, positions, pixels):
        """:meth:`_record` for a whole call's components at once: the fluxes of every position are
        added in cycle order (``np.add.at`` is unbuffered and goes through its indices in order),
        exactly as repeated calls of :meth:`_record` would add them."""
        width = self.buffer('dirty').shape[2]
        keys = np.array([y * width + x for y, x in positions], np.int64)
        uniq, first, inverse = np.unique(keys, return_index=True, return_inverse=True)
        total = np.zeros((len(uniq),) + pixels.shape[1:], pixels.dtype)
        have = np.zeros(len(uniq), bool)
        for k, i in enumerate(first):
            old = self._components.get(positions[i])
            if old is not None:
                total[k] = old
                have[k] = True
        # a new position starts from its first component itself (not 0 + component)
        start = ~have
        total[start] = pixels[first[start]]
        rest = np.ones(len(keys), bool)
        rest[first[start]] = False
        np.add.at(total, inverse[rest], pixels[rest])
        for k, i in enumerate(first):
            self._components[positions[i]] = total[k]

    # ---- buffers -----------------------------------------------------------------------
    @_serial
    def get_buffer(self, name):
        """Contents of a buffer as a numpy array (imaging.py:399-401)."""
        self._ready()
        return self.buffer(name).get(self.command_queue)

    @_serial
    def set_buffer(self, name, data):
        self._ready()
        self.buffer(name).set(self.command_queue, data)

    @_serial
    def free_buffer(self, name):
        if name == 'dirty':
            self._dirty_cleared = False     # (a deferred fill has nothing left to fill)
        if name in self.slots:
            self.slots[name].bind(None)
