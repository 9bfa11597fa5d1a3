"""Tiled Hogbom CLEAN on MI355X.

Operator surface of the reference's ``katsdpimager.clean`` (PsfPatch, NoiseEst,
_UpdateTiles, _FindPeak, _SubtractPsf, Clean; clean.py:37-891) on libkimg.so,
plus :meth:`Clean.run_cycles`: a device-resident minor-cycle loop that removes
the per-cycle host round trip of clean.py:870-878 while producing the same
component list.

Peak selection and tie-breaks are bit-exact with the reference HOST path
(CleanHost, clean.py:971-1075); see csrc/clean.hip.
"""
import math

import numpy as np

from . import accel, types
from ._lib import lib, check
from .parameters import CLEAN_I, CLEAN_SUMSQ  # noqa: F401

#: median(|x|) -> sigma for a zero-mean Gaussian (clean.py:34)
_MEDIAN_TO_RMS = 1.4826022185056031

TILE = 32


def metric_to_power(mode, metric):
    """clean.py:166-174."""
    if mode == CLEAN_I:
        return metric
    elif mode == CLEAN_SUMSQ:
        return math.sqrt(metric)
    raise ValueError('Invalid mode {}'.format(mode))


def power_to_metric(mode, power):
    """clean.py:177-184."""
    if mode == CLEAN_I:
        return power
    elif mode == CLEAN_SUMSQ:
        return power * power
    raise ValueError('Invalid mode {}'.format(mode))


def noise_threshold_scale(mode, threshold, num_polarizations):
    """clean.py:187-203."""
    if mode == CLEAN_I:
        return threshold
    elif mode == CLEAN_SUMSQ:
        import scipy.stats
        p = 2 * scipy.stats.norm.sf(threshold)
        return np.sqrt(scipy.stats.chi2.isf(p, num_polarizations))
    raise ValueError('Invalid mode {}'.format(mode))


def _image_args(array):
    P, H, W = array.shape
    return array.ptr, W, H * W, W, H, P


class PsfPatchTemplate:
    """clean.py:37-69."""
    def __init__(self, context, dtype, num_polarizations, tuning=None):
        types.require_float32(dtype, 'PsfPatchTemplate')
        lib()
        self.context = context
        self.num_polarizations = num_polarizations
        self.dtype = np.dtype(dtype)

    def instantiate(self, *args, **kwargs):
        return PsfPatch(self, *args, **kwargs)


class PsfPatch(accel.Operation):
    """Size of the central PSF box holding every |psf| >= threshold (clean.py:72-163).
    Slots: **psf** [P][H][W]; **bound** int32 [2]."""

    def __init__(self, template, command_queue, shape, allocator=None):
        if shape[0] != template.num_polarizations:
            raise ValueError('Mismatch in number of polarizations')
        super().__init__(command_queue, allocator)
        self.template = template
        self.slots['psf'] = accel.IOSlot(shape, template.dtype)
        self.slots['bound'] = accel.IOSlot((2,), np.int32)

    def _run(self):
        pass

    def __call__(self, threshold, limit=None, **kwargs):
        return self.finish(self.enqueue(threshold, limit, **kwargs))

    def finish(self, bound_values=None):
        """The patch from the two bounds :meth:`enqueue` left in **bound** (read here if the caller
        has not fetched them some other way)."""
        P, H, W = self.buffer('psf').shape
        b = self.buffer('bound').get(self.command_queue) if bound_values is None or bound_values is True \
            else np.asarray(bound_values)
        box = 2 * b + 1
        return (P, int(min(box[1], H)), int(min(box[0], W)))

    def enqueue(self, threshold, limit=None, **kwargs):
        """Launch the search only; :meth:`finish` reads its result."""
        self.bind(**kwargs)
        self.ensure_all_bound()
        psf = self.buffer('psf')
        bound = self.buffer('bound')
        P, H, W = psf.shape
        min_x, min_y, max_x, max_y = 0, 0, W - 1, H - 1
        mid_x, mid_y = W // 2, H // 2
        if limit is not None:
            hlimit = (round(limit * min(H, W)) - 1) // 2
            min_x = max(min_x, mid_x - hlimit)
            min_y = max(min_y, mid_y - hlimit)
            max_x = min(max_x, mid_x + hlimit)
            max_y = min(max_y, mid_y + hlimit)
        rc = lib().kimg_psf_patch(psf.ptr, W, H * W, P, min_x, min_y, max_x, max_y, mid_x, mid_y,
                                  threshold, bound.ptr, self.command_queue.handle)
        check(rc, 'kimg_psf_patch')
        return True


class NoiseEstTemplate:
    """clean.py:206-244."""
    def __init__(self, context, dtype, num_polarizations, tuning=None):
        types.require_float32(dtype, 'NoiseEstTemplate')
        lib()
        self.context = context
        self.dtype = np.dtype(dtype)
        self.num_polarizations = num_polarizations

    def instantiate(self, *args, **kwargs):
        return NoiseEst(self, *args, **kwargs)


class NoiseEst(accel.Operation):
    """Robust noise estimate median(|dirty| inside the border) * 1.4826 (clean.py:247-353).

    The reference's GPU class bisects the float bit pattern to ~1e-4 relative accuracy with
    ~30 ranking passes; this one finds the EXACT median (as the reference's host path,
    clean.py:938-943) with a 4-pass byte-wise radix select that resolves the two middle ranks side by side.
    Slots: **dirty** [P][H][W]; **rank** uint32 [256] (histogram scratch).
    """

    def __init__(self, template, command_queue, image_shape, border, allocator=None):
        if image_shape[0] != template.num_polarizations:
            raise ValueError('Mismatch in number of polarizations')
        if border >= 0.5:
            raise ValueError('Border must be less than half the image size')
        super().__init__(command_queue, allocator)
        self.template = template
        self.border_pixels = round(border * min(image_shape[1], image_shape[2]))
        self.slots['dirty'] = accel.IOSlot(image_shape, template.dtype)
        self.slots['rank'] = accel.IOSlot((256,), np.uint32)
        self._scratch = None
        self._result = None

    def _run(self):
        pass

    def _kth(self, k):
        """k-th smallest (0-based) |x| as a uint32 bit pattern."""
        dirty, hist = self.buffer('dirty'), self.buffer('rank')
        prefix = 0
        for p in (3, 2, 1, 0):
            rc = lib().kimg_abs_histogram(*_image_args(dirty), self.border_pixels, p, prefix,
                                          hist.ptr, self.command_queue.handle)
            check(rc, 'kimg_abs_histogram')
            h = hist.get(self.command_queue).astype(np.int64)
            cum = np.cumsum(h)
            digit = int(np.searchsorted(cum, k, side='right'))
            k -= int(cum[digit - 1]) if digit else 0
            prefix = (prefix << 8) | digit
        return prefix

    def __call__(self, **kwargs):
        """One call, one 4-byte read-back: the radix-select passes pick their byte on the device
        (``kimg_noise_est``)."""
        self.bind(**kwargs)
        self.ensure_all_bound()
        dirty = self.buffer('dirty')
        if self._scratch is None:
            ctx = self.command_queue.context
            self._scratch = accel.DeviceArray(
                ctx, (lib().kimg_noise_est_scratch_bytes() // 4,), np.uint32, queue=self.command_queue)
            self._result = accel.DeviceArray(ctx, (1,), np.float32, queue=self.command_queue)
        rc = lib().kimg_noise_est(*_image_args(dirty), self.border_pixels,
                                  float(np.float32(_MEDIAN_TO_RMS)), self._scratch.ptr,
                                  self._result.ptr, self.command_queue.handle)
        check(rc, 'kimg_noise_est')
        return self._result.get(self.command_queue)[0]

    def host_select(self):
        """The same estimate with the byte selection on the host (one round trip per pass); kept
        as the reference implementation of the device selection (tests)."""
        self.ensure_all_bound()
        dirty = self.buffer('dirty')
        P, H, W = dirty.shape
        bp = self.border_pixels
        n = (H - 2 * bp) * (W - 2 * bp) * P
        lo_bits = self._kth((n - 1) // 2)
        lo = np.array([lo_bits], np.uint32).view(np.float32)[0]
        hi = lo
        if n % 2 == 0:
            out = self.buffer('rank')
            rc = lib().kimg_abs_count_le(*_image_args(dirty), bp, float(lo), out.ptr,
                                         self.command_queue.handle)
            check(rc, 'kimg_abs_count_le')
            res = out.get(self.command_queue)
            if int(res[0]) <= n // 2:          # the upper middle element is the next value up
                hi = np.array([res[1]], np.uint32).view(np.float32)[0]
        median = (lo + hi) / np.float32(2)     # np.median of float32 data (clean.py:942)
        return median * np.float32(_MEDIAN_TO_RMS)


class _CleanStepTemplate:
    def __init__(self, context, dtype, num_polarizations, tuning=None):
        types.require_float32(dtype, type(self).__name__)
        lib()
        self.context = context
        self.dtype = np.dtype(dtype)
        self.num_polarizations = num_polarizations


def _tile_shape(image_shape, border_pixels):
    return (accel.divup(image_shape[1] - 2 * border_pixels, TILE),
            accel.divup(image_shape[2] - 2 * border_pixels, TILE))


class _UpdateTilesTemplate(_CleanStepTemplate):
    """clean.py:356-395."""
    def __init__(self, context, dtype, num_polarizations, mode, tuning=None):
        super().__init__(context, dtype, num_polarizations, tuning)
        self.mode = mode
        self.tilex = self.tiley = TILE

    def instantiate(self, *args, **kwargs):
        return _UpdateTiles(self, *args, **kwargs)


class _UpdateTiles(accel.Operation):
    """Peak (value and position) of every 32x32 tile intersecting a window
    (clean.py:398-480).  Slots **dirty**, **tile_max** [ty][tx], **tile_pos** [ty][tx][2]."""

    def __init__(self, template, command_queue, image_shape, border, allocator=None):
        if image_shape[0] != template.num_polarizations:
            raise ValueError('Mismatch in number of polarizations')
        if border >= 0.5:
            raise ValueError('Border must be less than half the image size')
        super().__init__(command_queue, allocator)
        self.template = template
        self.border_pixels = round(border * min(image_shape[1], image_shape[2]))
        ty, tx = _tile_shape(image_shape, self.border_pixels)
        self.slots['dirty'] = accel.IOSlot(image_shape, template.dtype)
        self.slots['tile_max'] = accel.IOSlot((ty, tx), template.dtype)
        self.slots['tile_pos'] = accel.IOSlot((ty, tx, accel.Dimension(2, exact=True)), np.int32)

    def _run(self):
        pass

    def __call__(self, x0, y0, x1, y1, **kwargs):
        """Update all tiles intersected by the pixel range [x0, x1) x [y0, y1)."""
        self.bind(**kwargs)
        self.ensure_all_bound()
        tile_max, tile_pos = self.buffer('tile_max'), self.buffer('tile_pos')
        bp = self.border_pixels
        x0 = max((x0 - bp) // TILE, 0)
        y0 = max((y0 - bp) // TILE, 0)
        x1 = min(accel.divup(x1 - bp, TILE), tile_max.shape[1])
        y1 = min(accel.divup(y1 - bp, TILE), tile_max.shape[0])
        if x0 < x1 and y0 < y1:
            rc = lib().kimg_update_tiles(
                *_image_args(self.buffer('dirty')), bp, self.template.mode,
                tile_max.ptr, tile_pos.ptr, tile_max.shape[1], tile_max.shape[0],
                x0, y0, x1, y1, self.command_queue.handle)
            check(rc, 'kimg_update_tiles')


class _FindPeakTemplate(_CleanStepTemplate):
    """clean.py:483-513."""
    def instantiate(self, *args, **kwargs):
        return _FindPeak(self, *args, **kwargs)


class _FindPeak(accel.Operation):
    """Global peak from the per-tile peaks (clean.py:516-587)."""

    def __init__(self, template, command_queue, image_shape, tile_shape, allocator=None):
        if image_shape[0] != template.num_polarizations:
            raise ValueError('Mismatch in number of polarizations')
        super().__init__(command_queue, allocator)
        self.template = template
        pair = accel.Dimension(2, exact=True)
        self.slots['dirty'] = accel.IOSlot(image_shape, template.dtype)
        self.slots['tile_max'] = accel.IOSlot(tile_shape, template.dtype)
        self.slots['tile_pos'] = accel.IOSlot(tuple(tile_shape) + (pair,), np.int32)
        self.slots['peak_value'] = accel.IOSlot([1], template.dtype)
        self.slots['peak_pos'] = accel.IOSlot([2], np.int32)
        self.slots['peak_pixel'] = accel.IOSlot([template.num_polarizations], template.dtype)

    def _run(self):
        dirty = self.buffer('dirty')
        tile_max = self.buffer('tile_max')
        P, H, W = dirty.shape
        rc = lib().kimg_find_peak(
            dirty.ptr, W, H * W, P, tile_max.ptr, self.buffer('tile_pos').ptr,
            tile_max.shape[1], tile_max.shape[0], self.buffer('peak_value').ptr,
            self.buffer('peak_pos').ptr, self.buffer('peak_pixel').ptr,
            self.command_queue.handle)
        check(rc, 'kimg_find_peak')


class _SubtractPsfTemplate(_CleanStepTemplate):
    """clean.py:590-622."""
    def instantiate(self, *args, **kwargs):
        return _SubtractPsf(self, *args, **kwargs)


class _SubtractPsf(accel.Operation):
    """dirty[patch] -= loop_gain * peak_pixel * psf[centre patch]; model[pos] += ...
    (clean.py:625-726)."""

    def __init__(self, template, command_queue, loop_gain, image_shape, psf_shape,
                 allocator=None):
        super().__init__(command_queue, allocator)
        if image_shape[0] != template.num_polarizations:
            raise ValueError('Mismatch in number of polarizations')
        if psf_shape[0] != template.num_polarizations:
            raise ValueError('Mismatch in number of polarizations')
        self.slots['dirty'] = accel.IOSlot(image_shape, template.dtype)
        self.slots['model'] = accel.IOSlot(image_shape, template.dtype)
        self.slots['psf'] = accel.IOSlot(psf_shape, template.dtype)
        self.slots['peak_pixel'] = accel.IOSlot([template.num_polarizations], template.dtype)
        self.loop_gain = loop_gain
        self.template = template

    def _run(self):
        pass

    def __call__(self, pos, psf_patch, **kwargs):
        self.bind(**kwargs)
        self.ensure_all_bound()
        dirty, psf = self.buffer('dirty'), self.buffer('psf')
        P, H, W = dirty.shape
        rc = lib().kimg_subtract_psf(
            dirty.ptr, self.buffer('model').ptr, W, H * W, W, H, P,
            psf.ptr, psf.shape[2], psf.shape[1] * psf.shape[2], psf.shape[2], psf.shape[1],
            psf_patch[2], psf_patch[1], self.buffer('peak_pixel').ptr,
            pos[1], pos[0], self.loop_gain, self.command_queue.handle)
        check(rc, 'kimg_subtract_psf')


CLEAN_FORMS = {'auto': 0, 'two_launch': 1, 'one_launch': 2, 'persistent': 3, 'one_workgroup': 4,
               'multi': 5}     # KIMG_CLEAN_FORM_*


class CleanTemplate:
    """clean.py:729-753.  ``tuning`` may hold ``{'form': 'auto'|'two_launch'|'one_launch'|
    'persistent'|'one_workgroup'|'multi'}``, the form of the device-resident loop of :meth:`Clean.run_cycles`
    (results are identical; ``auto`` takes the fastest one the PSF patch allows), and with ``multi``
    ``'components'``: the most lattices a launch may plan (1 to 8, default 8); ``'repeats'``: the
    most steps it may take at one peak (1 to 8, default: as many as the form takes -- 8, or 4 with
    several polarizations; the loop takes repeated steps only while the field shows few components per
    launch, ``'repeats_always'``: from the first launch on, whatever the field looks like)."""
    def __init__(self, context, clean_parameters, dtype, num_polarizations, tuning=None):
        types.require_float32(dtype, 'CleanTemplate')
        tuning = tuning or {}
        if set(tuning) - {'form', 'components', 'repeats', 'repeats_always'} or tuning.get('form', 'auto') not in CLEAN_FORMS \
                or not 0 <= int(tuning.get('components', 0)) <= 8 \
                or not 0 <= int(tuning.get('repeats', 0)) <= 8:
            raise ValueError('bad CleanTemplate tuning {}'.format(tuning))
        self.form = CLEAN_FORMS[tuning.get('form', 'auto')] | int(tuning.get('components', 0)) << 8 \
            | int(tuning.get('repeats', 0)) << 16 | (1 << 20 if tuning.get('repeats_always') else 0)
        self.context = context
        self.clean_parameters = clean_parameters
        self.dtype = np.dtype(dtype)
        self.num_polarizations = num_polarizations
        self._update_tiles = _UpdateTilesTemplate(context, dtype, num_polarizations,
                                                  clean_parameters.mode)
        self._find_peak = _FindPeakTemplate(context, dtype, num_polarizations)
        self._subtract_psf = _SubtractPsfTemplate(context, dtype, num_polarizations)

    def instantiate(self, *args, **kwargs):
        return Clean(self, *args, **kwargs)


class Clean(accel.OperationSequence):
    """CLEAN minor cycles (clean.py:756-891).  Slots: **dirty**, **model**, **psf**,
    **tile_max**, **tile_pos**, **peak_value**, **peak_pos**, **peak_pixel**."""

    def __init__(self, template, command_queue, image_parameters, allocator=None):
        if image_parameters.fixed.real_dtype != template.dtype:
            raise ValueError('dtype mismatch')
        image_shape = (len(image_parameters.fixed.polarizations),
                       image_parameters.pixels, image_parameters.pixels)
        self.template = template
        cp = template.clean_parameters
        self._update_tiles = template._update_tiles.instantiate(
            command_queue, image_shape, cp.border, allocator)
        tile_shape = self._update_tiles.slots['tile_max'].shape
        self._find_peak = template._find_peak.instantiate(
            command_queue, image_shape, tile_shape, allocator)
        self._subtract_psf = template._subtract_psf.instantiate(
            command_queue, cp.loop_gain, image_shape, image_shape, allocator)
        ops = [('update_tiles', self._update_tiles), ('find_peak', self._find_peak),
               ('subtract_psf', self._subtract_psf)]
        compounds = {
            'dirty': ['update_tiles:dirty', 'find_peak:dirty', 'subtract_psf:dirty'],
            'model': ['subtract_psf:model'],
            'psf': ['subtract_psf:psf'],
            'tile_max': ['update_tiles:tile_max', 'find_peak:tile_max'],
            'tile_pos': ['update_tiles:tile_pos', 'find_peak:tile_pos'],
            'peak_value': ['find_peak:peak_value'],
            'peak_pos': ['find_peak:peak_pos'],
            'peak_pixel': ['find_peak:peak_pixel', 'subtract_psf:peak_pixel'],
        }
        super().__init__(command_queue, ops, compounds, allocator=allocator)
        self._state = accel.DeviceArray(
            command_queue.context, (lib().kimg_clean_state_bytes(image_shape[0], tile_shape[1], tile_shape[0]) // 4,),
            np.int32, queue=command_queue)
        self._log = None
        self._log_rows = 0
        # the read-back run_major_cycles starts: its stream, its pinned buffer, (event, rows) while
        # it is in flight, (host copy, rows) once it has arrived
        self._read_stream = self._read_host = self._read_back = self._read_result = None

    def _run(self):
        raise NotImplementedError('use __call__(psf_patch, threshold) or run_cycles')

    def reset(self):
        """Call after populating the buffers but before the first minor cycle (clean.py:842)."""
        self.ensure_all_bound()
        dirty = self.buffer('dirty')
        self._update_tiles(0, 0, dirty.shape[2], dirty.shape[1])

    def __call__(self, psf_patch, threshold=0.0):
        """One minor cycle with a host round trip, as clean.py:848-891.  Returns
        (peak_value, (y, x), loop_gain * pixel) or (None, None, None) below threshold."""
        self.ensure_all_bound()
        self._find_peak()
        q = self.command_queue
        # value, position and pixel in one read-back (three would be three host round trips)
        import torch
        parts = [self.buffer(name) for name in ('peak_value', 'peak_pos', 'peak_pixel')]
        for part in parts:
            part.used_on(q)
        with torch.cuda.stream(q.stream):
            both = torch.cat([parts[0].tensor, parts[1].tensor.view(torch.float32),
                              parts[2].tensor]).cpu().numpy()
        peak_value = both[:1]
        if peak_value[0] < threshold:
            return None, None, None
        peak_pos = tuple(int(x) for x in both[1:3].view(np.int32))
        peak_pixel = both[3:].copy()
        model_pixel = np.float32(self.template.clean_parameters.loop_gain) * peak_pixel
        self._subtract_psf(peak_pos, psf_patch)
        x0 = peak_pos[1] - psf_patch[2] // 2
        y0 = peak_pos[0] - psf_patch[1] // 2
        self._update_tiles(x0, y0, x0 + psf_patch[2], y0 + psf_patch[1])
        return peak_value[0], peak_pos, model_pixel

    def run_cycles(self, psf_patch, threshold, max_cycles, collect=True):
        """Up to `max_cycles` minor cycles entirely on the device (no host sync between
        cycles).  Returns a list of (peak_value, (y, x), model_pixel) -- exactly what
        `max_cycles` calls of :meth:`__call__` would have returned before the first None.
        ``collect=False`` only enqueues (read the results with :meth:`_collect_cycle_arrays`)."""
        self.ensure_all_bound()
        if max_cycles <= 0:
            return []
        dirty, psf = self.buffer('dirty'), self.buffer('psf')
        P, H, W = dirty.shape
        cp = self.template.clean_parameters
        self._ensure_log(max_cycles)
        tile_max = self.buffer('tile_max')
        rc = lib().kimg_clean_cycles(
            dirty.ptr, self.buffer('model').ptr, W, H * W, W, H, P,
            psf.ptr, psf.shape[2], psf.shape[1] * psf.shape[2], psf.shape[2], psf.shape[1],
            psf_patch[2], psf_patch[1], self._update_tiles.border_pixels, cp.mode,
            cp.loop_gain, threshold, tile_max.ptr, self.buffer('tile_pos').ptr,
            tile_max.shape[1], tile_max.shape[0], max_cycles, self.template.form,
            self._state.ptr, self._log.ptr, self.command_queue.handle)
        check(rc, 'kimg_clean_cycles')
        return self._collect_cycles() if collect else None

    def run_major_cycles(self, psf_patch, noise_threshold, left_for_next, max_cycles):
        """The minor cycles of one major cycle in one call (frontend.py:560-585; include/kimg.h:
        kimg_clean_major_cycles): the first cycle runs without a threshold, the others stop below
        ``max(noise_threshold, left_for_next * power of the first peak)`` -- worked out on the device,
        in the host's arithmetic -- or do not run if the first peak is not above that.  Only
        enqueues: read the results with :meth:`_collect_cycle_arrays` (the first row is the first
        cycle's).  Returns (cycles done, metric of the first one) -- which the call has without
        reading the device back; the read-back of state and log is started on a stream of its own, so
        that whatever the caller enqueues next does not wait for it -- or False (having done nothing)
        where the form that can do this does not run: the caller then takes the reference's two
        steps."""
        import ctypes
        self.ensure_all_bound()
        if max_cycles <= 0 or (self.template.form & 0xff) not in (CLEAN_FORMS['auto'], CLEAN_FORMS['multi']):
            return False
        dirty, psf = self.buffer('dirty'), self.buffer('psf')
        P, H, W = dirty.shape
        cp = self.template.clean_parameters
        self._ensure_log(max_cycles)
        tile_max = self.buffer('tile_max')
        done, first = ctypes.c_int(0), ctypes.c_float(0.0)
        rc = lib().kimg_clean_major_cycles(
            dirty.ptr, self.buffer('model').ptr, W, H * W, W, H, P,
            psf.ptr, psf.shape[2], psf.shape[1] * psf.shape[2], psf.shape[2], psf.shape[1],
            psf_patch[2], psf_patch[1], self._update_tiles.border_pixels, cp.mode,
            cp.loop_gain, float(noise_threshold), float(left_for_next), tile_max.ptr,
            self.buffer('tile_pos').ptr, tile_max.shape[1], tile_max.shape[0], max_cycles,
            self.template.form, self._state.ptr, self._log.ptr, self.command_queue.handle,
            ctypes.byref(done), ctypes.byref(first))
        if rc == -10001:            # KIMG_EUNSUPPORTED
            return False
        check(rc, 'kimg_clean_major_cycles')
        self._start_read_back()
        return int(done.value), float(first.value)

    def _start_read_back(self):
        """State head and log on their way to pinned host memory, on a stream of their own behind
        what the queue holds now (nothing of the caller's next stage waits for the copy, and the
        copy does not wait for that stage)."""
        import torch
        q = self.command_queue
        self._state.used_on(q)
        self._log.used_on(q)
        rows = min(self._log_rows, self._log.shape[0])
        n = 4 + rows * self._log.shape[1]
        if self._read_stream is None:
            self._read_stream = torch.cuda.Stream(device=q.stream.device)
        if self._read_host is None or self._read_host.numel() < n:
            self._read_host = torch.empty(n, dtype=torch.float32, pin_memory=True)
        ready = torch.cuda.Event()
        ready.record(q.stream)
        with torch.cuda.stream(self._read_stream):
            self._read_stream.wait_event(ready)
            head = self._state.tensor.reshape(-1)[:4]
            if head.dtype != torch.float32:
                head = head.view(torch.float32)
            self._read_host[:4].copy_(head, non_blocking=True)
            self._read_host[4:n].copy_(self._log.tensor[:rows].reshape(-1), non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self._read_stream)
        # (the queue's next work on these buffers comes after the copy: _finish_read_back)
        self._read_back = (copied, rows)

    def _finish_read_back(self):
        """Wait for a read-back in flight (before the state or the log are written again) and keep
        what it brought."""
        if self._read_back is not None:
            copied, rows = self._read_back
            self._read_back = None
            copied.synchronize()
            both = self._read_host[:4 + rows * self._log.shape[1]].numpy().copy()
            self._read_result = (both, rows)

    def _ensure_log(self, max_cycles):
        self._finish_read_back()
        self._read_result = None        # (of the loop before this one)
        P = self.buffer('dirty').shape[0]
        if self._log is None or self._log.shape[0] < max_cycles:
            self._log = accel.DeviceArray(self.command_queue.context, (max_cycles, 3 + P),
                                          np.float32, queue=self.command_queue)
        self._log_rows = max_cycles         # rows the coming call can write (the read-back's size)

    def _collect_cycle_arrays(self):
        """Read back what the device-resident loop logged (synchronises with the queue):
        (peak metrics float32 [n], positions int32 [n][2] as (y, x), model pixels float32 [n][P])."""
        # (only the head of the state buffer: it also holds the persistent form's per-workgroup
        # replicas, tens of megabytes)
        # ... and state and log in ONE read-back (each is a host round trip)
        import torch
        self._finish_read_back()
        if self._read_result is not None:
            both, rows = self._read_result      # (run_major_cycles started the read-back itself)
        else:
            q = self.command_queue
            self._state.used_on(q)
            self._log.used_on(q)
            with torch.cuda.stream(q.stream):
                head = self._state.tensor.reshape(-1)[:4]
                if head.dtype != torch.float32:
                    head = head.view(torch.float32)
                rows = min(self._log_rows, self._log.shape[0])
                both = torch.cat([head, self._log.tensor[:rows].reshape(-1)]).cpu().numpy()
        state = both[:4].view(np.int32)
        if int(state[1]) == 2:
            check(-10004, 'kimg_clean_cycles')       # KIMG_ETIMEOUT: the persistent loop gave up
        count = int(state[0])
        log = both[4:].reshape(rows, self._log.shape[1])[:count]
        return log[:, 0].copy(), log[:, 1:3].copy().view(np.int32), log[:, 3:].copy()

    def last_launches(self):
        """Launches the last :meth:`run_cycles` took if it ran in the multi-component form
        (KIMG_CLEAN_FORM_MULTI: 1 to 8 components per launch), else None.  Synchronises."""
        import torch
        q = self.command_queue
        self._state.used_on(q)
        with torch.cuda.stream(q.stream):
            head = self._state.tensor.reshape(-1)[:1660].cpu().numpy().view(np.int32)
        if head[4] != 0x4d554c54:       # mc_scratch.pad[0], set by mc_init_kernel
            return None
        # mc_state.launches of the two state buffers (byte offsets 64 + 28 and 64 + 6464 + 28)
        return int(max(head[16 + 7], head[16 + 1616 + 7]))

    def _collect_cycles(self):
        """The same as a list of (peak_value, (y, x), model_pixel), the reference's per-cycle
        results (clean.py:848-891).  (Built without a Python-level loop over numpy scalars: at 1000
        cycles per call that loop cost 0.7 ms, a tenth of the device time of the cycles.)"""
        values, pos, pix = self._collect_cycle_arrays()
        return list(zip(values, [tuple(p) for p in pos.tolist()], pix))

    def _batch_key(self):
        """Cleans with equal keys may share the launches of :func:`run_cycles_batch`."""
        dirty, psf = self.buffer('dirty'), self.buffer('psf')
        cp = self.template.clean_parameters
        return (dirty.shape, psf.shape, self.buffer('tile_max').shape,
                self._update_tiles.border_pixels, cp.mode, float(cp.loop_gain),
                str(self.command_queue.context.device))


def batch_supported(clean, psf_patch):
    """Can this patch take the one-launch-per-cycle form that :func:`run_cycles_batch` needs?
    (kimg_clean_cycles_batch: at most 32 x 32 lattice blocks, all of them plus one bookkeeping
    row resident at once.)"""
    bx = accel.divup(psf_patch[2], TILE) + 1
    by = accel.divup(psf_patch[1], TILE) + 1
    return bx <= 32 and by <= 32 and bx * (by + 1) <= 256


def multi_components(clean, psf_patch):
    """Components per launch of the multi-component form of the device-resident loop
    (KIMG_CLEAN_FORM_MULTI, csrc/clean_multi.hip: kimg_clean_multi_components) for this patch:
    256 record slots per launch, a power of two (at least 16) of them per lattice; 0 = not available."""
    bx = accel.divup(psf_patch[2], TILE) + 1
    by = accel.divup(psf_patch[1], TILE) + 1
    tiles = clean.buffer('tile_max').shape
    if bx * by > 256 or max(tiles) > 2047 or tiles[0] * tiles[1] < 4:
        return 0
    seg = 16
    while seg < bx * by:
        seg *= 2
    return min(8, 256 // seg)


def prefers_solo(clean, psf_patch, max_cycles):
    """Is this channel's loop better run on its own than in a batch?  Yes where a launch can plan
    several components (what ``auto`` then does: kimg_clean_cycles): several components per
    launch beat one per channel and launch, and the launches of that form wait for nobody, so the
    channels' loops share the device freely."""
    template = getattr(clean, 'template', None)
    if template is None or (template.form & 0xff) != CLEAN_FORMS['auto'] or max_cycles < 4:
        return False
    return multi_components(clean, psf_patch) >= 2


def enqueue_cycles_batch(cleans, psf_patches, thresholds, max_cycles, command_queue=None):
    """Enqueue the minor-cycle loops of several channels as ONE sequence of launches
    (``kimg_clean_cycles_batch``: cycle i of every channel shares a launch, so the kernel boundary
    that sets the pace of a cycle is paid once for all of them).  ``cleans``: up to 8
    :class:`Clean` operations with images of the same shape and the same CLEAN parameters, each
    bound to its own buffers; ``psf_patches``, ``thresholds``, ``max_cycles``: one entry per
    channel.  The launches go to ``command_queue`` (default: the first channel's), which first waits
    (on the device) for the work already enqueued on the other channels' queues.  Returns that
    queue: the caller must ``finish()`` it before any other queue touches the channels' buffers
    again.  (The other queues are deliberately NOT made to wait on the device for the batch: a
    hardware queue that holds an unsatisfied barrier while the batch runs doubles the time between
    the batch's dependent launches -- measured 14.4 against 6.6 ms per 1000 cycles of 4 channels.)
    Read the results with ``Clean._collect_cycles`` (or use :func:`run_cycles_batch`)."""
    from . import _lib
    if not 1 <= len(cleans) <= _lib.CLEAN_BATCH_MAX:
        raise ValueError('1 to {} channels per batch'.format(_lib.CLEAN_BATCH_MAX))
    if len({id(c) for c in cleans}) != len(cleans):
        raise ValueError('a channel appears twice in the batch')
    if len({c._batch_key() for c in cleans}) != 1:
        raise ValueError('the channels of a batch need images of the same shape and the same '
                         'CLEAN parameters')
    q = command_queue if command_queue is not None else cleans[0].command_queue
    torch = accel._torch()
    table = (_lib.CleanChannel * len(cleans))()
    for c, ch, patch, threshold, cycles in zip(cleans, table, psf_patches, thresholds, max_cycles):
        c.ensure_all_bound()
        if cycles <= 0:
            raise ValueError('max_cycles must be positive')
        c._ensure_log(cycles)
        ch.dirty, ch.model, ch.psf = c.buffer('dirty').ptr, c.buffer('model').ptr, c.buffer('psf').ptr
        ch.tile_max, ch.tile_pos = c.buffer('tile_max').ptr, c.buffer('tile_pos').ptr
        ch.state, ch.log = c._state.ptr, c._log.ptr
        ch.patch_width, ch.patch_height = int(patch[2]), int(patch[1])
        ch.threshold, ch.max_cycles = float(threshold), int(cycles)
        if c.command_queue is not q:
            ev = torch.cuda.Event()
            ev.record(c.command_queue.stream)
            q.stream.wait_event(ev)
    first = cleans[0]
    dirty, psf, tile_max = first.buffer('dirty'), first.buffer('psf'), first.buffer('tile_max')
    P, H, W = dirty.shape
    cp = first.template.clean_parameters
    rc = lib().kimg_clean_cycles_batch(
        table, len(cleans), W, H * W, W, H, P, psf.shape[2], psf.shape[1] * psf.shape[2],
        psf.shape[2], psf.shape[1], first._update_tiles.border_pixels, cp.mode, cp.loop_gain,
        tile_max.shape[1], tile_max.shape[0], q.handle)
    check(rc, 'kimg_clean_cycles_batch')
    return q


def run_cycles_batch(cleans, psf_patches, thresholds, max_cycles, command_queue=None):
    """:func:`enqueue_cycles_batch`, a host wait for the batch, and the read-back: a list, per
    channel, of what ``Clean.run_cycles`` returns -- and exactly what it would have returned for
    that channel alone."""
    enqueue_cycles_batch(cleans, psf_patches, thresholds, max_cycles, command_queue).finish()
    return [c._collect_cycles() for c in cleans]


class CleanBatcher:
    """Rendezvous of the channels a process images concurrently (``frontend.process_channels``:
    one host thread and one HIP stream per channel in flight): a thread that reaches the minor
    cycles of its channel hands its :class:`Clean` in and waits until the other channels in flight
    have reached theirs (or ``timeout`` seconds have passed: nobody waits for a channel that
    has no cycles to run in this major cycle); the channels present are then CLEANed together with
    :func:`enqueue_cycles_batch`, each thread reads its own results back.  ``parties`` = channels in
    flight; a thread that will not come back calls :meth:`leave`.

    ``phased``: the threads also pass their throughput-bound stages (weights, gridding, FFTs,
    prediction) through :meth:`device_phase`, one channel at a time.  Channels that all start at
    once otherwise stay in step for good -- they share the device while gridding, reach CLEAN
    together, and the device idles through the latency-bound cycles of all of them; taking turns
    puts one channel's gridding next to the others' cycles.  The rendezvous then waits only for
    the threads that are on their way from a device phase to their cycles (not for one that is
    gridding, loading or CLEANing), and with ``overlap`` a batch does not wait for the batch
    before it either (each runs on the stream of its first channel).
    """

    def __init__(self, parties, timeout=0.02, phased=False, overlap=True, phase_permits=1):
        import threading
        self._cond = threading.Condition()
        self._parties = int(parties)
        self._timeout = float(timeout)
        self._waiting = []
        self.phased = bool(phased)
        self._overlap = bool(overlap)
        self._phase_lock = threading.Semaphore(max(1, int(phase_permits)))    # channels inside a device phase
        self._expected = set()      # threads between the end of a device phase and their cycles
        self._cleaning = 0          # channels whose cycles are running
        #: (number of channels, cycles asked for) of every launch sequence so far, for tests / reports
        self.batches = []

    def leave(self):
        import threading
        with self._cond:
            self._parties -= 1
            self._expected.discard(threading.get_ident())
            self._cond.notify_all()

    def idle(self):
        """The calling thread is not on its way to its cycles (its channel is finished)."""
        import threading
        with self._cond:
            self._expected.discard(threading.get_ident())
            self._cond.notify_all()

    def device_phase(self):
        """Context manager around a throughput-bound stage: one channel at a time."""
        import contextlib
        import threading

        @contextlib.contextmanager
        def phase():
            me = threading.get_ident()
            with self._cond:
                self._expected.discard(me)
                self._cond.notify_all()     # (nobody waits for a thread that is gridding)
            with self._phase_lock:
                try:
                    yield
                finally:
                    with self._cond:
                        self._expected.add(me)
        return phase()

    def _may_launch(self):
        """With the lock held: is there nobody left to wait for?"""
        if self.phased:
            return not self._expected and (self._overlap or not self._cleaning)
        return len(self._waiting) >= self._parties

    def _launch(self):
        """With the lock held (released while the device works): run everything that is waiting."""
        entries, self._waiting = self._waiting, []
        self._cleaning += len(entries)
        groups = {}
        for e in entries:
            if prefers_solo(e['clean'], e['patch'], e['max_cycles']):
                e['solo'] = True
            elif batch_supported(e['clean'], e['patch']):
                groups.setdefault(e['clean']._batch_key(), []).append(e)
            else:
                e['solo'] = True
        from . import _lib
        parts = []
        for group in groups.values():
            for i in range(0, len(group), _lib.CLEAN_BATCH_MAX):
                part = group[i:i + _lib.CLEAN_BATCH_MAX]
                if len(part) == 1:
                    part[0]['solo'] = True
                else:
                    parts.append(part)
        for e in entries:
            if e['solo']:
                e['ready'] = True       # (runs in its own thread, next to the batches)
        self._cond.notify_all()
        self._cond.release()
        finished = []
        try:
            for part in parts:
                try:
                    # (the host waits for the batch here, with its threads parked: see
                    # enqueue_cycles_batch for why their queues are not made to wait instead)
                    enqueue_cycles_batch([e['clean'] for e in part], [e['patch'] for e in part],
                                         [e['threshold'] for e in part],
                                         [e['max_cycles'] for e in part],
                                         part[0]['clean'].command_queue).finish()
                    part[0]['batch'] = (len(part), max(e['max_cycles'] for e in part))
                except Exception as exc:        # noqa: B902 -- handed to the threads concerned
                    for e in part:
                        e['error'] = exc
                finished.append(part)
        finally:
            # (also when the leader is interrupted -- KeyboardInterrupt, SystemExit: the threads
            # whose entries it took over are no longer waiting in the list and would never be
            # woken; they get an error instead and the count of running channels stays right)
            self._cond.acquire()
            for part in parts:
                if not any(part is f for f in finished):
                    for e in part:
                        e['error'] = e['error'] or RuntimeError('the CLEAN batch was interrupted')
                if 'batch' in part[0]:
                    self.batches.append(part[0]['batch'])
                self._cleaning -= len(part)
                for e in part:
                    e['ready'] = True
            self._cond.notify_all()

    def run_major_cycles(self, clean, psf_patch, noise_threshold, left_for_next, max_cycles):
        """``clean.run_major_cycles(...)`` (the minor cycles of a major cycle in one call, first
        cycle included) for a channel whose cycles run on their own anyway (:func:`prefers_solo`):
        nobody waits for it, and it waits for nobody.  Returns what that returns -- (cycles done,
        metric of the first) -- or None if the channel's cycles are of the kind that shares launches
        (the caller then runs the first cycle itself and comes back with :meth:`run_cycles`)."""
        import threading
        if max_cycles <= 0 or not prefers_solo(clean, psf_patch, max_cycles):
            return None
        with self._cond:
            self._expected.discard(threading.get_ident())
            self._cleaning += 1
            self._cond.notify_all()
        try:
            return clean.run_major_cycles(psf_patch, noise_threshold, left_for_next, max_cycles) or None
        finally:
            with self._cond:
                self._cleaning -= 1
                self._cond.notify_all()

    def run_cycles(self, clean, psf_patch, threshold, max_cycles, arrays=False):
        """``clean.run_cycles(psf_patch, threshold, max_cycles)``, sharing its launches with the
        other channels in flight.  ``arrays``: return ``Clean._collect_cycle_arrays()`` instead of
        the list of per-cycle tuples."""
        import time
        if max_cycles <= 0:
            return (np.zeros(0, np.float32), np.zeros((0, 2), np.int32),
                    np.zeros((0, clean.buffer('dirty').shape[0]), np.float32)) if arrays else []
        import threading
        entry = dict(clean=clean, patch=psf_patch, threshold=threshold, max_cycles=max_cycles,
                     ready=False, solo=False, error=None)
        with self._cond:
            self._expected.discard(threading.get_ident())
            self._waiting.append(entry)
            self._cond.notify_all()
            deadline = time.monotonic() + self._timeout
            while not entry['ready']:
                if self._waiting and self._waiting[0] is entry:
                    # the longest waiter leads: when nobody is left to wait for, or after
                    # `timeout` without the others
                    remaining = deadline - time.monotonic()
                    if self._may_launch() or remaining <= 0:
                        self._launch()
                        continue
                else:
                    remaining = self._timeout
                self._cond.wait(remaining)
        if entry['error'] is not None:
            raise entry['error']
        if entry['solo']:
            try:
                clean.run_cycles(psf_patch, threshold, max_cycles, collect=False)
            finally:
                with self._cond:
                    self._cleaning -= 1
                    self._cond.notify_all()
        return clean._collect_cycle_arrays() if arrays else clean._collect_cycles()
