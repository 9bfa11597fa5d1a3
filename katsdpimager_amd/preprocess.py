"""Visibility preprocessing and the device-resident visibility store.

Mirror of ``katsdpimager.preprocess`` (preprocess.py:73-420) with the C++ collector
(preprocess.cpp) replaced by HIP kernels and the HDF5 / in-memory stores replaced by a
store that keeps every preprocessed visibility of the channels being imaged in HBM:

* :class:`VisibilityCollectorDevice` — ``add(uvw, weights, vis, feed_angle1, feed_angle2,
  mueller_stokes, mueller_circular)``, ``close()``, ``reader()``, ``num_input``,
  ``num_output`` as ``VisibilityCollector`` (preprocess.py:73-156).  Inputs may be numpy
  arrays (uploaded once per call) or :class:`accel.DeviceArray`.
* :class:`VisibilityReaderDevice` — ``num_channels``, ``num_w_slices(channel)``,
  ``len(channel, w_slice)``, ``iter_slice(channel, w_slice, block_size)`` yielding host
  record arrays in the reference's store dtype (so frontend.make_weights / make_dirty run
  on it unchanged) and ``iter_slice_device`` yielding zero-copy :class:`DeviceChunk` views
  for ``Imaging.set_chunk_device`` / ``Imaging.grid_weights_device``.

Layout in HBM, per (channel, w_slice): ``uv`` int16 [N][4] = (u, v, sub_u, sub_v),
``w_plane`` int16 [N], ``weights`` float32 [N][P], ``vis`` complex64 [N][P] (22 B per
visibility at P=1, 58 B at P=4) — exactly the arrays the gridder, degridder, predictor and
weight kernels consume, so a major cycle touches no host memory.
"""
import numpy as np

from . import accel
from ._lib import lib, check


def _trace_range(name):
    from . import trace
    return trace.range(name)


def make_store_dtype(num_polarizations):
    """Record type yielded by ``iter_slice``: the fields and offsets of the reference's
    ``_make_dtype`` applied to ``vis_t<P>`` (preprocess.py:42-56, preprocess.cpp:39-52)."""
    P = num_polarizations
    return np.dtype(dict(
        names=['uv', 'sub_uv', 'w_plane', 'weights', 'vis'],
        formats=[('i2', (2,)), ('i2', (2,)), 'i2', ('f4', (P,)), ('c8', (P,))],
        offsets=[0, 4, 8, 12, 12 + 4 * P], itemsize=12 + 12 * P))


class DeviceChunk:
    """A block of one (channel, w_slice) in HBM.  The arrays are views of the store with
    ``block_size`` rows of which the first ``num_vis`` are valid."""

    def __init__(self, num_vis, uv, w_plane, weights, vis, locality=None):
        self.num_vis = num_vis
        self.uv = uv
        self.w_plane = w_plane
        self.weights = weights
        self.vis = vis
        #: what the store knows about the order of these records, for ``Gridder.locality_hint``:
        #: True = consecutive records stay close (window kernel as is), False = bin them first,
        #: None = unknown (the gridder's `auto` variant measures on every call)
        self.locality = locality

    def __len__(self):
        return self.num_vis


def reorder_device_arrays(queue, num_polarizations, num_vis, arrays, kernel_width, oversample,
                          w_planes, merge, rows=None):
    """``kimg_store_reorder`` on a dict of device arrays ``uv`` int16 [N][4], ``w_plane`` int16 [N],
    ``weights`` float32 [N][P], ``vis`` complex64 [N][P] (the first ``num_vis`` rows): returns
    (dict of new arrays with ``rows`` rows, zero beyond the data; number of records kept).
    Synchronises with ``queue`` (one 8-byte read-back: the record count)."""
    context = queue.context
    P = num_polarizations
    L = lib()
    n = int(num_vis)
    rows = n if rows is None else int(rows)
    ws_bytes = int(L.kimg_store_reorder_workspace_bytes(n))
    if ws_bytes == 0:
        raise ValueError('cannot re-order {} records'.format(n))
    ws = accel.DeviceArray(context, (ws_bytes,), np.uint8, queue=queue)
    out = dict(
        uv=accel.DeviceArray(context, (rows, 4), np.int16, queue=queue),
        w_plane=accel.DeviceArray(context, (rows,), np.int16, queue=queue),
        weights=accel.DeviceArray(context, (rows, P), np.float32, queue=queue),
        vis=accel.DeviceArray(context, (rows, P), np.complex64, queue=queue))
    count = accel.DeviceArray(context, (1,), np.int64, queue=queue)
    check(L.kimg_store_reorder(
        P, n, int(kernel_width), int(oversample), int(w_planes), 1 if merge else 0,
        arrays['uv'].ptr, arrays['w_plane'].ptr, arrays['weights'].ptr, arrays['vis'].ptr,
        out['uv'].ptr, out['w_plane'].ptr, out['weights'].ptr, out['vis'].ptr,
        count.ptr, ws.ptr, ws_bytes, queue.handle), 'kimg_store_reorder')
    kept = int(count.get(queue)[0])
    if kept < rows:
        # rows beyond the data are read (and masked) by full-block views: keep them zero
        torch = accel._torch()
        with torch.cuda.stream(queue.stream):
            for a in out.values():
                a.tensor[kept:].zero_()
        queue.finish()
    return out, kept


class _SliceStore:
    """Growable structure-of-arrays for one (channel, w_slice).  ``slack`` rows beyond the
    data are always allocated so that a full-size block view can start at any row."""

    def __init__(self, context, command_queue, num_polarizations, slack):
        self.context = context
        self.queue = command_queue
        self.P = num_polarizations
        self.slack = slack
        self.length = 0
        self.capacity = 0
        self.arrays = None
        #: True once :meth:`reorder` has put the records into the window kernels' order
        self.ordered = False
        import threading
        self._slack_lock = threading.Lock()     # (readers on several host threads: ensure_slack)

    def _allocate(self, rows):
        P = self.P
        arrays = dict(
            uv=accel.DeviceArray(self.context, (rows, 4), np.int16, queue=self.queue),
            w_plane=accel.DeviceArray(self.context, (rows,), np.int16, queue=self.queue),
            weights=accel.DeviceArray(self.context, (rows, P), np.float32, queue=self.queue),
            vis=accel.DeviceArray(self.context, (rows, P), np.complex64, queue=self.queue))
        for a in arrays.values():
            a.zero(self.queue)       # the slack rows are read (and masked) by full-block views
        return arrays

    def ensure_slack(self, slack):
        """Make full-size block views of up to ``slack`` rows possible (reallocates once).

        This is reached from whoever READS the store -- a channel's imager, on a queue and possibly
        a host thread of its own -- while the copy into the larger arrays runs on the store's queue:
        the copy is waited for here, so that the reader's stream finds the records there (the first
        channels of a band imaged concurrently otherwise gridded whatever the new arrays held:
        nothing, for a weights grid of zeros and a PSF of zeros)."""
        with self._slack_lock:
            if slack > self.slack:
                self.slack = slack
                before = self.arrays
                self.reserve(0, exact=True)
                if self.arrays is not before:
                    self.queue.finish()

    def reserve(self, extra, exact=False):
        need = self.length + extra + self.slack
        if need <= self.capacity:
            return
        rows = need if exact else max(need, 2 * self.capacity)
        arrays = self._allocate(rows)
        if self.arrays is not None and self.length:
            for name, old in self.arrays.items():
                old.copy_region(self.queue, arrays[name], np.s_[:self.length], np.s_[:self.length])
        self.arrays = arrays
        self.capacity = rows

    def append(self, src, start, count):
        """Copy ``count`` rows starting at ``start`` from the dict of arrays ``src``."""
        self.reserve(count)
        for name, dst in self.arrays.items():
            src[name].copy_region(self.queue, dst, np.s_[start:start + count],
                                  np.s_[self.length:self.length + count])
        self.length += count

    def reorder(self, kernel_width, oversample, w_planes, merge):
        """Put the slice into the order the window gridder / degridder run fastest on (strips of
        grid columns swept along v: ``kimg_store_reorder``, csrc/store.hip) and, with ``merge``,
        sum all records with equal quantised coordinates into one.  Once per channel; every later
        pass reads the result.  Returns the number of records kept."""
        if self.length == 0 or self.ordered:
            self.ordered = True
            return self.length
        n = self.length
        try:
            out, kept = reorder_device_arrays(self.queue, self.P, n, self.arrays, kernel_width,
                                              oversample, w_planes, merge, rows=n + self.slack)
        except (ValueError, MemoryError, RuntimeError) as exc:
            # The re-order needs a second copy of the slice and 28 bytes of scratch per record while
            # it runs, and kimg_store_reorder takes at most 2^31 - 1 records: a slice that does not
            # allow it stays in arrival order, as the reference's store is (results are the same,
            # the window kernels run slower on it).  Other errors are not memory's: re-raised.
            too_big = isinstance(exc, ValueError) or 'out of memory' in str(exc).lower()
            if not too_big:
                raise
            import logging
            logging.getLogger(__name__).warning(
                'slice of %d records left in arrival order (%s)', n, exc)
            accel._torch().cuda.empty_cache()
            self.ordered = False
            return n
        self.arrays = out
        self.capacity = n + self.slack
        self.length = kept
        self.ordered = True
        return kept

    def view(self, start, rows):
        out = {}
        for name, a in self.arrays.items():
            t = a.tensor[start:start + rows]
            out[name] = accel.DeviceArray(self.context, tuple(t.shape), a.dtype, tensor=t)
        return out


def _to_device(context, queue, ary, dtype):
    if ary is None:
        return None
    if isinstance(ary, accel.DeviceArray):
        if ary.dtype != np.dtype(dtype):
            raise TypeError('device input must have dtype {}'.format(np.dtype(dtype)))
        if not ary.tensor.is_contiguous():
            raise ValueError('device input must be contiguous')
        return ary
    host = np.require(np.asarray(ary), dtype, 'C')
    dev = accel.DeviceArray(context, host.shape, dtype)
    dev.set(queue, host)
    return dev


class VisibilityCollectorDevice:
    """Preprocess visibilities on the GPU and keep them there (preprocess.py:73-156,
    preprocess.cpp:390-513).

    Parameters
    ----------
    command_queue : :class:`accel.CommandQueue`
    image_parameters, grid_parameters : lists, one entry per channel
        (``cell_size``; ``fixed.max_w``, ``w_slices``, ``w_planes``, ``fixed.oversample``)
    buffer_size : int
        Visibilities converted and compressed together (merging never crosses a buffer,
        preprocess.cpp:431-509); also the largest block ``iter_slice_device`` can serve.
    """

    #: records per pass of the device kernels (a whole number of buffers; see __init__)
    MAX_BATCH = 1 << 24

    def __init__(self, command_queue, image_parameters, grid_parameters, buffer_size,
                 reorder=True, merge=True):
        if len(image_parameters) != len(grid_parameters):
            raise ValueError('Inconsistent lengths of image_parameters and grid_parameters')
        if buffer_size <= 0:
            raise ValueError('buffer_size must be positive')
        self.queue = command_queue
        self.context = command_queue.context
        self.image_parameters = list(image_parameters)
        self.grid_parameters = list(grid_parameters)
        self.buffer_size = int(buffer_size)
        self.num_polarizations = P = len(image_parameters[0].fixed.polarizations)
        for ip in image_parameters:
            if len(ip.fixed.polarizations) != P:
                raise ValueError('all channels must have the same polarizations')
        self.store_dtype = make_store_dtype(P)
        self.num_input = 0
        self.num_output = 0
        #: records held after :meth:`close` (``num_output`` counts what the reference counts: the
        #: records its compress() emits; the whole-slice merge of the re-order can only lower it)
        self.num_stored = 0
        self.reorder = bool(reorder)
        self.merge = bool(merge)
        self._closed = False
        self._stores = [
            [_SliceStore(self.context, self.queue, P, self.buffer_size)
             for _ in range(gp.w_slices)]
            for gp in self.grid_parameters]
        self._lib = lib()
        # Buffers are the reference's unit of compression (merging never crosses one,
        # preprocess.cpp:431-509); the device passes run over up to MAX_BATCH records = several
        # buffers at a time (kimg_preprocess_compress, merge_window = buffer_size: same records as
        # buffer-by-buffer calls, but the ~10 us of fixed cost of each of the seven launches is
        # paid once per batch: 9 -> 20 G inputs/s with buffers of 1 Mi)
        self._batch = 0                 # records the staging arrays hold (grown by add())
        if int(self._lib.kimg_preprocess_workspace_bytes(self.buffer_size, P)) == 0:
            raise ValueError('unsupported buffer_size / polarizations')

    def _ensure_staging(self, num_vis):
        """Staging arrays for passes of up to MAX_BATCH records, never more than the call needs."""
        buffers = max(1, min(self.MAX_BATCH // self.buffer_size, -(-num_vis // self.buffer_size)))
        B = self.buffer_size * buffers
        if B <= self._batch:
            return
        P = self.num_polarizations
        ctx = self.context
        self._key = accel.DeviceArray(ctx, (B, 6), np.int16, queue=self.queue)
        self._cw = accel.DeviceArray(ctx, (B, P), np.float32, queue=self.queue)
        self._cvis = accel.DeviceArray(ctx, (B, P), np.complex64, queue=self.queue)
        # two sets of compress outputs + slice counts: the host reads the counts of pass i
        # (needed to place its records) while the device already works on pass i + 1
        max_slices = max(gp.w_slices for gp in self.grid_parameters)
        torch = accel._torch()
        self._sets = []
        for _ in range(2):
            out = dict(
                uv=accel.DeviceArray(ctx, (B, 4), np.int16, queue=self.queue),
                w_plane=accel.DeviceArray(ctx, (B,), np.int16, queue=self.queue),
                weights=accel.DeviceArray(ctx, (B, P), np.float32, queue=self.queue),
                vis=accel.DeviceArray(ctx, (B, P), np.complex64, queue=self.queue))
            counts = accel.DeviceArray(ctx, (max_slices,), np.int64, queue=self.queue)
            host = torch.empty((max_slices,), dtype=torch.int64).pin_memory()
            self._sets.append((out, counts, host, torch.cuda.Event()))
        self._ws_bytes = int(self._lib.kimg_preprocess_workspace_bytes(B, P))
        self._ws = accel.DeviceArray(ctx, (self._ws_bytes,), np.uint8, queue=self.queue)
        self._batch = B

    @property
    def num_channels(self):
        return len(self.image_parameters)

    def _matrix(self, m, shape):
        if m is None:
            return None
        m = np.ascontiguousarray(np.asarray(m), np.complex64)
        if m.shape != shape:
            raise ValueError('matrix has shape {}, expected {}'.format(m.shape, shape))
        return m

    def add(self, uvw, weights, vis, feed_angle1, feed_angle2, mueller_stokes, mueller_circular):
        """Add N visibilities of every channel (preprocess.py:117-150).

        uvw: N x 3 float32 metres; weights: C x N x Q float32; vis: C x N x Q complex64;
        feed_angle1/2: N float32 radians or None; mueller_stokes: P x Q (no feed angles) or
        P x 4; mueller_circular: 4 x Q or None.

        Device inputs are read on this collector's command queue: whatever produced them on
        another stream must have completed (or been synchronised with) before the call.
        """
        if self._closed:
            raise RuntimeError('collector is closed')
        if (feed_angle1 is None) != (feed_angle2 is None) or \
                (feed_angle1 is None) != (mueller_circular is None):
            raise ValueError('feed angles and mueller_circular must be given together')
        P = self.num_polarizations
        d_uvw = _to_device(self.context, self.queue, uvw, np.float32)
        d_weights = _to_device(self.context, self.queue, weights, np.float32)
        d_vis = _to_device(self.context, self.queue, vis, np.complex64)
        d_fa1 = _to_device(self.context, self.queue, feed_angle1, np.float32)
        d_fa2 = _to_device(self.context, self.queue, feed_angle2, np.float32)
        if len(d_uvw.shape) != 2 or d_uvw.shape[1] != 3:
            raise ValueError('Array has incorrect size')
        N = d_uvw.shape[0]
        C = self.num_channels
        if len(d_vis.shape) != 3:
            raise ValueError('Array has incorrect number of dimensions')
        Q = d_vis.shape[2]
        if Q < 1 or Q > 4:
            raise ValueError('only 4 input polarizations are supported')
        if d_vis.shape != (C, N, Q) or d_weights.shape != (C, N, Q):
            raise ValueError('Array has incorrect size')
        if d_fa1 is not None and (d_fa1.shape != (N,) or d_fa2.shape != (N,)):
            raise ValueError('Array has incorrect size')
        stokes = self._matrix(mueller_stokes, (P, 4) if d_fa1 is not None else (P, Q))
        circular = self._matrix(mueller_circular, (4, Q))
        self._ensure_staging(N)
        stream = self.queue.handle
        f32 = np.dtype(np.float32).itemsize
        torch = accel._torch()
        pending = None
        step = 0

        def retire(p):
            """Place the records of a finished buffer (needs its slice counts on the host)."""
            stores, w_slices, (out, _, host, event) = p
            event.synchronize()
            pos = 0
            for s in range(w_slices):
                c = int(host[s])
                if c:
                    stores[s].append(out, pos, c)
                    pos += c
            self.num_output += pos

        for ch in range(C):
            gp = self.grid_parameters[ch]
            cell = float(self.image_parameters[ch].cell_size)
            for i0 in range(0, N, self._batch):
                n = min(N, i0 + self._batch) - i0
                row = (ch * N + i0) * Q
                cur = self._sets[step % 2]
                step += 1
                out, counts, host, event = cur
                check(self._lib.kimg_preprocess_convert(
                    P, Q, n, d_uvw.ptr + i0 * 3 * f32, d_weights.ptr + row * f32,
                    d_vis.ptr + row * 2 * f32,
                    d_fa1.ptr + i0 * f32 if d_fa1 is not None else None,
                    d_fa2.ptr + i0 * f32 if d_fa2 is not None else None,
                    stokes.ctypes.data, circular.ctypes.data if circular is not None else None,
                    float(gp.fixed.max_w), gp.w_slices, gp.w_planes, gp.fixed.oversample, cell,
                    self._key.ptr, self._cw.ptr, self._cvis.ptr, stream), 'kimg_preprocess_convert')
                check(self._lib.kimg_preprocess_compress(
                    P, n, gp.w_slices, self._key.ptr, self._cw.ptr, self._cvis.ptr,
                    out['uv'].ptr, out['w_plane'].ptr, out['weights'].ptr, out['vis'].ptr,
                    counts.ptr, self.buffer_size if n > self.buffer_size else 0,
                    self._ws.ptr, self._ws_bytes, stream), 'kimg_preprocess_compress')
                with torch.cuda.stream(self.queue.stream):
                    host.copy_(counts.tensor, non_blocking=True)
                    event.record(self.queue.stream)
                if pending is not None:
                    retire(pending)
                pending = (self._stores[ch], gp.w_slices, cur)
        if pending is not None:
            retire(pending)
        self.num_input += C * N
        # the staging arrays of this call may be freed once the stream has consumed them
        self.queue.finish()

    def close(self):
        """preprocess.py:152-156.  New here: with ``reorder`` (the default) every stored W-slice is
        put, once, into the order the window gridder / degridder run fastest on, and with ``merge``
        all its records with equal quantised coordinates are summed into one
        (:meth:`_SliceStore.reorder`); every later pass of the channel -- weights, PSF, image,
        degrid + regrid per major cycle -- reads that order."""
        if self._closed:
            return
        self._closed = True
        if self.reorder:
            with _trace_range('store_reorder'):
                for ch, stores in enumerate(self._stores):
                    gp = self.grid_parameters[ch]
                    kernel_width = getattr(gp.fixed, 'kernel_width', None)
                    if kernel_width is None or kernel_width > 64:
                        continue            # (no window kernel for this geometry)
                    for store in stores:
                        store.reorder(kernel_width, gp.fixed.oversample, gp.w_planes, self.merge)
        self.num_stored = sum(s.length for ch in self._stores for s in ch)

    def reader(self):
        """Only after :meth:`close` (preprocess.py:152-156)."""
        if not self._closed:
            raise RuntimeError('reader() may only be called after close()')
        return VisibilityReaderDevice(self)

    def nbytes(self):
        """HBM held by the stored visibilities (excluding slack)."""
        return sum(s.length for ch in self._stores for s in ch) * self.store_dtype.itemsize


class VisibilityReaderDevice:
    """Reader over a closed :class:`VisibilityCollectorDevice` (preprocess.py:277-420)."""

    def __init__(self, collector):
        self.collector = collector
        self._stores = collector._stores
        self.store_dtype = collector.store_dtype
        self._locality_cache = {}

    @property
    def num_channels(self):
        return len(self._stores)

    def num_w_slices(self, channel):
        return len(self._stores[channel])

    def len(self, channel, w_slice):
        return self._stores[channel][w_slice].length

    def iter_slice_device(self, channel, w_slice, block_size=None):
        """Yield :class:`DeviceChunk` views of ``block_size`` rows (the last one partly
        valid).  ``block_size`` defaults to the collector's buffer size; a larger one — up to a
        whole slice per block, i.e. one gridder launch per slice — makes the store reallocate
        once so that every block view has its full ``block_size`` rows."""
        store = self._stores[channel][w_slice]
        if block_size is None:
            block_size = self.collector.buffer_size
        if block_size <= 0:
            raise ValueError('block_size must be positive')
        if store.length == 0:
            return
        store.ensure_slack(block_size)
        locality = self._locality(channel, w_slice)
        for start in range(0, store.length, block_size):
            v = store.view(start, block_size)
            yield DeviceChunk(min(block_size, store.length - start), v['uv'], v['w_plane'],
                              v['weights'], v['vis'], locality)

    def _locality(self, channel, w_slice):
        """Measured once per stored slice (``kimg_grid_jumps`` + a 4-byte read-back): does the
        slice's order suit the window gridder as it is (grid.AUTO_JUMP_FRACTION)?"""
        key = (channel, w_slice)
        if key not in self._locality_cache:
            from . import grid
            store = self._stores[channel][w_slice]
            if store.ordered:
                self._locality_cache[key] = True        # strip order: made for the window kernels
                return True
            kernel_width = getattr(self.collector.grid_parameters[channel].fixed, 'kernel_width', None)
            if kernel_width is None or store.length < grid.AUTO_MIN_VIS:
                self._locality_cache[key] = None
            else:
                queue = self.collector.queue
                count = accel.DeviceArray(queue.context, (1,), np.uint32, queue=queue)
                check(lib().kimg_grid_jumps(store.arrays['uv'].ptr, store.length, int(kernel_width),
                                            count.ptr, queue.handle), 'kimg_grid_jumps')
                jumps = int(count.get(queue)[0])
                self._locality_cache[key] = jumps <= grid.AUTO_JUMP_FRACTION * store.length
        return self._locality_cache[key]

    def iter_slice(self, channel, w_slice, block_size=None):
        """Yield host record arrays (fields uv, sub_uv, w_plane, weights, vis) like
        VisibilityReaderMem.iter_slice (preprocess.py:398-403)."""
        store = self._stores[channel][w_slice]
        queue = self.collector.queue
        if block_size is None:
            block_size = self.collector.buffer_size
        for start in range(0, store.length, block_size):
            n = min(block_size, store.length - start)
            v = store.view(start, n)
            rec = np.rec.recarray((n,), self.store_dtype)
            uv = v['uv'].get(queue)
            rec.uv = uv[:, 0:2]
            rec.sub_uv = uv[:, 2:4]
            rec.w_plane = v['w_plane'].get(queue)
            rec.weights = v['weights'].get(queue)
            rec.vis = v['vis'].get(queue)
            yield rec

    def close(self):
        pass
