"""The numerical part of the per-channel driver: imaging weights, PSF, and the major /
minor cycle loop, reading visibilities from a (device-resident) reader.

Mirrors ``make_weights`` (frontend.py:86-106), ``make_dirty`` (:110-142) and the loop of
``process_channel`` (:465-585) of the reference.  Everything the reference's driver does around
that loop — dataset loading, FITS output, restoring beam, progress bars, statistics — is
outside the hot path and is not reproduced.

A reader with ``iter_slice_device`` (``preprocess.VisibilityReaderDevice``) is consumed
zero-copy: the visibilities stay in HBM for all major cycles.  Any other reader with the
reference's ``iter_slice`` works through the façade's host setters.
"""
import numpy as np

from . import clean, trace, weight


def _device_reader(reader):
    return hasattr(reader, 'iter_slice_device')


def make_weights(reader, rel_channel, imager, weight_type, vis_block):
    """frontend.py:86-106.  Returns (noise, normalized_noise) of ``finalize_weights``."""
    imager.clear_weights()
    if weight_type != weight.WeightType.NATURAL:
        for w_slice in range(reader.num_w_slices(rel_channel)):
            if _device_reader(reader):
                for chunk in reader.iter_slice_device(rel_channel, w_slice, vis_block):
                    imager.grid_weights_device(chunk)
            else:
                for chunk in reader.iter_slice(rel_channel, w_slice, vis_block):
                    imager.grid_weights(chunk.uv, chunk.weights)
    return imager.finalize_weights()


def make_dirty(reader, rel_channel, field, imager, mid_w, vis_block, degrid,
               full_cycle=False, subtract_model=False):
    """frontend.py:110-142: grid (after optional model subtraction) every W-slice of the
    channel and accumulate the slices' images into ``dirty``."""
    imager.clear_dirty()
    if full_cycle and not degrid:
        imager.model_to_predict()
    for w_slice in range(reader.num_w_slices(rel_channel)):
        if reader.len(rel_channel, w_slice) == 0:
            continue
        if full_cycle and degrid:
            imager.model_to_grid(mid_w[w_slice])
        imager.clear_grid()
        if _device_reader(reader):
            for chunk in reader.iter_slice_device(rel_channel, w_slice, vis_block):
                imager.set_chunk_device(chunk, field)
                if subtract_model:
                    imager.continuum_predict(mid_w[w_slice])
                if full_cycle:
                    imager.predict(mid_w[w_slice])
                imager.grid()
        else:
            for chunk in reader.iter_slice(rel_channel, w_slice, vis_block):
                imager.num_vis = len(chunk.uv)
                imager.set_coordinates(chunk)
                v = np.ascontiguousarray(chunk[field])
                imager.set_vis(v.astype(np.complex64) if field == 'weights' else v)
                if full_cycle or subtract_model:
                    imager.set_weights(np.ascontiguousarray(chunk.weights))
                if subtract_model:
                    imager.continuum_predict(mid_w[w_slice])
                if full_cycle:
                    imager.predict(mid_w[w_slice])
                imager.grid()
        imager.grid_to_image(mid_w[w_slice])


def slice_mid_w(image_p, grid_p):
    """Central w (in wavelengths) of each W-slice, frontend.py:508-509."""
    slice_w_step = float(grid_p.fixed.max_w / image_p.wavelength / (grid_p.w_slices - 0.5))
    return np.arange(grid_p.w_slices) * slice_w_step


def process_channel(reader, rel_channel, imager, image_p, grid_p, clean_p, weight_type,
                    vis_block, major, degrid, subtract_model=False, batched_clean=True,
                    fit_beam=False, clean_batcher=None):
    """The loop of frontend.process_channel (frontend.py:497-585) from "Compute imaging
    weights" to the end of the last major cycle.

    Returns a dict: ``weights_noise``, ``normalized_noise``, ``psf_patch``, ``scale``,
    ``noise`` (last estimate), ``major``, ``minor``, ``peaks`` (first peak metric of every
    major cycle), or None when the channel has no usable data (frontend.py:524-527).  The
    images stay in the imager's ``dirty`` (residual), ``model`` and ``psf`` buffers.

    ``fit_beam`` adds ``restoring_beam``: the Gaussian fitted to the central PSF patch
    (frontend.py:534-535), as ``beam.restore`` takes it.

    ``batched_clean`` runs the minor cycles of one major cycle with ``Imaging.clean_cycles``
    (no host round trip per cycle); the result is identical to the per-cycle loop.
    ``clean_batcher`` (:class:`clean.CleanBatcher`, set by :func:`process_channels`) lets those
    cycles share their launches with the other channels in flight.
    """
    if not any(reader.len(rel_channel, s) for s in range(reader.num_w_slices(rel_channel))):
        return None
    _check_imager_parameters(imager, image_p, grid_p)
    import contextlib
    if clean_batcher is not None and getattr(clean_batcher, 'phased', False):
        try:
            # the channels in flight take turns at their throughput-bound stages
            return _process_channel_stages(
                reader, rel_channel, imager, image_p, grid_p, clean_p, weight_type, vis_block,
                major, degrid, subtract_model, batched_clean, fit_beam, clean_batcher,
                clean_batcher.device_phase)
        finally:
            clean_batcher.idle()        # (nobody waits for this thread until its next stage)
    return _process_channel_stages(
        reader, rel_channel, imager, image_p, grid_p, clean_p, weight_type, vis_block, major,
        degrid, subtract_model, batched_clean, fit_beam, clean_batcher, contextlib.nullcontext)


def _check_imager_parameters(imager, image_p, grid_p):
    """An imager is made for one set of image and grid parameters (the reference makes one per
    channel, frontend.py:497-520); a worker that re-uses its imager for the channels of a band may
    do so only where their parameters are the imager's (an imager made for another channel's cell
    size and wavelength images this one wrongly, by per cent, without any error)."""
    own_image = getattr(imager, 'image_parameters', None)
    own_grid = getattr(imager, 'grid_parameters', None)
    if own_image is None or own_grid is None:
        return
    for name in ('pixels', 'pixel_size', 'cell_size', 'wavelength'):
        a, b = getattr(own_image, name, None), getattr(image_p, name, None)
        if a is not None and b is not None and float(a) != float(b):
            raise ValueError('the imager was made for {} = {!r}, the channel has {!r}'.format(name, a, b))
    for name in ('w_slices', 'w_planes'):
        a, b = getattr(own_grid, name, None), getattr(grid_p, name, None)
        if a is not None and b is not None and a != b:
            raise ValueError('the imager was made for {} = {!r}, the channel has {!r}'.format(name, a, b))
    a, b = getattr(own_grid, 'fixed', None), getattr(grid_p, 'fixed', None)
    if a is not None and b is not None and getattr(a, 'max_w', None) is not None \
            and getattr(b, 'max_w', None) is not None and float(a.max_w) != float(b.max_w):
        raise ValueError('the imager was made for max_w = {!r}, the channel has {!r}'.format(a.max_w, b.max_w))


def _process_channel_stages(reader, rel_channel, imager, image_p, grid_p, clean_p, weight_type,
                            vis_block, major, degrid, subtract_model, batched_clean, fit_beam,
                            clean_batcher, device_phase):
    num_pols = len(image_p.fixed.polarizations)
    with device_phase():
        imager.clear_model()
        with trace.range('make_weights'):
            weights_noise, normalized_noise = make_weights(reader, rel_channel, imager,
                                                           weight_type, vis_block)
        mid_w = slice_mid_w(image_p, grid_p)
        with trace.range('make_psf'):
            make_dirty(reader, rel_channel, 'weights', imager, mid_w, vis_block, degrid)
        started = None
        deferred = (not fit_beam and major > 0 and getattr(imager, 'device_psf_stage', False) is True)
        if deferred:
            # the PSF's scaling and patch without the host in between: the reciprocal of the
            # central pixel stays on the device, the patch's bounds and the factors come back on
            # a stream of their own while the first dirty image is gridded; a channel without
            # data (central pixel 0) is found out when they are looked at, after that gridding
            with trace.range('psf_patch'):
                imager.scale_dirty_by_centre()
                imager.dirty_to_psf()
                started = imager.psf_patch_start()
        else:
            with trace.range('psf_patch'):
                dirty = imager.buffer('dirty')
                centre = dirty.shape[1] // 2
                psf_peak = np.zeros((dirty.shape[0],), dirty.dtype)
                # (this read is where the PSF's gridding and transform are waited for: the turn at
                # the device ends here; what follows is small launches and host round trips, which
                # may run next to another channel's gridding)
                dirty.get_region(imager.command_queue, psf_peak, np.s_[:, centre, centre], np.s_[:])
    out = dict(weights_noise=weights_noise, normalized_noise=normalized_noise, major=0, minor=0,
               peaks=[], noise=None)
    if not deferred:
        with trace.range('psf_patch'):
            if np.any(psf_peak == 0):
                return None
            scale = np.reciprocal(psf_peak)
            imager.scale_dirty(scale)
            imager.dirty_to_psf()
            psf_patch = imager.psf_patch()
        out.update(psf_patch=tuple(int(x) for x in psf_patch), scale=scale)
        if fit_beam:
            from . import beam
            psf_core = extract_psf(imager.command_queue, imager.buffer('psf'), psf_patch[1:])
            out['restoring_beam'] = beam.fit_beam(psf_core)
    for i in range(major):
        with device_phase():
            with trace.range('make_dirty[%d]' % i):
                make_dirty(reader, rel_channel, 'vis', imager, mid_w, vis_block, degrid,
                           i != 0, subtract_model)
            if deferred and 'scale' not in out:
                imager.scale_dirty_by_kept()
            else:
                imager.scale_dirty(scale)
            out['major'] += 1
            with trace.range('noise_est'):
                noise = imager.noise_est()      # (waits for the gridding: the turn ends here)
            out['noise'] = noise
        if started is not None:
            psf_patch, scale = imager.psf_patch_finish(started)        # (arrived long ago)
            started = None
            if np.any(np.isinf(scale)):
                return None             # (a central pixel of 0: frontend.py:543-544)
            out.update(psf_patch=tuple(int(x) for x in psf_patch), scale=scale)
        noise_threshold = noise * clean.noise_threshold_scale(clean_p.mode, clean_p.threshold,
                                                              num_pols)
        values = None
        if batched_clean and getattr(imager, 'one_call_major_cycles', False) is True \
                and noise_threshold == noise_threshold:         # (not NaN)
            # the first cycle and the others in one call: the threshold follows from the first
            # peak on the device, in this arithmetic (no host round trip in between)
            with trace.range('clean[%d]' % i):
                imager.clean_reset()
                if clean_batcher is not None:
                    values = imager.clean_major_cycles(psf_patch, noise_threshold,
                                                       1.0 - clean_p.major_gain, clean_p.minor,
                                                       batcher=clean_batcher)
                else:
                    values = imager.clean_major_cycles(psf_patch, noise_threshold,
                                                       1.0 - clean_p.major_gain, clean_p.minor)
        rest = None                     # cycles after the first, where one call ran them all
        if values:
            peak_value, cycles = values
            rest = cycles - 1
            values = None
        if rest is None:
            with trace.range('first_cycle'):
                imager.clean_reset()
                peak_value = imager.clean_cycle(psf_patch)
        out['peaks'].append(peak_value)
        peak_power = clean.metric_to_power(clean_p.mode, peak_value)
        mgain_threshold = (1.0 - clean_p.major_gain) * peak_power
        threshold = max(noise_threshold, mgain_threshold)
        if peak_power <= threshold:
            break
        threshold_metric = clean.power_to_metric(clean_p.mode, threshold)
        if rest is not None:
            # the reference counts the cycle that found the peak below threshold too (:579-582)
            out['minor'] += rest + (1 if rest < clean_p.minor - 1 else 0)
        elif batched_clean:
            with trace.range('clean[%d]' % i):
                if clean_batcher is not None:
                    values = imager.clean_cycles(psf_patch, threshold_metric, clean_p.minor - 1,
                                                 batcher=clean_batcher)
                else:
                    values = imager.clean_cycles(psf_patch, threshold_metric, clean_p.minor - 1)
            # the reference counts the cycle that found the peak below threshold too (:579-582)
            out['minor'] += len(values) + (1 if len(values) < clean_p.minor - 1 else 0)
        else:
            for _ in range(clean_p.minor - 1):
                value = imager.clean_cycle(psf_patch, threshold_metric)
                out['minor'] += 1
                if value is None:
                    break
        if i == major - 1:
            with trace.range('noise_est'):
                out['noise'] = imager.noise_est()
    return out


def extract_psf(queue, psf, psf_patch):
    """Central ``psf_patch`` = (height, width) region of the first polarization of a device
    PSF, as a host array (frontend.py:146-168; what the reference hands to ``fit_beam``)."""
    y0 = (psf.shape[1] - psf_patch[0]) // 2
    x0 = (psf.shape[2] - psf_patch[1]) // 2
    out = np.empty((psf_patch[0], psf_patch[1]), psf.dtype)
    psf.get_region(queue, out, np.s_[0, y0:y0 + psf_patch[0], x0:x0 + psf_patch[1]], np.s_[:, :])
    return out


def find_peak(queue, image, pbeam, noise):
    """frontend.find_peak (frontend.py:171-194) on device arrays: the largest |pixel| among those
    with |pixel| * pbeam > 7.5 * noise, NaN if there is none.  ``pbeam`` is a device array of
    shape (height, width) or None (no primary-beam correction)."""
    from . import accel
    from ._lib import lib, check
    P, H, W = image.shape
    out = accel.DeviceArray(queue.context, (1,), np.float32, queue=queue)
    check(lib().kimg_image_peak(image.ptr, W, H * W, pbeam.ptr if pbeam is not None else None, W,
                                W, H, P, float(noise), out.ptr, queue.handle), 'kimg_image_peak')
    peak = float(out.get(queue)[0])
    return peak if peak != 0.0 else float('nan')


def get_totals(queue, image, restoring_beam):
    """frontend.get_totals (frontend.py:197-209): total flux density per polarization (a list in
    polarization order) = NaN-ignoring pixel sum / area under the restoring beam in pixels."""
    import math
    from . import accel
    from ._lib import lib, check
    P, H, W = image.shape
    sums = accel.DeviceArray(queue.context, (P,), np.float64, queue=queue)
    check(lib().kimg_image_nansum(image.ptr, W, H * W, W, H, P, sums.ptr, queue.handle),
          'kimg_image_nansum')
    beam_area = 2 * math.pi * restoring_beam.major * restoring_beam.minor / (8 * math.log(2))
    return [float(x) / beam_area for x in sums.get(queue)]


def process_channels(jobs, workers=4, batch_clean=True, stagger=None):
    """Image several channels of one GPU concurrently, one host thread and one HIP stream
    (command queue) per channel in flight.

    The reference loops over channels serially (frontend.py:749-767).  Within a channel the
    stages are dependent, and the CLEAN minor cycles are a latency-bound chain of small launches
    that leaves most of the device idle; running a second channel's gridding / FFTs next to it on
    another stream fills those gaps, and with ``batch_clean`` the minor cycles of the channels in
    flight share their launches (``clean.CleanBatcher``: cycle i of every channel in ONE launch;
    results identical).  ``jobs`` is a list of dicts of :func:`process_channel` keyword arguments
    (each with its own ``imager``, hence its own command queue; they may share a reader).  Returns
    the list of results in job order.  For more channels than fit in memory at once see
    :func:`process_channel_stream`.
    """
    if workers <= 1 or len(jobs) <= 1:
        process_channel_stream.last_batches = []
        return [process_channel(**job) for job in jobs]
    queues = {id(job['imager'].command_queue) for job in jobs}
    if len(queues) != len(jobs):
        raise ValueError('concurrent channels need one command queue each')
    return process_channel_stream(lambda index: jobs[index], range(len(jobs)), workers=workers,
                                  batch_clean=batch_clean, stagger=stagger)


#: CUs the window kernels fill while several channels share the device (of 256)
WINDOW_CUS_SHARED = 192


def process_channel_stream(make_job, channels, workers=4, batch_clean=True, stagger=None,
                           window_cus=None, device=None):
    """Image ``channels`` (any number) with at most ``workers`` of them in flight AND in memory.

    :func:`process_channels` takes ready-made jobs, i.e. one imager per channel; a band of
    hundreds of channels per GPU would not fit that way.  Here ``workers`` host threads draw
    channels from a queue; a thread makes the job of its next channel only when it is free to
    image it (``make_job(channel)`` or ``make_job(channel, worker)``, see
    ``parallel.image_assigned_channels``) and drops it when the channel is done, so at most
    ``workers`` imagers exist at a time -- or exactly ``workers`` for the whole band when the
    callback re-uses one imager per ``worker`` index.  With ``batch_clean`` the minor cycles of
    the channels in flight share their launches (see :func:`process_channels`).  ``stagger``: the
    channels in flight take turns at their throughput-bound stages (``clean.CleanBatcher``
    ``phased``) so that one channel grids while the others CLEAN; None = off (it paid while the
    channels' cycles shared one-component launches; it costs now that they do not).  ``window_cus``: how many CUs the gridder and degridder of the imagers in
    flight fill while there are several (``Imaging.set_window_cus``: it travels with every
    kimg_grid / kimg_degrid call of those imagers, so two streams in one process -- on the same or
    on different GPUs -- do not see each other's setting; None = 192 of 256, so that the other
    channels' CLEAN launches find room).  ``device``: the GPU the jobs are made on (an imager built
    lazily inside ``make_job`` must find its device current in the worker thread, where HIP starts
    on device 0); None = the caller's current device.  Returns the results in ``channels`` order.
    """
    import inspect
    import queue
    import threading
    channels = list(channels)
    try:
        takes_worker = len(inspect.signature(make_job).parameters) >= 2
    except (TypeError, ValueError):
        takes_worker = False
    results = [None] * len(channels)
    errors = []
    todo = queue.Queue()
    for item in enumerate(channels):
        todo.put(item)
    count = max(1, min(int(workers), len(channels)))
    if device is None:
        try:
            import torch
            device = torch.cuda.current_device() if torch.cuda.is_available() else None
        except Exception:       # noqa: B902 -- no GPU runtime: nothing to make current
            device = None
    if window_cus is None:
        window_cus = WINDOW_CUS_SHARED
    shared_cus = int(window_cus) if count > 1 else 0
    batcher = None
    if batch_clean and count > 1:
        if stagger is None:
            # Taking turns won with 2-4 channels in flight while a channel's minor cycles were one
            # component per launch batched across the channels (round 3: 11.0 -> 10.2, 8.2 -> 7.3 ms
            # per channel).  With several components per launch the cycles run on their own, wait
            # for nobody and leave most of the device to whoever grids, and the channels do better
            # in step (round 4, 12 channels, four in flight: 4.8 ms per channel against 5.9 taking
            # turns; four channels at once: 20.2 against 23.4 ms): off unless asked for.
            stagger = False
        batcher = clean.CleanBatcher(count, phased=bool(stagger))

    def one(index, channel, worker):
        job = make_job(channel, worker) if takes_worker else make_job(channel)
        kwargs = dict(job)          # (the caller's dict is left as it is)
        if batcher is not None and kwargs.get('batched_clean', True):
            kwargs.setdefault('clean_batcher', batcher)
        imager = kwargs.get('imager')
        restore = None
        if imager is not None and hasattr(imager, 'set_window_cus'):
            # The gridder's and degridder's workgroups stay on their CUs for a whole launch and
            # leave no room for another channel's CLEAN workgroups, whose chain of launches then
            # stands still: with several channels in flight those kernels keep off a quarter of
            # the device (7.2 -> 6.1 ms per channel with four in flight).
            restore = getattr(imager, '_window_cus', 0)
            imager.set_window_cus(shared_cus)
        try:
            if imager is not None and hasattr(imager, 'command_queue'):
                import torch
                with torch.cuda.device(imager.command_queue.context.device):
                    results[index] = process_channel(**kwargs)
            else:
                results[index] = process_channel(**kwargs)
        finally:
            if restore is not None:
                imager.set_window_cus(restore)

    def work(worker):
        try:
            while not errors:
                try:
                    index, channel = todo.get_nowait()
                except queue.Empty:
                    return
                # the current HIP device is per host thread and new threads start on device 0:
                # everything of a channel, the making of its job included (an imager built lazily
                # there allocates, plans transforms and launches), runs with the stream's device
                if device is not None:
                    import torch
                    with torch.cuda.device(device):
                        one(index, channel, worker)
                else:
                    one(index, channel, worker)
        except BaseException as exc:        # noqa: B902 -- re-raised in the caller's thread
            errors.append(exc)
        finally:
            if batcher is not None:
                batcher.leave()         # the channels still in flight stop waiting for this thread

    if count == 1:
        work(0)
    else:
        # (daemon threads: an interrupt of the caller must not leave the process waiting for them)
        threads = [threading.Thread(target=work, args=(w,), name='kimg-channel-%d' % w, daemon=True)
                   for w in range(count)]
        for t in threads:
            t.start()
        try:
            for t in threads:
                t.join()
        except BaseException as exc:        # noqa: B902 -- KeyboardInterrupt in the caller
            errors.append(exc)              # (the workers draw no further channels)
            raise
    #: (channels, cycles) of every shared CLEAN launch sequence of the last call, for reports
    process_channel_stream.last_batches = list(batcher.batches) if batcher is not None else []
    if errors:
        raise errors[0]
    return results


process_channel_stream.last_batches = []
