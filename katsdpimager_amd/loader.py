"""A minimal visibility loader: arrays in memory or in a ``.npz`` file, delivered
the way the reference's loaders deliver them.

The reference reads measurement sets (python-casacore) and katdal files; both are outside the hot
path and their libraries are not available here.  What the path needs from a loader is the *shape
of the stream*: ``data_iter`` yields blocks of at most ``max_chunk_vis`` visibilities (over the
selected channels), each block **sorted by baseline** with a stable sort so that the
preprocessor's adjacent-merge compression finds runs (loader_ms.py:377-467; interface
loader_core.py:149-200), as dicts with ``uvw`` [N][3] metres, ``weights`` / ``vis``
[channel][N][polarization], optional ``feed_angle1/2`` [N], ``progress`` and ``total``.
:func:`preprocess_visibilities` is the loop of frontend.preprocess_visibilities
(frontend.py:40-84) around ``VisibilityCollectorDevice.add``.

File layout of :class:`LoaderArrays.save` / ``load``: ``uvw`` f4 [R][3] (metres), ``vis`` c8
[R][C][P], ``weights`` f4 [R][C][P], ``baseline`` i4 [R] (any integer that identifies the antenna
pair), ``frequency`` f8 [C] (Hz), ``polarizations`` i4 [P] (the reference's polarization
enumeration), ``phase_centre`` f8 [2] (RA, Dec in radians), ``antenna_diameter`` f8,
``longest_baseline`` f8 (metres), and optionally ``feed_angle1`` / ``feed_angle2`` f4 [R].  Rows
are in time order, as a correlator writes them.
"""
import numpy as np

from . import parameters

_LIGHTSPEED = 299792458.0


class LoaderArrays:
    """Loader over arrays held in memory (see the module docstring for their meaning)."""

    def __init__(self, uvw, vis, weights, baseline, frequency, polarizations, phase_centre=(0.0, 0.0),
                 antenna_diameter=13.5, longest_baseline=None, feed_angle1=None, feed_angle2=None):
        self.uvw = np.ascontiguousarray(uvw, np.float32)
        self.vis = np.asarray(vis, np.complex64)
        self.weights = np.asarray(weights, np.float32)
        self.baseline = np.asarray(baseline)
        self.frequencies = np.atleast_1d(np.asarray(frequency, np.float64))
        self._polarizations = [int(p) for p in polarizations]
        self._phase_centre = (float(phase_centre[0]), float(phase_centre[1]))
        self._antenna_diameter = float(antenna_diameter)
        rows = len(self.uvw)
        if self.uvw.shape != (rows, 3) or self.vis.shape != self.weights.shape \
                or self.vis.shape != (rows, len(self.frequencies), len(self._polarizations)) \
                or self.baseline.shape != (rows,):
            raise ValueError('inconsistent array shapes')
        if longest_baseline is None:
            longest_baseline = float(np.sqrt((self.uvw.astype(np.float64) ** 2).sum(axis=1)).max()) \
                if rows else 0.0
        self._longest_baseline = float(longest_baseline)
        self.feed_angle1 = None if feed_angle1 is None else np.asarray(feed_angle1, np.float32)
        self.feed_angle2 = None if feed_angle2 is None else np.asarray(feed_angle2, np.float32)
        if (self.feed_angle1 is None) != (self.feed_angle2 is None):
            raise ValueError('feed angles come in pairs')

    # ---- persistence -------------------------------------------------------------------------
    _FIELDS = ('uvw', 'vis', 'weights', 'baseline')

    def save(self, filename):
        extra = {}
        if self.feed_angle1 is not None:
            extra = dict(feed_angle1=self.feed_angle1, feed_angle2=self.feed_angle2)
        np.savez(filename, uvw=self.uvw, vis=self.vis, weights=self.weights, baseline=self.baseline,
                 frequency=self.frequencies, polarizations=np.array(self._polarizations, np.int32),
                 phase_centre=np.array(self._phase_centre), antenna_diameter=self._antenna_diameter,
                 longest_baseline=self._longest_baseline, **extra)

    @classmethod
    def load(cls, filename):
        with np.load(filename) as f:
            return cls(f['uvw'], f['vis'], f['weights'], f['baseline'], f['frequency'],
                       f['polarizations'], f['phase_centre'], float(f['antenna_diameter']),
                       float(f['longest_baseline']),
                       f['feed_angle1'] if 'feed_angle1' in f.files else None,
                       f['feed_angle2'] if 'feed_angle2' in f.files else None)

    @classmethod
    def match(cls, filename):
        """loader_core.py:32"""
        return filename.lower().endswith('.npz')

    # ---- the LoaderBase interface the path uses (loader_core.py:36-239) ------------------------
    def antenna_diameter(self):
        return self._antenna_diameter

    def longest_baseline(self):
        return self._longest_baseline

    def array_parameters(self):
        return parameters.ArrayParameters(self._antenna_diameter, self._longest_baseline)

    def num_channels(self):
        return len(self.frequencies)

    def frequency(self, channel):
        return float(self.frequencies[channel])

    def wavelength(self, channel):
        return _LIGHTSPEED / self.frequency(channel)

    def phase_centre(self):
        return self._phase_centre

    def polarizations(self):
        return list(self._polarizations)

    def has_feed_angles(self):
        return self.feed_angle1 is not None

    def channel_enabled(self, channel):
        return True

    def data_iter(self, start_channel, stop_channel, max_chunk_vis=None):
        """loader_core.py:149-200 with the block shaping of loader_ms.py:377-467: at most
        ``max_chunk_vis`` visibilities (rows x channels) per block, rows of a block in a stable
        sort by baseline, channel axis first."""
        if not 0 <= start_channel < stop_channel <= self.num_channels():
            raise ValueError('bad channel range')
        rows = len(self.uvw)
        num_channels = stop_channel - start_channel
        if max_chunk_vis is None:
            max_chunk_vis = max(rows * num_channels, 1)
        max_chunk_rows = max(1, max_chunk_vis // num_channels)
        for start in range(0, rows, max_chunk_rows):
            end = min(rows, start + max_chunk_rows)
            order = np.argsort(self.baseline[start:end], kind='stable') + start
            ret = dict(
                uvw=self.uvw[order],
                weights=np.ascontiguousarray(
                    np.swapaxes(self.weights[order][:, start_channel:stop_channel], 0, 1)),
                vis=np.ascontiguousarray(
                    np.swapaxes(self.vis[order][:, start_channel:stop_channel], 0, 1)),
                baselines=self.baseline[order], progress=end, total=rows)
            if self.feed_angle1 is not None:
                ret['feed_angle1'] = self.feed_angle1[order]
                ret['feed_angle2'] = self.feed_angle2[order]
            yield ret

    def close(self):
        pass


def load(filename):
    """loader.load (loader.py:13-33) for the one format this package reads."""
    if not LoaderArrays.match(filename):
        raise ValueError('{}: only .npz visibility files are supported'.format(filename))
    return LoaderArrays.load(filename)


def data_iter(dataset, vis_limit, vis_load, start_channel, stop_channel):
    """loader.data_iter (loader.py:36-59): stop after ``vis_limit`` rows."""
    remaining = vis_limit
    for chunk in dataset.data_iter(start_channel, stop_channel, vis_load):
        if remaining is not None and remaining < len(chunk['uvw']):
            for key in ('uvw', 'baselines', 'feed_angle1', 'feed_angle2'):
                if key in chunk:
                    chunk[key] = chunk[key][:remaining]
            for key in ('weights', 'vis'):
                chunk[key] = chunk[key][:, :remaining]
            chunk['progress'] = chunk['total']
        yield chunk
        if remaining is not None:
            remaining -= len(chunk['uvw'])
            if remaining <= 0:
                return


def preprocess_visibilities(dataset, collector, start_channel, stop_channel, polarization_matrices,
                            vis_load=32 * 1048576, vis_limit=None):
    """frontend.preprocess_visibilities (frontend.py:40-84): feed every block of the loader, in
    loader order, to ``collector.add`` (a :class:`~.preprocess.VisibilityCollectorDevice`; its
    conversion and compression kernels run asynchronously on its queue, which plays the role of
    the reference's preprocessing thread).  ``polarization_matrices`` = (mueller_stokes,
    mueller_circular) as ``polarization.polarization_matrices`` returns them.  Closes the
    collector and returns it."""
    try:
        for chunk in data_iter(dataset, vis_limit, vis_load, start_channel, stop_channel):
            collector.add(chunk['uvw'], chunk['weights'], chunk['vis'],
                          chunk.get('feed_angle1'), chunk.get('feed_angle2'), *polarization_matrices)
    finally:
        collector.close()
    return collector
