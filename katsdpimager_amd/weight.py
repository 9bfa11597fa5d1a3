"""Imaging (density) weights: natural, uniform and robust.

Operator surface of the reference's ``katsdpimager.weight`` (GridWeights,
DensityWeights, MeanWeight, compound Weights; weight.py:55-538) on libkimg.so.
See the reference module docstring (weight.py:3-44) for the definitions [Bri95].
"""
import enum

import numpy as np

from . import accel
from ._lib import lib, check


class WeightType(enum.Enum):
    """weight.py:55-58."""
    NATURAL = 0
    UNIFORM = 1
    ROBUST = 2


class GridWeightsTemplate:
    """weight.py:61-87."""
    def __init__(self, context, num_polarizations, tuning=None):
        lib()
        self.context = context
        self.num_polarizations = num_polarizations

    def instantiate(self, *args, **kwargs):
        return GridWeights(self, *args, **kwargs)


class GridWeights(accel.Operation):
    """Accumulate statistical weights on the grid without convolution (weight.py:90-183).
    Slots: **uv** int16 [max_vis][4], **weights** float32 [max_vis][pols],
    **grid** float32 [pols][H][W]."""

    def __init__(self, template, command_queue, grid_shape, max_vis, allocator=None):
        super().__init__(command_queue, allocator)
        self.template = template
        if grid_shape[0] != template.num_polarizations:
            raise ValueError('Mismatch in number of polarizations')
        if grid_shape[1] % 2 or grid_shape[2] % 2:
            raise ValueError('Odd-sized grid not currently supported')
        self.max_vis = max_vis
        self.slots['grid'] = accel.IOSlot(grid_shape, np.float32)
        self.slots['uv'] = accel.IOSlot((max_vis, accel.Dimension(4, exact=True)), np.int16)
        self.slots['weights'] = accel.IOSlot(
            (max_vis, accel.Dimension(template.num_polarizations, exact=True)), np.float32)
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

    def _run(self):
        grid = self.buffer('grid')
        P, H, W = grid.shape
        rc = lib().kimg_grid_weights(grid.ptr, W, H * W, W, H, P, self.buffer('uv').ptr,
                                     self.buffer('weights').ptr, self._num_vis,
                                     self.command_queue.handle)
        check(rc, 'kimg_grid_weights')


class DensityWeightsTemplate:
    """weight.py:186-214."""
    def __init__(self, context, num_polarizations, tuning=None):
        lib()
        self.context = context
        self.num_polarizations = num_polarizations

    def instantiate(self, *args, **kwargs):
        return DensityWeights(self, *args, **kwargs)


class DensityWeights(accel.Operation):
    """In place W -> 1/(a W + b) (0 where W == 0); returns (rms, normalized_rms)
    (weight.py:217-293).  Slot **sums** is float64 [3] here."""

    def __init__(self, template, command_queue, grid_shape, allocator=None):
        super().__init__(command_queue, allocator)
        self.template = template
        if grid_shape[0] != template.num_polarizations:
            raise ValueError('Mismatch in number of polarizations')
        self.a = 1.0
        self.b = 0.0
        self.slots['grid'] = accel.IOSlot(grid_shape, np.float32)
        self.slots['sums'] = accel.IOSlot((3,), np.float64)

    def _run(self, mean_sums=None, robust=None):
        """``mean_sums`` (device float64 [2] as :class:`MeanWeight` leaves it) and ``robust``: a =
        robust / (mean weight) worked out on the device instead of ``self.a``."""
        grid = self.buffer('grid')
        sums = self.buffer('sums')
        P, H, W = grid.shape
        if mean_sums is not None:
            rc = lib().kimg_density_weights_robust(sums.ptr, grid.ptr, W, H * W, W, H, P, mean_sums.ptr,
                                                   float(robust), self.b, self.command_queue.handle)
            check(rc, 'kimg_density_weights_robust')
        else:
            rc = lib().kimg_density_weights(sums.ptr, grid.ptr, W, H * W, W, H, P, self.a, self.b,
                                            self.command_queue.handle)
            check(rc, 'kimg_density_weights')
        s = sums.get(self.command_queue)
        rms = np.sqrt(s[2]) / s[1]
        return rms, rms * np.sqrt(s[0])


class MeanWeightTemplate:
    """weight.py:296-324."""
    def __init__(self, context, tuning=None):
        lib()
        self.context = context

    def instantiate(self, *args, **kwargs):
        return MeanWeight(self, *args, **kwargs)


class MeanWeight(accel.Operation):
    """sum W^2 / sum W over the first polarization (weight.py:326-376)."""

    def __init__(self, template, command_queue, grid_shape, allocator=None):
        super().__init__(command_queue, allocator)
        self.template = template
        self.slots['grid'] = accel.IOSlot(grid_shape, np.float32)
        self.slots['sums'] = accel.IOSlot((2,), np.float64)

    def _run(self):
        s = self.enqueue().get(self.command_queue)
        return s[1] / s[0]

    def enqueue(self):
        """The two sums, left on the device (returns the buffer)."""
        grid = self.buffer('grid')
        sums = self.buffer('sums')
        P, H, W = grid.shape
        rc = lib().kimg_mean_weight(sums.ptr, grid.ptr, W, W, H, self.command_queue.handle)
        check(rc, 'kimg_mean_weight')
        return sums


class WeightsTemplate:
    """weight.py:379-416."""
    def __init__(self, context, weight_type, num_polarizations,
                 grid_weights_tuning=None, mean_weight_tuning=None, density_weights_tuning=None):
        lib()
        self.context = context
        self.weight_type = weight_type
        self.num_polarizations = num_polarizations
        natural = weight_type == WeightType.NATURAL
        self.grid_weights = None if natural else GridWeightsTemplate(context, num_polarizations)
        self.density_weights = None if natural else DensityWeightsTemplate(context,
                                                                           num_polarizations)
        self.mean_weight = MeanWeightTemplate(context) if weight_type == WeightType.ROBUST \
            else None

    def instantiate(self, *args, **kwargs):
        return Weights(self, *args, **kwargs)


class Weights(accel.OperationSequence):
    """Compound imaging-weights operation (weight.py:419-538): ``clear()``, ``grid(N)``
    per batch, then ``finalize()`` -> (rms, normalized_rms).  Slots **grid** (and **uv**,
    **weights** unless natural weighting)."""

    def __init__(self, template, command_queue, grid_shape, max_vis, allocator=None):
        self.template = template
        operations = []
        compounds = {'grid': []}
        self._grid_weights = self._mean_weight = self._density_weights = None
        self.robustness = None
        if template.grid_weights is not None:
            self._grid_weights = template.grid_weights.instantiate(
                command_queue, grid_shape, max_vis, allocator)
            operations.append(('grid_weights', self._grid_weights))
            compounds['grid'].append('grid_weights:grid')
            compounds['uv'] = ['grid_weights:uv']
            compounds['weights'] = ['grid_weights:weights']
        if template.mean_weight is not None:
            self._mean_weight = template.mean_weight.instantiate(
                command_queue, grid_shape, allocator)
            operations.append(('mean_weight', self._mean_weight))
            compounds['grid'].append('mean_weight:grid')
            self.robustness = 0.0
        if template.density_weights is not None:
            self._density_weights = template.density_weights.instantiate(
                command_queue, grid_shape, allocator)
            operations.append(('density_weights', self._density_weights))
            compounds['grid'].append('density_weights:grid')
        super().__init__(command_queue, operations, compounds, allocator=allocator)
        if not compounds['grid']:
            # natural weighting: just a grid that finalize() fills with ones
            self.slots['grid'] = accel.IOSlot(grid_shape, np.float32)

    def _run(self):
        raise NotImplementedError('Weights should not be used as a callable')

    def clear(self):
        self.ensure_all_bound()
        if self._grid_weights is not None:
            self.buffer('grid').zero(self.command_queue)

    def grid(self, N):
        self.ensure_all_bound()
        if self._grid_weights is not None:
            self._grid_weights.num_vis = N
            return self._grid_weights()

    def finalize(self):
        self.ensure_all_bound()
        if self._mean_weight is not None:
            # weight.py:525-531 without the host between the two kernels: S2 = robust / mean weight
            # is worked out where the sums are (kimg_density_weights_robust)
            self._density_weights.b = 1.0
            return self._density_weights._run(self._mean_weight.enqueue(),
                                              (5 * 10**(-self.robustness))**2)
        if self._density_weights is not None:
            return self._density_weights()
        grid = self.buffer('grid')
        check(lib().kimg_fill(grid.ptr, int(np.prod(grid.shape)), 1.0,
                              self.command_queue.handle), 'kimg_fill')
        return None, 1.0
