"""Minimal device runtime behind the operator classes.

Plays the role that the third-party ``katsdpsigproc.accel`` package plays for
the reference (context / command queue / DeviceArray / IOSlot / Operation /
OperationSequence), with the subset of that interface the imaging operators
use.  Device memory and streams come from PyTorch-ROCm (plumbing only); all
compute goes through libkimg.so.
"""
import numpy as np


def divup(x, y):
    return (x + y - 1) // y


def roundup(x, y):
    return divup(x, y) * y


def _torch():
    import torch
    return torch


_TORCH_DTYPES = None


def torch_dtype(dtype):
    global _TORCH_DTYPES
    torch = _torch()
    if _TORCH_DTYPES is None:
        _TORCH_DTYPES = {
            np.dtype(np.int16): torch.int16, np.dtype(np.int32): torch.int32,
            np.dtype(np.uint32): torch.int32, np.dtype(np.float32): torch.float32,
            np.dtype(np.float64): torch.float64, np.dtype(np.complex64): torch.complex64,
            np.dtype(np.uint8): torch.uint8, np.dtype(np.int64): torch.int64,
        }
    return _TORCH_DTYPES[np.dtype(dtype)]


class Context:
    """One GPU.  ``create_some_context()`` picks ``cuda:LOCAL_RANK``."""

    def __init__(self, device_index=0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError('katsdpimager_amd needs a HIP device; there is no CPU fallback')
        self.device = torch.device('cuda', device_index)
        torch.cuda.set_device(self.device)
        # every code object of libkimg loaded on this device now, by this thread: first launches
        # that several threads make at the same moment are not safe against the runtime's lazy
        # loading (include/kimg.h: kimg_preload)
        from . import _lib
        _lib.preload()

    def create_command_queue(self, stream=None):
        return CommandQueue(self, stream)


def create_some_context(device_index=None):
    import os
    if device_index is None:
        device_index = int(os.environ.get('LOCAL_RANK', '0'))
    return Context(device_index)


class Event:
    def __init__(self, event):
        self._event = event

    def wait(self):
        self._event.synchronize()


class CommandQueue:
    """In-order queue = one HIP stream."""

    def __init__(self, context, stream=None):
        torch = _torch()
        self.context = context
        self.stream = stream if stream is not None else torch.cuda.Stream(device=context.device)

    @property
    def handle(self):
        return self.stream.cuda_stream

    def finish(self):
        self.stream.synchronize()

    def enqueue_marker(self):
        ev = _torch().cuda.Event()
        ev.record(self.stream)
        return Event(ev)

    def enqueue_wait_for_events(self, events):
        """Later work on this queue starts only after `events` (of other queues) have fired;
        the host does not wait (katsdpsigproc's AbstractCommandQueue.enqueue_wait_for_events)."""
        for ev in events:
            self.stream.wait_event(ev._event)


class HostArray(np.ndarray):
    """Page-locked numpy array (falls back to pageable memory without a device)."""

    def __new__(cls, shape, dtype, padded_shape=None, context=None):
        torch = _torch()
        shape = tuple(int(s) for s in shape)
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if torch.cuda.is_available():
            storage = torch.empty(max(n, 1), dtype=torch.uint8, pin_memory=True)
            base = storage.numpy()[:n].view(dtype).reshape(shape)
            obj = base.view(cls)
            obj._storage = storage
        else:
            obj = np.empty(shape, dtype).view(cls)
            obj._storage = None
        return obj

    def __array_finalize__(self, obj):
        self._storage = getattr(obj, '_storage', None)

    @classmethod
    def padded_view(cls, array):
        return array


class DeviceArray:
    """Dense device array.  ``padded_shape`` always equals ``shape`` here (the HIP
    kernels take explicit strides, so no padding is ever required)."""

    def __init__(self, context, shape, dtype, padded_shape=None, tensor=None, queue=None):
        torch = _torch()
        self.context = context
        self.shape = tuple(int(s) for s in shape)
        self.padded_shape = self.shape
        self.dtype = np.dtype(dtype)
        if tensor is None:
            tensor = torch.empty(self.shape, dtype=torch_dtype(dtype), device=context.device)
        self.tensor = tensor
        self._streams = set()
        if queue is not None:
            self.used_on(queue)

    @property
    def ptr(self):
        return self.tensor.data_ptr()

    def used_on(self, command_queue):
        """Tell torch's caching allocator that `command_queue`'s stream uses this memory (the
        kernels of libkimg see raw pointers, so torch cannot know): when the array is freed, its
        block is then not handed out again before the work enqueued on that stream so far has
        finished.  Cheap after the first call per stream."""
        key = command_queue.handle
        if key not in self._streams:
            self._streams.add(key)
            if self.tensor.is_cuda:
                self.tensor.record_stream(command_queue.stream)

    @property
    def buffer(self):
        return self.tensor

    def empty_like(self):
        return HostArray(self.shape, self.dtype, context=self.context)

    def _as_host_tensor(self, ary):
        torch = _torch()
        ary = np.ascontiguousarray(ary, self.dtype)
        if self.dtype == np.uint32:
            ary = ary.view(np.int32)
        return torch.from_numpy(ary)

    def set(self, command_queue, ary):
        ary = np.asarray(ary)
        if ary.shape != self.shape:
            raise ValueError('shape mismatch: {} vs {}'.format(ary.shape, self.shape))
        self.used_on(command_queue)
        with _torch().cuda.stream(command_queue.stream):
            self.tensor.copy_(self._as_host_tensor(ary), non_blocking=False)

    def set_async(self, command_queue, ary):
        self.set(command_queue, ary)

    def get(self, command_queue, ary=None):
        self.used_on(command_queue)
        with _torch().cuda.stream(command_queue.stream):
            host = self.tensor.cpu().numpy()
        if self.dtype == np.uint32:
            host = host.view(np.uint32)
        if ary is None:
            return host
        ary[...] = host
        return ary

    def get_async(self, command_queue, ary=None):
        return self.get(command_queue, ary)

    def zero(self, command_queue):
        self.used_on(command_queue)
        with _torch().cuda.stream(command_queue.stream):
            self.tensor.zero_()

    def set_region(self, command_queue, ary, device_region, ary_region, blocking=True):
        torch = _torch()
        src = np.asarray(ary)[ary_region]
        if self.dtype == np.uint32:
            src = src.view(np.int32)
        self.used_on(command_queue)
        with torch.cuda.stream(command_queue.stream):
            dst = self.tensor[device_region]
            src_t = torch.from_numpy(np.ascontiguousarray(src)) if not src.flags.c_contiguous \
                else torch.from_numpy(src)
            dst.copy_(src_t.reshape(dst.shape), non_blocking=not blocking)

    def get_region(self, command_queue, ary, device_region, ary_region, blocking=True):
        self.used_on(command_queue)
        with _torch().cuda.stream(command_queue.stream):
            host = self.tensor[device_region].cpu().numpy()
        np.asarray(ary)[ary_region] = host.reshape(np.asarray(ary)[ary_region].shape)

    def copy_region(self, command_queue, dest, src_region, dest_region):
        self.used_on(command_queue)
        dest.used_on(command_queue)
        with _torch().cuda.stream(command_queue.stream):
            d = dest.tensor[dest_region]
            d.copy_(self.tensor[src_region].reshape(d.shape))


class DeviceAllocator:
    def __init__(self, context):
        self.context = context

    def allocate(self, shape, dtype, padded_shape=None):
        return DeviceArray(self.context, shape, dtype)


class Dimension:
    """Size of one axis of a slot.  Alignment/padding requests of the reference API are
    accepted and ignored (our kernels need none)."""

    def __init__(self, size, alignment=1, min_padded_size=None, exact=False):
        self.size = int(size)

    def link(self, other):
        if self.size != other.size:
            raise ValueError('linked dimensions have different sizes')

    def __int__(self):
        return self.size

    def __index__(self):
        return self.size


def _dims(shape):
    return tuple(int(s) for s in shape)


class IOSlotBase:
    pass


class IOSlot(IOSlotBase):
    def __init__(self, dimensions, dtype):
        self.dimensions = [d if isinstance(d, Dimension) else Dimension(d) for d in dimensions]
        self.shape = _dims(dimensions)
        self.dtype = np.dtype(dtype)
        self.buffer = None

    def required_padded_shape(self):
        return self.shape

    def is_bound(self):
        return self.buffer is not None

    def bind(self, buffer):
        if buffer is not None:
            if buffer.shape != self.shape or buffer.dtype != self.dtype:
                raise ValueError('buffer of shape {} / {} cannot bind to slot {} / {}'.format(
                    buffer.shape, buffer.dtype, self.shape, self.dtype))
        self.buffer = buffer

    def allocate(self, allocator, bind=True):
        buf = allocator.allocate(self.shape, self.dtype)
        if bind:
            self.bind(buf)
        return buf


class AliasIOSlot(IOSlotBase):
    """Several slots of child operations that share one buffer (a "compound")."""

    def __init__(self, children):
        self.children = list(children)
        first = self.children[0]
        for c in self.children[1:]:
            if c.shape != first.shape or c.dtype != first.dtype:
                raise ValueError('aliased slots differ: {} {} vs {} {}'.format(
                    c.shape, c.dtype, first.shape, first.dtype))
        self.shape = first.shape
        self.dtype = first.dtype
        self.dimensions = first.dimensions

    @property
    def buffer(self):
        return self.children[0].buffer

    def required_padded_shape(self):
        return self.shape

    def is_bound(self):
        return self.buffer is not None

    def bind(self, buffer):
        for c in self.children:
            c.bind(buffer)

    def allocate(self, allocator, bind=True):
        buf = allocator.allocate(self.shape, self.dtype)
        if bind:
            self.bind(buf)
        return buf


class Operation:
    def __init__(self, command_queue, allocator=None):
        self.command_queue = command_queue
        self.allocator = allocator if allocator is not None \
            else DeviceAllocator(command_queue.context)
        self.slots = {}

    def bind(self, **kwargs):
        for name, buffer in kwargs.items():
            self.slots[name].bind(buffer)
            if buffer is not None:
                buffer.used_on(self.command_queue)

    def buffer(self, name):
        return self.slots[name].buffer

    def ensure_bound(self, name):
        slot = self.slots[name]
        if not slot.is_bound():
            slot.allocate(self.allocator).used_on(self.command_queue)

    def ensure_all_bound(self):
        for name in self.slots:
            self.ensure_bound(name)

    def required_bytes(self):
        return sum(int(np.prod(s.shape)) * s.dtype.itemsize for s in self.slots.values())

    def parameters(self):
        return {}

    def _run(self):
        raise NotImplementedError

    def __call__(self, **kwargs):
        self.bind(**kwargs)
        self.ensure_all_bound()
        return self._run()


class OperationSequence(Operation):
    """Operations run in order, with named groups of their slots aliased together."""

    def __init__(self, command_queue, operations, compounds=None, allocator=None):
        super().__init__(command_queue, allocator)
        self.operations = dict(operations)
        self._order = [name for name, _ in operations]
        used = set()
        for name, members in (compounds or {}).items():
            children = []
            for m in members:
                op_name, slot_name = m.split(':')
                children.append(self.operations[op_name].slots[slot_name])
                used.add(m)
            if children:
                self.slots[name] = AliasIOSlot(children)
        for op_name, op in operations:
            for slot_name, slot in op.slots.items():
                key = op_name + ':' + slot_name
                if key not in used:
                    self.slots[key] = slot

    def _run(self):
        for name in self._order:
            self.operations[name]()
