"""Channel-sharded multi-GPU execution (new relative to the reference, which is single
device and loops over channels serially, frontend.py:749-767).

Spectral channels are independent (own kernel table, visibilities, weights, PSF, images,
CLEAN loop), so the path shards with no data-path collective: channel c runs on rank
c mod world_size, one process per GPU.  The only collectives are a start-up broadcast of
channel-independent tables from the loading rank (RCCL over xGMI on GPUs; gloo in CPU
tests) and small reductions/gathers of per-channel statistics and timings.
"""
import torch
import torch.distributed as dist


def world():
    """(rank, world_size), (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _collective():
    """Do the helpers below go through torch.distributed?  Whenever a process group exists, also
    one of a single rank (the backend's code path then runs on the one device there is: how the
    RCCL path is exercised on a one-GPU box)."""
    return dist.is_available() and dist.is_initialized()


def free_port():
    """A TCP port nobody listens on, for a rendezvous on 127.0.0.1 started without a launcher."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


def assign_channels(num_channels, world_size, rank):
    """Channels imaged by `rank`: c with c mod world_size == rank."""
    if not 0 <= rank < world_size:
        raise ValueError('rank {} outside world of {}'.format(rank, world_size))
    return list(range(rank, num_channels, world_size))


def channel_frequency_scale(channel, num_channels, spread=0.03):
    """Relative frequency of a channel in a band of +-spread around 1 (uvw in wavelengths
    scales with it); 1.0 for a single channel."""
    if num_channels <= 1:
        return 1.0
    return 1.0 + spread * (2.0 * channel / (num_channels - 1) - 1.0)


def broadcast_shared(tensors, src=0):
    """Broadcast channel-independent tables (dict name -> tensor, allocated with the right
    shape/dtype on every rank) from `src`, in place.  No-op without a process group."""
    if not _collective():
        return tensors
    for name in sorted(tensors):
        dist.broadcast(tensors[name], src=src)
    return tensors


def max_over_ranks(value, device='cpu'):
    """Maximum of a Python float over all ranks (timings)."""
    if not _collective():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def gather_stats(values, device='cpu'):
    """All-gather a fixed-length list of floats (per-channel statistics such as noise, peak,
    minor cycles); returns a [world_size][len(values)] tensor on every rank."""
    rank, size = world()
    t = torch.tensor([values], dtype=torch.float64, device=device)
    if not _collective():
        return t
    out = [torch.empty_like(t) for _ in range(size)]
    dist.all_gather(out, t)
    return torch.cat(out, dim=0)


def image_assigned_channels(make_job, num_channels, workers=4, runner=None):
    """Image this rank's share of `num_channels` channels: channel c belongs to rank c mod world_size
    (:func:`assign_channels`); a rank with several channels keeps up to `workers` of them in flight
    on its GPU (one host thread and one HIP stream per channel in flight).

    ``make_job(channel)`` -- or ``make_job(channel, worker)`` with ``worker`` in ``range(workers)``,
    so that a callback can keep ONE imager and command queue per worker and re-use it for every
    channel that worker images, where the channels' image and grid parameters are the same
    (``frontend.process_channel`` refuses an imager made for others) -- returns the keyword
    arguments of ``frontend.process_channel`` for one channel.  Jobs are made lazily, when a worker is free to image them, and dropped when their
    channel is done: at most `workers` imagers (grids, images, FFT layer, CLEAN state, workspaces)
    exist at any time however many channels the rank owns (the reference images a band's channels one
    after the other with one imager, frontend.py:749-767).  Returns {channel: result} for this rank's
    channels; no collective is involved (gather statistics with :func:`gather_stats` if needed).
    ``runner(make_job, channels, workers)`` replaces ``frontend.process_channel_stream`` in tests."""
    rank, size = world()
    mine = assign_channels(num_channels, size, rank)
    if runner is None:
        from . import frontend
        runner = frontend.process_channel_stream
    results = runner(make_job, mine, workers=workers) if mine else []
    return dict(zip(mine, results))


def check_rank_devices(device_index, device='cpu'):
    """Every rank of the job must drive its own GPU: gathers ``device_index`` (the rank's
    ``torch.cuda.current_device()``, or any per-node unique id) and raises on rank collisions.
    Returns the list of indices in rank order.  (One node: the driver launches one rank per GPU.)"""
    rank, size = world()
    ids = [int(x) for x in gather_stats([float(device_index)], device)[:, 0].tolist()]
    if len(set(ids)) != len(ids):
        raise RuntimeError('ranks share a GPU: device index by rank = {}'.format(ids))
    return ids
