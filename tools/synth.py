"""Synthetic observation for benchmarks and full-size tests (no katpoint / casacore).

A MeerKAT-like 64-antenna array (48 antennas in a Gaussian core of sigma 300 m, 16 on three
spiral arms out to 4 km), earth-rotation synthesis at declination -45 deg over hour angle
+-4 h, all 2016 baselines, baseline-major order as the reference's loaders deliver it
(loader_ms.py:465-467).  UVW is quantised with the reference preprocessor's rules
(preprocess.cpp:313-323, 435-507): w<0 flip, u/v scaled to cells and split into cell /
sub-cell, w_plane = min(trunc(w*w_scale + W/2), slices*W-1).  No compression (worst case).

Everything is generated with torch on the requested device so that a 50 M visibility
channel takes a second or two on the GPU; the same code runs on the CPU for tests.
"""
import math

import numpy as np
import torch

LATITUDE = math.radians(-30.71)
DECLINATION = math.radians(-45.0)
N_ANTENNAS = 64
LAYOUT_SEED = 20241008


def antenna_positions():
    """ENU positions in metres, float64 [64][3]."""
    rng = np.random.default_rng(LAYOUT_SEED)
    core = rng.normal(0.0, 300.0, size=(48, 2))
    arms = []
    for k in range(16):
        arm = k % 3
        r = 800.0 + (4000.0 - 800.0) * (k // 3 + rng.uniform(0.2, 0.9)) / 6.0
        theta = 2 * math.pi * arm / 3 + 0.9 * r / 4000.0 + rng.normal(0, 0.05)
        arms.append([r * math.cos(theta), r * math.sin(theta)])
    en = np.concatenate([core, np.array(arms)])
    up = rng.normal(0.0, 3.0, size=(N_ANTENNAS, 1))
    return np.concatenate([en, up], axis=1)


def baselines_equatorial():
    """Baseline vectors (Lx, Ly, Lz) in the equatorial frame, float64 [2016][3]."""
    enu = antenna_positions()
    i, j = np.triu_indices(N_ANTENNAS, 1)
    d = enu[j] - enu[i]
    e, n, u = d[:, 0], d[:, 1], d[:, 2]
    sl, cl = math.sin(LATITUDE), math.cos(LATITUDE)
    return np.stack([-sl * n + cl * u, e, cl * n + sl * u], axis=1)


class Observation:
    """Quantised visibilities of one channel, resident on `device`.

    Attributes: uv int16 [N][4], w_plane int16 [N], vis complex64 [N][P],
    weights float32 [N][P]; plus the parameters needed to build matching operators:
    cell_size, longest_baseline, max_w (metres), wavelength, pixel_size.
    """


def track_uvw(n_vis, device='cpu'):
    """The channel-independent part of an observation: UVW in metres, float32 [n_vis][3], in loader
    order (baseline-major earth-rotation tracks), what SURVEY 8e has the loading rank broadcast."""
    dev = torch.device(device)
    bl = torch.from_numpy(baselines_equatorial()).to(dev)                 # [B][3] f64
    nb = bl.shape[0]
    T = -(-n_vis // nb)
    ha = torch.linspace(-math.pi / 3, math.pi / 3, T, dtype=torch.float64, device=dev)
    sh, ch = torch.sin(ha)[None, :], torch.cos(ha)[None, :]
    sd, cd = math.sin(DECLINATION), math.cos(DECLINATION)
    lx, ly, lz = bl[:, 0:1], bl[:, 1:2], bl[:, 2:3]
    u = (sh * lx + ch * ly).reshape(-1)[:n_vis]
    v = (-sd * ch * lx + sd * sh * ly + cd * lz).reshape(-1)[:n_vis]
    w = (cd * ch * lx - cd * sh * ly + sd * lz).reshape(-1)[:n_vis]
    return torch.stack([u.to(torch.float32), v.to(torch.float32), w.to(torch.float32)], dim=1).contiguous()


def make_observation(pixels, n_vis, w_planes, num_polarizations=1, oversample=8,
                     wavelength=0.21, cover=0.30, seed=2, device='cpu', w_slices=1,
                     channel_scale=1.0, uvw=None):
    """`cover`: longest baseline as a fraction of the grid size in cells (0.30).
    `channel_scale`: relative frequency of this channel (scales uvw in wavelengths).
    `uvw`: the tracks in metres (:func:`track_uvw`) if they were made elsewhere (broadcast by the
    loading rank); made here otherwise."""
    dev = torch.device(device)
    bl = torch.from_numpy(baselines_equatorial()).to(dev)                 # [B][3] f64
    if uvw is None:
        uvw = track_uvw(n_vis, dev)
    assert tuple(uvw.shape) == (n_vis, 3) and uvw.dtype == torch.float32
    u, v, w = uvw[:, 0], uvw[:, 1], uvw[:, 2]
    # array-wide scales are taken over the full tracks so that they do not depend on n_vis
    longest = float(torch.sqrt((bl ** 2).sum(dim=1)).max())
    max_w = float(torch.sqrt((bl ** 2).sum(dim=1)).max())
    cell_size = longest / (cover * pixels)            # metres per uv cell at channel_scale 1
    obs = Observation()
    obs.wavelength = wavelength / channel_scale
    obs.cell_size = cell_size / channel_scale
    obs.longest_baseline = longest
    obs.max_w = max_w
    obs.pixels = pixels
    obs.pixel_size = obs.wavelength / (obs.cell_size * pixels)
    obs.w_planes = w_planes
    obs.w_slices = w_slices
    obs.oversample = oversample

    # preprocess.cpp:435-507 in float32
    obs.uvw = uvw                                             # raw metres, loader order
    flip = w < 0
    u = torch.where(flip, -u, u)
    v = torch.where(flip, -v, v)
    w = torch.where(flip, -w, w)
    uv_scale = np.float32(1.0) / np.float32(obs.cell_size)
    w_scale = np.float32((np.float32(w_slices) - np.float32(0.5)) * np.float32(w_planes)
                         / np.float32(max_w))
    u = u * float(uv_scale)
    v = v * float(uv_scale)
    wq = torch.trunc(w * float(w_scale) + float(np.float32(w_planes) * np.float32(0.5)))
    wsp = torch.clamp(wq.to(torch.int32), max=w_slices * w_planes - 1)

    def split(x):
        xs = torch.floor(x * float(oversample)).to(torch.int32)
        pix = torch.div(xs, oversample, rounding_mode='floor')
        return pix, xs - pix * oversample

    pu, su = split(u)
    pv, sv = split(v)
    obs.uv = torch.stack([pu, pv, su, sv], dim=1).to(torch.int16).contiguous()
    obs.w_plane = (wsp % w_planes).to(torch.int16).contiguous()
    obs.w_slice = torch.div(wsp, w_planes, rounding_mode='floor').to(torch.int16).contiguous()
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    re = torch.rand((n_vis, num_polarizations), generator=gen, device=dev) * 2 - 1
    im = torch.rand((n_vis, num_polarizations), generator=gen, device=dev) * 2 - 1
    vis = torch.complex(re, im)
    obs.vis = torch.where(flip[:, None], torch.conj(vis), vis).contiguous()
    obs.weights = torch.rand((n_vis, num_polarizations), generator=gen, device=dev) * 0.9 + 0.1
    obs.n_vis = n_vis
    return obs


def make_parameters(obs, num_polarizations=1, kernel_width=28, antialias_width=7.0,
                    image_oversample=4, degrid=False):
    """katsdpimager_amd parameter objects matching an :class:`Observation`."""
    from katsdpimager_amd import parameters
    fixed_i = parameters.FixedImageParameters(list(range(num_polarizations)), np.float32)
    ip = parameters.ImageParameters(fixed_i, 1.0, None, obs.wavelength, None,
                                    pixel_size=obs.pixel_size, pixels=obs.pixels)
    fixed_g = parameters.FixedGridParameters(antialias_width, obs.oversample, image_oversample,
                                             obs.max_w, kernel_width, degrid=degrid)
    gp = parameters.GridParameters(fixed_g, obs.w_slices, obs.w_planes)
    ap = parameters.ArrayParameters(13.5, obs.longest_baseline)
    return ip, gp, ap


# ---- input orders --------------------------------------------------
def _track_indices(obs):
    """(baseline, time sample) of every visibility of a baseline-major observation."""
    nb = baselines_equatorial().shape[0]
    T = -(-obs.n_vis // nb)
    idx = torch.arange(obs.n_vis, device=obs.uv.device)
    return idx // T, idx % T, nb, T


def _copy_with(obs, uv, w_plane, vis, weights):
    out = Observation()
    out.__dict__.update(obs.__dict__)
    out.uv, out.w_plane, out.vis, out.weights = uv.contiguous(), w_plane.contiguous(), \
        vis.contiguous(), weights.contiguous()
    out.n_vis = int(uv.shape[0])
    for name in ('uvw', 'w_slice'):
        out.__dict__.pop(name, None)        # (not reordered: unused by the consumers of these)
    return out


def _reordered(obs, order):
    return _copy_with(obs, obs.uv[order], obs.w_plane[order], obs.vis[order], obs.weights[order])


def order_time_major(obs):
    """All baselines of dump 0, then of dump 1, ...: consecutive visibilities never share a
    baseline (the order of a time-sorted measurement set read without the loaders' sort)."""
    b, t, nb, T = _track_indices(obs)
    order = torch.argsort(t * nb + b)
    return {'obs': _reordered(obs, order), 'note': 'time-major: a baseline jump at every record'}


def order_shuffled(obs, seed=3):
    """No locality at all (SURVEY 8d "adversarial")."""
    gen = torch.Generator(device=obs.uv.device)
    gen.manual_seed(seed)
    order = torch.randperm(obs.n_vis, generator=gen, device=obs.uv.device)
    return {'obs': _reordered(obs, order), 'note': 'uniformly shuffled'}


def compress_adjacent(obs):
    """Merge runs of adjacent records with equal (u, v, sub_u, sub_v, w_plane), summing their
    visibilities and weights: what the reference's preprocessor hands to the gridder
    (preprocess.cpp:334-397; sums here in atomic order, which only matters to the last bit)."""
    uv = obs.uv.to(torch.int64)
    key = ((uv[:, 0] + 32768) << 48) | ((uv[:, 1] + 32768) << 32) | (uv[:, 2] << 24) \
        | (uv[:, 3] << 16) | obs.w_plane.to(torch.int64)
    head = torch.ones(obs.n_vis, dtype=torch.bool, device=key.device)
    head[1:] = key[1:] != key[:-1]
    seg = torch.cumsum(head.to(torch.int64), 0) - 1
    m = int(seg[-1]) + 1
    P = obs.vis.shape[1]
    vis = torch.zeros((m, P, 2), dtype=torch.float32, device=key.device)
    vis.index_add_(0, seg, torch.view_as_real(obs.vis))
    weights = torch.zeros((m, P), dtype=torch.float32, device=key.device)
    weights.index_add_(0, seg, obs.weights)
    heads = torch.nonzero(head)[:, 0]
    return _copy_with(obs, obs.uv[heads], obs.w_plane[heads], torch.view_as_complex(vis), weights)


def order_loader_blocks(obs, dumps=256):
    """The stream the reference's gridder sees: the loaders deliver blocks of `dumps` consecutive
    dumps (--vis-load 32 Mi over a batch of 16 channels: about 250 dumps of a 64-antenna katdal
    file, loader_katdal.py:326, about 1000 rows per baseline of a measurement set,
    loader_ms.py:383), each block sorted by baseline (loader_ms.py:465-467), and the preprocessor
    merges adjacent records that fall on the same sub-cell (preprocess.cpp:334-397)."""
    b, t, nb, T = _track_indices(obs)
    order = torch.argsort((t // dumps) * (nb * dumps) + b * dumps + (t % dumps))
    merged = compress_adjacent(_reordered(obs, order))
    return {'obs': merged,
            'note': 'baseline-sorted blocks of {} dumps, adjacent-merged: {} of {} records kept '
                    '({:.1f} %); the rate counts merged records'.format(
                        dumps, merged.n_vis, obs.n_vis, 100.0 * merged.n_vis / obs.n_vis)}


def grid_truth_fp64(kernel, uv, w_plane, vis, weights_grid, kernel_width, batch=16384):
    """float64 evaluation of the gridding sum (grid.py:1032-1052) with torch on the tensors' device:
    grid[p][v0+j][u0+k] += vis * wgrid[p][v+G/2][u+G/2] * conj(kern[w][sv][j] kern[w][su][k]).
    `kernel`: numpy complex64 [W][OV][K]; returns complex128 [P][G][G].  Test / bench
    infrastructure (an independent check of both gridder forms), not part of the product."""
    dev = uv.device
    kern = torch.from_numpy(np.ascontiguousarray(kernel)).to(dev).to(torch.complex128)
    P, G = weights_grid.shape[0], weights_grid.shape[-1]
    K = kernel_width
    half = G // 2
    bias = (K - 1) // 2 - half
    out = torch.zeros((P, G * G, 2), dtype=torch.float64, device=dev)
    taps = torch.arange(K, device=dev)
    n = uv.shape[0]
    for s in range(0, n, batch):
        e = min(s + batch, n)
        u = uv[s:e, 0].to(torch.int64)
        v = uv[s:e, 1].to(torch.int64)
        su = uv[s:e, 2].to(torch.int64)
        sv = uv[s:e, 3].to(torch.int64)
        wp = w_plane[s:e].to(torch.int64)
        kv = torch.conj(kern[wp, sv])                       # [B][K]
        ku = torch.conj(kern[wp, su])
        idx = ((v - bias)[:, None, None] + taps[None, :, None]) * G \
            + ((u - bias)[:, None, None] + taps[None, None, :])
        for p in range(P):
            wgt = weights_grid[p][v + half, u + half].to(torch.float64)
            smp = vis[s:e, p].to(torch.complex128) * wgt
            vals = smp[:, None, None] * kv[:, :, None] * ku[:, None, :]
            out[p].index_add_(0, idx.reshape(-1), torch.view_as_real(vals).reshape(-1, 2))
    return torch.view_as_complex(out).reshape(P, G, G)


def add_point_sources(obs, n_sources=200, seed=4, flux=(0.5, 2.0), noise=0.01, batch=1 << 18):
    """Replace obs.vis by weights * (visibilities of `n_sources` point sources + noise): a sky that
    gives CLEAN something to do (BASELINE config 5).  Sources sit on pixel centres of the inner 60 %
    of the image; V = sum_s flux_s exp(-2 pi i (l u + m v + (n - 1) w)) with (u, v, w) the exact
    (unquantised) coordinates in wavelengths after the w >= 0 flip, i.e. matching obs.uv.
    Returns (pixel positions int [S][2] as (y, x), flux [S])."""
    dev = obs.uv.device
    rs = np.random.RandomState(seed)
    G = obs.pixels
    pos = rs.randint(int(0.2 * G), int(0.8 * G), (n_sources, 2))
    fl = rs.uniform(flux[0], flux[1], n_sources)
    l = (pos[:, 1] - 0.5 * G) * obs.pixel_size
    m = (pos[:, 0] - 0.5 * G) * obs.pixel_size
    n1 = np.sqrt(1.0 - l * l - m * m) - 1.0
    lt = torch.from_numpy(l.astype(np.float32)).to(dev)
    mt = torch.from_numpy(m.astype(np.float32)).to(dev)
    nt = torch.from_numpy(n1.astype(np.float32)).to(dev)
    ft = torch.from_numpy(fl.astype(np.float32)).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    P = obs.vis.shape[1]
    inv_wl = 1.0 / obs.wavelength
    # what a telescope delivers for obs.uvw: unweighted, before the w >= 0 flip (the
    # preprocessor's input; obs.vis is its output: flipped / conjugated and times the weight)
    raw = torch.empty_like(obs.vis)
    for s in range(0, obs.n_vis, batch):
        e = min(s + batch, obs.n_vis)
        uvw = obs.uvw[s:e]
        sign = torch.where(uvw[:, 2] < 0, -inv_wl, inv_wl).to(torch.float32)
        u, v, w = uvw[:, 0] * sign, uvw[:, 1] * sign, uvw[:, 2] * sign
        turns = u[:, None] * lt[None, :] + v[:, None] * mt[None, :] + w[:, None] * nt[None, :]
        turns = turns - torch.floor(turns)
        ang = turns * (-2.0 * math.pi)
        re = (torch.cos(ang) * ft[None, :]).sum(dim=1)
        im = (torch.sin(ang) * ft[None, :]).sum(dim=1)
        nre = torch.randn((e - s, P), generator=gen, device=dev) * noise
        nim = torch.randn((e - s, P), generator=gen, device=dev) * noise
        model = torch.complex(re[:, None] + nre, im[:, None] + nim)
        obs.vis[s:e] = model * obs.weights[s:e]
        raw[s:e] = torch.where((uvw[:, 2] < 0)[:, None], torch.conj(model), model)
    obs.raw_vis = raw
    return pos, fl
