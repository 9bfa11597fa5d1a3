"""GPU parity of the device preprocessor / visibility store against the
restated collector (oracle.VisibilityCollector, pinned by test_preprocess.py's known answers)."""
import types

import numpy as np
import pytest

import golden_inputs as gi
from helpers import context_queue
from oracle import kimg_oracle as orc

pytestmark = pytest.mark.gpu


def _params(configs, P):
    ips, gps = [], []
    for c in configs:
        fixed_i = types.SimpleNamespace(polarizations=list(range(P)))
        ips.append(types.SimpleNamespace(fixed=fixed_i, cell_size=c['cell_size']))
        fixed_g = types.SimpleNamespace(max_w=c['max_w'], oversample=c['oversample'])
        gps.append(types.SimpleNamespace(fixed=fixed_g, w_slices=c['w_slices'],
                                         w_planes=c['w_planes']))
    return ips, gps


def _collect_device(configs, P, buffer_size, batches):
    from katsdpimager_amd import preprocess
    ctx, q = context_queue()
    ips, gps = _params(configs, P)
    coll = preprocess.VisibilityCollectorDevice(q, ips, gps, buffer_size)
    for b in batches:
        coll.add(*b)
    coll.close()
    return coll


def _read(reader, channel, w_slice, block_size, dtype):
    pieces = [p.copy() for p in reader.iter_slice(channel, w_slice, block_size)]
    if pieces:
        return np.rec.array(np.hstack(pieces))
    return np.rec.recarray(0, dtype)


def _known():
    import test_oracle_known_answers as tk
    return tk._known_preprocess_inputs()


def test_store_merge_of_a_million_equal_coordinates():
    """A slice that is ONE run of equal coordinates (constant uv: a snapshot's PSF, autocorrelations)
    through kimg_store_reorder with merge: runs are cut every STORE_MAX_RUN sorted positions, so no
    thread sums more than that many records, and the records that come out are the restatement's,
    bit for bit (each the left-to-right float32 sum of its part, in arrival order)."""
    from katsdpimager_amd import accel
    from katsdpimager_amd._lib import lib, check
    ctx, q = context_queue()
    rs = np.random.RandomState(77)
    n, P = 1_200_000, 2
    uv4 = np.tile(np.array([[3, -2, 5, 1]], np.int16), (n, 1))
    uv4[:1000, 0] = 4                       # (and a second, short run)
    w_plane = np.full(n, 7, np.int16)
    weights = rs.uniform(0.5, 1.5, (n, P)).astype(np.float32)
    vis = (rs.standard_normal((n, P)) + 1j * rs.standard_normal((n, P))).astype(np.complex64)
    d = [accel.DeviceArray(ctx, a.shape, a.dtype) for a in (uv4, w_plane, weights, vis)]
    for dev, a in zip(d, (uv4, w_plane, weights, vis)):
        dev.set(q, a)
    o = [accel.DeviceArray(ctx, a.shape, a.dtype) for a in (uv4, w_plane, weights, vis)]
    count = accel.DeviceArray(ctx, (1,), np.int64)
    L = lib()
    ws_bytes = int(L.kimg_store_reorder_workspace_bytes(n))
    ws = accel.DeviceArray(ctx, (ws_bytes,), np.uint8)
    check(L.kimg_store_reorder(P, n, 28, 8, 32, 1, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, o[0].ptr,
                               o[1].ptr, o[2].ptr, o[3].ptr, count.ptr, ws.ptr, ws_bytes, q.handle),
          'kimg_store_reorder')
    m = int(count.get(q)[0])
    want = orc.store_reorder(uv4, w_plane, weights, vis, 28, 8, 32, True)
    assert m == len(want[0]) and n // orc.STORE_MAX_RUN <= m <= n // orc.STORE_MAX_RUN + 3
    for dev, w_ in zip(o, want):
        np.testing.assert_array_equal(dev.get(q)[:m].view(np.uint8), np.ascontiguousarray(w_).view(np.uint8))


@pytest.mark.parametrize('use_feed_angles', [False, True])
def test_known_answers(use_feed_angles):
    """test_preprocess.py:76-136 (`_test_impl` + `check`) on the device collector."""
    uvw, weights, vis, configs, expected = _known()
    ident = np.identity(4, np.complex64)
    fa = np.zeros(4, np.float32) if use_feed_angles else None
    coll = _collect_device(configs, 4, 64,
                           [(uvw, weights, vis, fa, fa, ident, ident if use_feed_angles else None)])
    assert coll.num_input == 8 and coll.num_output == 5
    reader = coll.reader()
    assert reader.num_channels == 2
    for ch, e in enumerate(expected):
        assert reader.len(ch, 0) == len(e['uv'])
        for block_size in [None, 1, 2, 100]:
            actual = _read(reader, ch, 0, block_size, coll.store_dtype)
            np.testing.assert_array_equal(actual.uv, e['uv'])
            np.testing.assert_array_equal(actual.sub_uv, e['sub_uv'])
            np.testing.assert_array_equal(actual.w_plane, e['w_plane'])
            np.testing.assert_allclose(actual.weights, e['weights'], rtol=1e-6)
            np.testing.assert_allclose(actual.vis, np.array(e['vis']), rtol=1e-5)


def test_empty():
    """test_preprocess.py:71-74."""
    _, _, _, configs, _ = _known()
    coll = _collect_device(configs, 4, 2, [])
    reader = coll.reader()
    for ch in range(2):
        assert reader.len(ch, 0) == 0
        assert list(reader.iter_slice(ch, 0)) == []
        assert list(reader.iter_slice_device(ch, 0)) == []


def _random_batch(rng, C, n, Q, clustered=8):
    base = rng.uniform(-300, 300, (-(-n // clustered), 3)).astype(np.float32)
    uvw = (np.repeat(base, clustered, axis=0)[:n] + rng.normal(0, 0.02, (n, 3))).astype(np.float32)
    weights = rng.uniform(0.5, 2, (C, n, Q)).astype(np.float32)
    for q in range(Q):
        weights[:, rng.random(n) < 0.05, q] = 0
    weights[0, rng.random(n) < 0.01, 0] = -0.0
    vis = (rng.normal(size=(C, n, Q)) + 1j * rng.normal(size=(C, n, Q))).astype(np.complex64)
    vis[:, rng.random(n) < 0.02, Q - 1] = np.nan
    vis[:, rng.random(n) < 0.02, 0] = np.inf
    return uvw, weights, vis


def _compare(coll, ref, configs, exact_float, P):
    reader = coll.reader()
    assert coll.num_input == ref.num_input
    assert coll.num_output == ref.num_output
    for ch, c in enumerate(configs):
        assert reader.num_w_slices(ch) == c['w_slices']
        for s in range(c['w_slices']):
            e = ref.slice_arrays(ch, s)
            assert reader.len(ch, s) == len(e['uv'])
            got = _read(reader, ch, s, 4096, coll.store_dtype)
            # index work: bit-exact
            np.testing.assert_array_equal(got.uv, e['uv'])
            np.testing.assert_array_equal(got.sub_uv, e['sub_uv'])
            np.testing.assert_array_equal(got.w_plane, e['w_plane'])
            if len(e['uv']) == 0:
                continue
            if exact_float:
                np.testing.assert_array_equal(np.asarray(got.weights), e['weights'])
                np.testing.assert_array_equal(np.ascontiguousarray(got.vis).view(np.float32),
                                              e['vis'].view(np.float32))
            else:
                # float32 tolerance 1e-5 (north_star); sincosf differs from glibc by an ulp
                np.testing.assert_allclose(got.weights, e['weights'], rtol=1e-5)
                scale = np.abs(e['vis']).max()
                assert np.abs(np.asarray(got.vis) - e['vis']).max() <= 1e-5 * scale


@pytest.mark.parametrize('P,Q,w_slices,buffer_size', [
    (1, 1, 1, 5000), (1, 2, 5, 3001), (2, 2, 3, 70000), (4, 4, 1, 4096), (4, 4, 7, 9999),
    (3, 4, 2, 20000), (2, 3, 33, 1 << 16)])
def test_random_simple_mueller(P, Q, w_slices, buffer_size):
    """Two add() calls, two channels, flags / -0 weights / NaN / inf visibilities, buffers that
    do not divide the batch: identical records (floats bit for bit) to the restated collector."""
    rng = np.random.default_rng(100 + P * 10 + Q)
    configs = [dict(max_w=320.0, w_slices=w_slices, w_planes=16, oversample=8, cell_size=1.7),
               dict(max_w=300.0, w_slices=w_slices, w_planes=8, oversample=4, cell_size=0.9)]
    stokes = (rng.normal(size=(P, Q)) + 1j * rng.normal(size=(P, Q))).astype(np.complex64)
    if Q > 1:
        stokes[0, 0] = 0      # exercises the MulZ zero rule
    batches = []
    for n in (30011, 12345):
        uvw, weights, vis = _random_batch(rng, 2, n, Q)
        batches.append((uvw, weights, vis, None, None, stokes, None))
    coll = _collect_device(configs, P, buffer_size, batches)
    ref = orc.VisibilityCollector(configs, P, buffer_size)
    for b in batches:
        ref.add(*b)
    assert 0 < ref.num_output < ref.num_input
    _compare(coll, ref, configs, True, P)


@pytest.mark.parametrize('P,Q', [(4, 4), (1, 2), (2, 4)])
def test_random_parallactic(P, Q):
    rng = np.random.default_rng(200 + P + Q)
    configs = [dict(max_w=320.0, w_slices=4, w_planes=16, oversample=8, cell_size=1.7)]
    n = 40000
    uvw, weights, vis = _random_batch(rng, 1, n, Q)
    stokes = (rng.normal(size=(P, 4)) + 1j * rng.normal(size=(P, 4))).astype(np.complex64)
    circ = (rng.normal(size=(4, Q)) + 1j * rng.normal(size=(4, Q))).astype(np.complex64)
    fa1 = rng.uniform(-3, 3, n).astype(np.float32)
    fa2 = rng.uniform(-3, 3, n).astype(np.float32)
    batch = (uvw, weights, vis, fa1, fa2, stokes, circ)
    coll = _collect_device(configs, P, 16384, [batch])
    ref = orc.VisibilityCollector(configs, P, 16384)
    ref.add(*batch)
    _compare(coll, ref, configs, False, P)


def test_long_runs_and_all_flagged():
    """One run spanning a whole buffer, a buffer with nothing valid, and a flagged head."""
    n = 5000
    uvw = np.tile(np.array([[10.0, -7.0, 3.0]], np.float32), (n, 1))
    weights = np.ones((1, n, 1), np.float32)
    weights[0, 0, 0] = 0                      # first record flagged
    weights[0, 2000:3000, 0] = 0              # the whole second buffer flagged
    vis = np.full((1, n, 1), 0.1 + 0.2j, np.complex64)
    configs = [dict(max_w=100.0, w_slices=1, w_planes=4, oversample=8, cell_size=1.0)]
    batch = (uvw, weights, vis, None, None, np.ones((1, 1), np.complex64), None)
    coll = _collect_device(configs, 1, 1000, [batch])
    ref = orc.VisibilityCollector(configs, 1, 1000)
    ref.add(*batch)
    assert ref.num_output == 4
    _compare(coll, ref, configs, True, 1)


def test_device_inputs_and_errors():
    from katsdpimager_amd import accel, preprocess
    ctx, q = context_queue()
    uvw, weights, vis, configs, expected = _known()
    ips, gps = _params(configs, 4)
    coll = preprocess.VisibilityCollectorDevice(q, ips, gps, 64)
    with pytest.raises(RuntimeError):
        coll.reader()                                   # before close()
    d = []
    for a, dt in ((uvw, np.float32), (weights, np.float32), (vis, np.complex64)):
        dev = accel.DeviceArray(ctx, a.shape, dt)
        dev.set(q, a)
        d.append(dev)
    ident = np.identity(4, np.complex64)
    coll.add(d[0], d[1], d[2], None, None, ident, None)
    with pytest.raises(ValueError):
        coll.add(uvw, weights[:, :3], vis, None, None, ident, None)
    with pytest.raises(ValueError):
        coll.add(uvw, weights, vis, np.zeros(4, np.float32), None, ident, None)
    with pytest.raises(ValueError):
        coll.add(uvw, weights, vis, None, None, np.identity(3, np.complex64), None)
    coll.close()
    with pytest.raises(RuntimeError):
        coll.add(uvw, weights, vis, None, None, ident, None)
    reader = coll.reader()
    assert [reader.len(0, 0), reader.len(1, 0)] == [2, 3]
    with pytest.raises(ValueError):
        list(reader.iter_slice_device(0, 0, 0))
    big = list(reader.iter_slice_device(1, 0, 1000))       # larger than the buffer: store regrows
    assert len(big) == 1 and big[0].num_vis == 3 and big[0].uv.shape == (1000, 4)
    np.testing.assert_array_equal(big[0].uv.get(q)[:3, :2], expected[1]['uv'])
    assert not np.any(big[0].vis.get(q)[3:])
    chunks = list(reader.iter_slice_device(1, 0, 2))
    assert [c.num_vis for c in chunks] == [2, 1]
    assert chunks[1].uv.shape == (2, 4) and chunks[1].vis.shape == (2, 4)
    np.testing.assert_array_equal(chunks[0].uv.get(q)[:, :2], expected[1]['uv'][:2])


def test_full_size_buffer():
    """One reference-sized buffer (vis_block = 1 048 576) of a synthetic track, 4 -> 4
    polarizations, against the C restatement; and the store grows across add() calls."""
    rng = np.random.default_rng(7)
    n = 1 << 20
    t = np.arange(n, dtype=np.float64)
    bl = rng.uniform(-2000, 2000, (n // 512 + 1, 3))
    ph = (t % 512) * 2e-5
    b = bl[(t // 512).astype(np.int64)]
    uvw = np.stack([b[:, 0] * np.cos(ph) + b[:, 1] * np.sin(ph),
                    -b[:, 0] * np.sin(ph) + b[:, 1] * np.cos(ph), b[:, 2]], axis=1).astype(np.float32)
    weights = rng.uniform(0.5, 2, (1, n, 4)).astype(np.float32)
    weights[0, rng.random(n) < 0.03, 1] = 0
    vis = (rng.normal(size=(1, n, 4)) + 1j * rng.normal(size=(1, n, 4))).astype(np.complex64)
    configs = [dict(max_w=2000.0, w_slices=3, w_planes=32, oversample=8, cell_size=2.5)]
    stokes = np.array([[1, 0, 0, 1], [0, 1, 1, 0], [0, -1j, 1j, 0], [1, 0, 0, -1]], np.complex64)
    batch = (uvw, weights, vis, None, None, stokes, None)
    coll = _collect_device(configs, 4, n, [batch, batch])
    ref = orc.VisibilityCollector(configs, 4, n)
    ref.add(*batch)
    ref.add(*batch)
    assert ref.num_output < 0.7 * ref.num_input       # the tracks really merge
    _compare(coll, ref, configs, True, 4)


@pytest.mark.parametrize('name', list(gi.E2E_CONFIGS))
def test_store_driven_channel_vs_golden(golden, name):
    """Raw uvw / vis / weights -> device preprocessor -> HBM-resident store -> the major-cycle
    driver (katsdpimager_amd.frontend.process_channel, zero-copy chunks) against the G9 golden
    of the reference's ImagingHost, and against the same driver fed through the host setters."""
    from helpers import make_params, kernel_taper, tapered_relerr, relerr
    from katsdpimager_amd import frontend, imaging, parameters, preprocess, weight
    ctx, q = context_queue()
    c = gi.E2E_CONFIGS[name]
    g = golden('g9_e2e_' + name)
    ip, gp, ap = make_params(c)
    wp = parameters.WeightParameters(weight.WeightType(c['weight_type']), c['robustness'])
    cp = parameters.CleanParameters(c['minor'], c['loop_gain'], c['major_gain'], c['threshold'],
                                    c['mode'], c['psf_cutoff'], c['psf_limit'], c['border'])
    uvw, vis, weights = gi.e2e_raw(c)
    n = len(uvw)
    vis = vis[:, None] if vis.ndim == 1 else vis
    data = gi.e2e_inputs(c)
    for reorder in (False, True):
        coll = preprocess.VisibilityCollectorDevice(q, [ip], [gp], max(n, c['vis_block']),
                                                    reorder=reorder)
        coll.add(uvw, weights[None], vis[None].astype(np.complex64), None, None,
                 np.identity(c['P'], dtype=np.complex64), None)
        coll.close()
        reader = coll.reader()
        for s, rec in enumerate(data['slices']):
            got = _read(reader, 0, s, None, coll.store_dtype)
            if not reorder:
                # in arrival order the store holds exactly the records the goldens were generated from
                np.testing.assert_array_equal(got.uv, rec.uv)
                np.testing.assert_array_equal(got.sub_uv, rec.sub_uv)
                np.testing.assert_array_equal(got.w_plane, rec.w_plane)
                np.testing.assert_array_equal(np.asarray(got.weights), rec.weights)
                np.testing.assert_array_equal(np.ascontiguousarray(got.vis).view(np.float32),
                                              np.ascontiguousarray(rec.vis).view(np.float32))
            else:
                # ... and after the once-per-channel re-order (the default; what the pipeline below
                # runs on) the records the restated re-order makes of them, bit for bit
                e_uv, e_wp, e_w, e_vis = orc.store_reorder(
                    np.concatenate([rec.uv, rec.sub_uv], axis=1), np.ascontiguousarray(rec.w_plane),
                    np.ascontiguousarray(rec.weights), np.ascontiguousarray(rec.vis),
                    c['kernel_width'], c['oversample'], c['w_planes'], True)
                assert reader.len(0, s) == len(e_uv) <= len(rec)
                np.testing.assert_array_equal(np.concatenate([got.uv, got.sub_uv], axis=1), e_uv)
                np.testing.assert_array_equal(got.w_plane, e_wp)
                np.testing.assert_array_equal(np.asarray(got.weights), e_w)
                np.testing.assert_array_equal(np.ascontiguousarray(got.vis).view(np.float32),
                                              e_vis.view(np.float32))
        assert coll.num_stored <= coll.num_output

    class HostReader:
        """iter_slice only: forces the facade's host setters."""
        num_channels = 1

        def num_w_slices(self, channel):
            return reader.num_w_slices(channel)

        def len(self, channel, w_slice):
            return reader.len(channel, w_slice)

        def iter_slice(self, channel, w_slice, block_size=None):
            return reader.iter_slice(channel, w_slice, block_size)

    results = []
    for rd, batched, streams in ((reader, True, 2), (HostReader(), False, 1)):
        template = imaging.ImagingTemplate(ctx, ap, ip.fixed, wp, gp.fixed, cp)
        im = template.instantiate(q, ip, gp, c['vis_block'], 0, c['major'], streams=streams)
        im.ensure_all_bound()
        if not batched:
            # the reference's own sequence of calls throughout: the PSF's peak and patch read back
            # by the host, the first minor cycle on its own (the other run takes the short cuts)
            im.device_psf_stage = False
            im.one_call_major_cycles = False
        stats = frontend.process_channel(rd, 0, im, ip, gp, cp, wp.weight_type, c['vis_block'],
                                         c['major'], c['degrid'], batched_clean=batched)
        results.append((stats, im.get_buffer('dirty'), im.get_buffer('model'),
                        dict(im._model_components)))
    (sa, da, ma, ca), (sb, db, mb, cb) = results
    assert sa['psf_patch'] == sb['psf_patch'] == tuple(g['psf_patch'])
    assert sa['minor'] == sb['minor'] and sa['major'] == sb['major'] == c['major']
    assert sorted(ca) == sorted(cb)                       # same components, bit-exact positions
    G = c['pixels']
    taper = kernel_taper(c)
    inner = np.s_[:, G // 8:-G // 8, G // 8:-G // 8]
    # float atomics make two runs differ in the last bits; the taper division amplifies that at
    # the image edge (see helpers.tapered_relerr)
    assert tapered_relerr(da, db, taper) < 1e-5 and relerr(da[inner], db[inner]) < 1e-4
    assert relerr(ma, mb) < 1e-5
    # against the reference's host pipeline
    np.testing.assert_array_equal(np.array(sorted(ca), np.int64).reshape(-1, 2), g['component_pos'])
    assert tapered_relerr(da, g['dirty_final'], taper) < 2e-4
    assert relerr(da[inner], g['dirty_final'][inner]) < 1e-3
    assert relerr(ma, g['model_final']) < 2e-4
    np.testing.assert_allclose(sa['peaks'][0], g['peak_values'][0], rtol=1e-4)
    # the reference counts minor cycles like the goldens' n_minor (first cycle + loop)
    assert sa['major'] == len(g['n_minor'])


def test_concurrent_channels_match_serial():
    """frontend.process_channels: two channels in flight on two streams (two host threads, one
    command queue each) give the same images as one after the other."""
    from helpers import make_params, relerr
    from katsdpimager_amd import frontend, imaging, parameters, preprocess, weight
    ctx, q = context_queue()
    c = gi.E2E_CONFIGS['degrid']
    ip, gp, ap = make_params(c)
    wp = parameters.WeightParameters(weight.WeightType(c['weight_type']), c['robustness'])
    cp = parameters.CleanParameters(c['minor'], c['loop_gain'], c['major_gain'], c['threshold'],
                                    c['mode'], c['psf_cutoff'], c['psf_limit'], c['border'])
    uvw, vis, weights = gi.e2e_raw(c)
    coll = preprocess.VisibilityCollectorDevice(q, [ip, ip], [gp, gp], len(uvw))
    both = np.stack([vis, 0.5 * vis])[:, :, None].astype(np.complex64)      # two "channels"
    coll.add(uvw, np.stack([weights, weights]), both, None, None, np.ones((1, 1), np.complex64), None)
    coll.close()
    reader = coll.reader()
    template = imaging.ImagingTemplate(ctx, ap, ip.fixed, wp, gp.fixed, cp)

    def jobs():
        out = []
        for ch in range(2):
            qi = ctx.create_command_queue()
            im = template.instantiate(qi, ip, gp, c['vis_block'], 0, c['major'])
            im.ensure_all_bound()
            out.append(dict(reader=reader, rel_channel=ch, imager=im, image_p=ip, grid_p=gp,
                            clean_p=cp, weight_type=wp.weight_type, vis_block=c['vis_block'],
                            major=c['major'], degrid=True))
        return out
    serial, conc, turns = jobs(), jobs(), jobs()
    rs = frontend.process_channels(serial, workers=1)
    rc = frontend.process_channels(conc, workers=2)
    # ... and with the channels taking turns at their throughput-bound stages (CleanBatcher phased)
    rt = frontend.process_channels(turns, workers=2, stagger=True)
    for other, jobs_other in ((rc, conc), (rt, turns)):
        for a, b, ja, jb in zip(rs, other, serial, jobs_other):
            assert a['minor'] == b['minor'] and a['psf_patch'] == b['psf_patch']
            assert sorted(ja['imager']._model_components) == sorted(jb['imager']._model_components)
            assert relerr(ja['imager'].get_buffer('model'), jb['imager'].get_buffer('model')) < 1e-5
    # the two channels differ by the factor put into the visibilities
    m0 = serial[0]['imager'].get_buffer('model')
    m1 = serial[1]['imager'].get_buffer('model')
    assert relerr(0.5 * m0, m1) < 1e-3
    with pytest.raises(ValueError):
        shared = jobs()
        shared[1]['imager'] = shared[0]['imager']
        frontend.process_channels(shared, workers=2)


def test_example_runs_and_recovers_sources():
    """examples/image_channel.py at a small size: the three synthetic sources come back at their
    positions with their fluxes (robust weighting, 3 major cycles, restored with the fitted beam)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        'examples', 'image_channel.py')
    spec = importlib.util.spec_from_file_location('image_channel_example', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    restored, stats = mod.main(['--pixels', '1024', '--vis', '600000', '--major', '3',
                                '--minor', '300', '--vis-block', '131072'])
    assert stats['major'] >= 2 and stats['minor'] > 10
    G = 1024
    # a Gaussian restoring beam of amplitude 1 keeps the peak = flux
    assert 1.0 < stats['restoring_beam'].minor <= stats['restoring_beam'].major < 20.0
    for (lp, mp), flux in [((40, -25), 1.0), ((-120, 60), 0.5), ((15, 200), 0.25)]:
        y, x = G // 2 + mp, G // 2 + lp
        peak = float(restored[y - 2:y + 3, x - 2:x + 3].max())
        assert abs(peak - flux) < 0.1 * flux, (lp, mp, flux, peak)


def test_continuum_subtraction_removes_the_sources():
    """frontend.make_dirty(..., subtract_model=True) (frontend.py:135-136): the façade's
    continuum predictor, fed the true source list with `set_sky_arrays`, removes the three
    synthetic sources from the dirty image; and it equals gridding the residuals that the
    oracle's predictor computes from the same records."""
    from helpers import make_params, kernel_taper, tapered_relerr
    from katsdpimager_amd import frontend, imaging, parameters, preprocess, weight
    ctx, q = context_queue()
    c = gi.E2E_CONFIGS['predict']
    ip, gp, ap = make_params(c)
    wp = parameters.WeightParameters(weight.WeightType.NATURAL, 0.0)
    cp = parameters.CleanParameters(c['minor'], c['loop_gain'], c['major_gain'], c['threshold'],
                                    c['mode'], c['psf_cutoff'], c['psf_limit'], c['border'])
    uvw, vis, weights = gi.e2e_raw(c)
    coll = preprocess.VisibilityCollectorDevice(q, [ip], [gp], len(uvw))
    coll.add(uvw, weights[None], vis[None, :, None].astype(np.complex64), None, None,
             np.ones((1, 1), np.complex64), None)
    coll.close()
    reader = coll.reader()
    # the sources of golden_inputs.e2e_raw: (l, m) in pixels and flux; vis carries flux / n
    src_lm = np.array([[20, -33], [-41, 12], [5, 60]]) * c['pixel_size']
    n = np.sqrt(1 - np.sum(src_lm ** 2, axis=1))
    lmn = np.concatenate([src_lm, (n - 1)[:, None]], axis=1).astype(np.float32)
    flux = (np.array([1.0, 0.6, 0.35]) / n)[:, None].astype(np.float32)
    im = imaging.ImagingTemplate(ctx, ap, ip.fixed, wp, gp.fixed, cp).instantiate(
        q, ip, gp, c['vis_block'], 3, c['major'])
    im.ensure_all_bound()
    im.set_sky_arrays(lmn, flux)
    mid_w = frontend.slice_mid_w(ip, gp)
    frontend.make_weights(reader, 0, im, wp.weight_type, c['vis_block'])
    frontend.make_dirty(reader, 0, 'vis', im, mid_w, c['vis_block'], False)
    with_sources = im.get_buffer('dirty')
    frontend.make_dirty(reader, 0, 'vis', im, mid_w, c['vis_block'], False, subtract_model=True)
    subtracted = im.get_buffer('dirty')
    assert np.abs(subtracted).max() < 0.1 * np.abs(with_sources).max()

    # expectation: oracle predict on the stored records, gridded by the same device path
    uv_scale, w_scale, w_bias = orc.uvw_scale_bias(c['cell_size'], c['wavelength'], c['max_w'],
                                                   c['w_slices'], c['w_planes'], c['oversample'])

    class Residuals:
        """Host reader whose visibilities already have the model subtracted."""
        def num_w_slices(self, ch):
            return reader.num_w_slices(ch)

        def len(self, ch, s):
            return reader.len(ch, s)

        def iter_slice(self, ch, s, block):
            for rec in reader.iter_slice(ch, s, block):
                rec = rec.copy()
                v = np.ascontiguousarray(rec.vis)
                orc.predict(v, rec.uv, rec.sub_uv, rec.w_plane, np.ascontiguousarray(rec.weights),
                            lmn, flux, c['oversample'], uv_scale, w_scale,
                            w_bias + mid_w[s])
                rec.vis = v
                yield rec
    frontend.make_dirty(Residuals(), 0, 'vis', im, mid_w, c['vis_block'], False)
    expected = im.get_buffer('dirty')
    taper = kernel_taper(c)
    scale = np.abs(with_sources).max() / np.abs(expected).max()
    # the difference is the float32 phase error of the predicted part (2e-3 of the sources)
    assert tapered_relerr(subtracted, expected, taper) < 2e-3 * scale


@pytest.mark.parametrize('seed', range(8))
def test_preprocess_fuzz(seed):
    """Seeded random collector problems (polarization counts, slices, planes, oversampling, buffer
    sizes, channel counts, run lengths, flag / NaN rates, with and without feed angles) against the
    restated collector: index fields bit-exact, floats bit-exact without feed angles."""
    rng = np.random.default_rng(9000 + seed)
    P = int(rng.integers(1, 5))
    Q = int(rng.integers(max(P - 1, 1), 5)) if rng.random() < 0.5 else int(rng.integers(1, 5))
    C = int(rng.integers(1, 4))
    n = int(rng.choice([1, 2, 63, 64, 65, 1000, 7777, 30000]))
    w_slices = int(rng.choice([1, 2, 5, 17]))
    configs = [dict(max_w=float(rng.uniform(100, 500)), w_slices=w_slices,
                    w_planes=int(rng.choice([1, 8, 33])), oversample=int(rng.choice([1, 4, 8])),
                    cell_size=float(rng.uniform(0.5, 3.0))) for _ in range(C)]
    buffer_size = int(rng.choice([1, 7, 64, 1000, 4096, 50000]))
    if n / buffer_size > 400:
        buffer_size = 4096
    clustered = int(rng.choice([1, 3, 20, 500]))
    base = rng.uniform(-300, 300, (-(-n // clustered), 3)).astype(np.float32)
    uvw = (np.repeat(base, clustered, axis=0)[:n] + rng.normal(0, 0.02, (n, 3))).astype(np.float32)
    weights = rng.uniform(0.5, 2, (C, n, Q)).astype(np.float32)
    weights[:, rng.random(n) < rng.choice([0.0, 0.1, 0.9]), int(rng.integers(0, Q))] = 0
    vis = (rng.normal(size=(C, n, Q)) + 1j * rng.normal(size=(C, n, Q))).astype(np.complex64)
    vis[:, rng.random(n) < 0.03, int(rng.integers(0, Q))] = np.nan
    feed = rng.random() < 0.3
    if feed:
        stokes = (rng.normal(size=(P, 4)) + 1j * rng.normal(size=(P, 4))).astype(np.complex64)
        circ = (rng.normal(size=(4, Q)) + 1j * rng.normal(size=(4, Q))).astype(np.complex64)
        fa1 = rng.uniform(-3, 3, n).astype(np.float32)
        fa2 = rng.uniform(-3, 3, n).astype(np.float32)
        batch = (uvw, weights, vis, fa1, fa2, stokes, circ)
    else:
        stokes = (rng.normal(size=(P, Q)) + 1j * rng.normal(size=(P, Q))).astype(np.complex64)
        batch = (uvw, weights, vis, None, None, stokes, None)
    coll = _collect_device(configs, P, buffer_size, [batch])
    ref = orc.VisibilityCollector(configs, P, buffer_size)
    ref.add(*batch)
    _compare(coll, ref, configs, not feed, P)


def test_loader_to_store_in_loader_order():
    """loader.LoaderArrays (time-ordered rows -> baseline-sorted blocks, as the
    reference's loaders deliver them) -> loader.preprocess_visibilities -> device collector: the
    same records, bit for bit, as the restated collector fed with the same blocks; sorting the
    blocks by baseline is what lets the adjacent-merge compression find its runs."""
    from katsdpimager_amd import loader, preprocess
    import test_host_logic as th
    ds, nb, dumps = th._loader_arrays(rows=40000, channels=2, pols=2, antennas=12, seed=3)
    # slow tracks: a baseline moves a small fraction of a cell per dump, so that consecutive
    # dumps of a baseline often fall on the same sub-cell
    rs = np.random.RandomState(4)
    base = rs.uniform(-250, 250, (nb, 3)).astype(np.float32)
    drift = rs.uniform(-0.02, 0.02, (nb, 3)).astype(np.float32)
    t = np.repeat(np.arange(dumps, dtype=np.float32), nb)
    b = np.tile(np.arange(nb), dumps)
    ds.uvw[:] = base[b] + drift[b] * t[:, None]
    configs = [dict(max_w=320.0, w_slices=2, w_planes=16, oversample=8, cell_size=1.7),
               dict(max_w=300.0, w_slices=2, w_planes=8, oversample=4, cell_size=0.9)]
    ctx, q = context_queue()
    ips, gps = _params(configs, 2)
    ident = np.identity(2, np.complex64)
    vis_load = 2 * 4000                    # rows x channels per block
    coll = preprocess.VisibilityCollectorDevice(q, ips, gps, 4096)
    loader.preprocess_visibilities(ds, coll, 0, 2, (ident, None), vis_load=vis_load)
    ref = orc.VisibilityCollector(configs, 2, 4096)
    blocks = 0
    for chunk in ds.data_iter(0, 2, vis_load):
        ref.add(chunk['uvw'], chunk['weights'], chunk['vis'], None, None, ident, None)
        blocks += 1
    assert blocks == 10
    _compare(coll, ref, configs, True, 2)
    with pytest.raises(RuntimeError):
        coll.add(ds.uvw[:4], np.swapaxes(ds.weights[:4], 0, 1), np.swapaxes(ds.vis[:4], 0, 1),
                 None, None, ident, None)           # closed by preprocess_visibilities
    # the same rows in time order (no baseline sort): far fewer merges
    unsorted = orc.VisibilityCollector(configs, 2, 4096)
    unsorted.add(ds.uvw, np.ascontiguousarray(np.swapaxes(ds.weights, 0, 1)),
                 np.ascontiguousarray(np.swapaxes(ds.vis, 0, 1)), None, None, ident, None)
    assert ref.num_output < 0.6 * unsorted.num_output


# ---- the resident store's once-per-channel re-order (csrc/store.hip) -------------------------------
def _arrival_stream(rs, n_track, n_scatter, P, W, OV, extent, repeat):
    """Baseline-sorted blocks of slowly moving tracks (the loaders' shape: consecutive records often
    share a sub-cell, and a track comes back to the same cells in a later block) plus scattered
    records, as (uv4 int16 [N][4], w_plane, weights, vis)."""
    tracks = 40
    per = max(n_track // (tracks * repeat), 1)
    pieces = []
    start = rs.uniform(-extent, extent, (tracks, 2))
    speed = rs.uniform(-0.2, 0.2, (tracks, 2))
    for block in range(repeat):
        for t in range(tracks):
            pos = start[t] + speed[t] * (np.arange(per) + 0.25 * per * block)[:, None]
            pieces.append(np.clip(pos, -extent, extent))
    pos = np.concatenate(pieces + [rs.uniform(-extent, extent, (n_scatter, 2))])
    fine = np.floor(pos * OV).astype(np.int64)
    uv4 = np.concatenate([fine // OV, fine % OV], axis=1).astype(np.int16)
    n = len(uv4)
    w_plane = rs.randint(0, W, n).astype(np.int16)
    w_plane[:n_track] = (np.arange(n_track)[:len(w_plane[:n_track])] // 97) % W
    weights = rs.uniform(0.5, 2.0, (n, P)).astype(np.float32)
    vis = (rs.standard_normal((n, P)) + 1j * rs.standard_normal((n, P))).astype(np.complex64)
    return uv4, w_plane, weights, vis


@pytest.mark.parametrize('merge', [False, True])
@pytest.mark.parametrize('K,P,W,OV', [(28, 1, 32, 8), (8, 2, 4, 4), (32, 1, 3, 8), (60, 4, 16, 8),
                                      (45, 3, 33, 5), (1, 1, 1, 1)])
def test_store_reorder_vs_restatement(K, P, W, OV, merge):
    """kimg_store_reorder through the C ABI against oracle.store_reorder: the same order (strips of
    window-slack + 1 grid columns, swept along v, serpentine; stable) and, with merge, the same
    records with the same float32 sums, bit for bit; a second call on its own output changes
    nothing; an empty slice is fine."""
    from katsdpimager_amd import accel
    from katsdpimager_amd._lib import lib, check
    ctx, q = context_queue()
    rs = np.random.RandomState(K * 7 + P)
    uv4, w_plane, weights, vis = _arrival_stream(rs, 30000, 5000, P, W, OV, 300, 3)
    n = len(uv4)
    L = lib()

    def run(uv4, w_plane, weights, vis):
        n = len(uv4)
        d = [accel.DeviceArray(ctx, a.shape, a.dtype) for a in (uv4, w_plane, weights, vis)]
        for dev, a in zip(d, (uv4, w_plane, weights, vis)):
            dev.set(q, a)
        o = [accel.DeviceArray(ctx, a.shape, a.dtype) for a in (uv4, w_plane, weights, vis)]
        count = accel.DeviceArray(ctx, (1,), np.int64)
        ws_bytes = int(L.kimg_store_reorder_workspace_bytes(n))
        assert ws_bytes > 0
        ws = accel.DeviceArray(ctx, (ws_bytes,), np.uint8)
        assert L.kimg_store_reorder(P, n, K, OV, W, int(merge), d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr,
                                    o[0].ptr, o[1].ptr, o[2].ptr, o[3].ptr, count.ptr, ws.ptr,
                                    ws_bytes - 1, q.handle) == -10003        # KIMG_EWORKSPACE
        check(L.kimg_store_reorder(P, n, K, OV, W, int(merge), d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr,
                                   o[0].ptr, o[1].ptr, o[2].ptr, o[3].ptr, count.ptr, ws.ptr,
                                   ws_bytes, q.handle), 'kimg_store_reorder')
        m = int(count.get(q)[0])
        return [a.get(q)[:m] for a in o]
    got = run(uv4, w_plane, weights, vis)
    want = orc.store_reorder(uv4, w_plane, weights, vis, K, OV, W, merge)
    assert len(got[0]) == len(want[0]) and (len(want[0]) < n) == (merge and K != 1 or merge)
    for g, w_ in zip(got, want):
        np.testing.assert_array_equal(g.view(np.uint8), np.ascontiguousarray(w_).view(np.uint8))
    # the order itself: strips ascend, v runs up in even strips and down in odd ones
    width = orc.store_strip_width(K)
    strip = (got[0][:, 0].astype(np.int64) + 32768) // width
    assert np.all(np.diff(strip) >= 0)
    same = np.diff(strip) == 0
    dv = np.diff(got[0][:, 1].astype(np.int64))
    assert np.all(np.where(strip[1:] & 1, -dv, dv)[same] >= 0)
    again = run(*got)
    # (a run that straddles a multiple of STORE_MAX_RUN sorted positions is stored as two records,
    # which a second pass may find side by side: only then does it change anything)
    coords = np.concatenate([got[0], got[1][:, None]], axis=1)
    if not merge or len(np.unique(coords, axis=0)) == len(coords):
        for g, a in zip(got, again):
            np.testing.assert_array_equal(g, a)
    else:
        assert len(again[0]) < len(got[0])
        np.testing.assert_allclose(again[2].sum(axis=0), got[2].sum(axis=0), rtol=1e-5)
    count = accel.DeviceArray(ctx, (1,), np.int64)
    count.set(q, np.array([7], np.int64))
    check(L.kimg_store_reorder(P, 0, K, OV, W, int(merge), None, None, None, None, None, None, None,
                               None, count.ptr, None, 0, q.handle), 'kimg_store_reorder')
    assert int(count.get(q)[0]) == 0
    assert L.kimg_store_reorder(5, 10, K, OV, W, 0, None, None, None, None, None, None, None, None,
                                count.ptr, None, 0, q.handle) == -10002


@pytest.mark.parametrize('merge', [False, True])
@pytest.mark.parametrize('arith', ['fp32', 'split_fp16'])
def test_reordered_store_grids_and_degrids_like_the_arrival_stream(arith, merge):
    """A store re-ordered on close() against the SAME visibilities in arrival order through the
    oracle: the grid to 1e-5 (`orc.grid` on the un-reordered stream), the density-weight grid
    exactly, and the degridder's residuals -- summed over the records a stored record was merged
    from -- to 1e-5; the window kernels take the stored order as it is (`locality` True)."""
    from helpers import make_params, relerr
    from katsdpimager_amd import accel, grid, preprocess, weight
    ctx, q = context_queue()
    c = gi.make_config(1024, 0.0001, 0.01, 2, 28, 16, grid_cover=700, n_vis=0)
    ip, gp, ap = make_params(c)
    rs = np.random.RandomState(5)
    # raw uvw in metres: slow tracks in baseline-sorted blocks + scattered positions
    cell = float(ip.cell_size)
    uv4, w_plane, weights, vis = _arrival_stream(rs, 60000, 10000, 2, 16, 8, 330, 4)
    n = len(uv4)
    uvw = np.zeros((n, 3), np.float32)
    uvw[:, 0] = (uv4[:, 0] + (uv4[:, 2] + 0.5) / 8) * cell
    uvw[:, 1] = (uv4[:, 1] + (uv4[:, 3] + 0.5) / 8) * cell
    uvw[:, 2] = rs.uniform(0, 0.9 * c['max_w'], n)
    ident = np.identity(2, np.complex64)
    colls = {}
    for reorder in (False, True):
        coll = preprocess.VisibilityCollectorDevice(q, [ip], [gp], 16384, reorder=reorder, merge=merge)
        coll.add(uvw, weights[None], vis[None], None, None, ident, None)
        coll.close()
        colls[reorder] = coll
    plain, stored = colls[False].reader(), colls[True].reader()
    a = _read(plain, 0, 0, None, colls[False].store_dtype)
    b = _read(stored, 0, 0, None, colls[True].store_dtype)
    assert (len(b) < len(a)) if merge else (len(b) == len(a))
    assert colls[True].num_output == colls[False].num_output and colls[True].num_stored == len(b)
    # gridder
    max_vis = len(a)
    g = grid.GridderTemplate(ctx, ip.fixed, gp.fixed, {'arith': arith}).instantiate(q, ap, ip, gp, max_vis)
    g.ensure_all_bound()
    Gg = g.buffer('grid').shape[1]
    wgrid = rs.uniform(0.5, 1.5, (2, Gg, Gg)).astype(np.float32)
    g.buffer('weights_grid').set(q, wgrid)
    g.buffer('grid').zero(q)
    chunks = list(stored.iter_slice_device(0, 0, max_vis))
    assert len(chunks) == 1 and chunks[0].locality is True and chunks[0].num_vis == len(b)
    ch = chunks[0]
    g.bind(uv=ch.uv, w_plane=ch.w_plane, vis=ch.vis)
    g.num_vis = ch.num_vis
    g.locality_hint = ch.locality
    g()
    assert g.last_variant == 'mfma'
    got = g.buffer('grid').get(q)
    want = np.zeros_like(got)
    orc.grid(g.convolve_kernel.data, want, wgrid, np.ascontiguousarray(a.uv), np.ascontiguousarray(a.sub_uv),
             np.ascontiguousarray(a.w_plane), np.ascontiguousarray(a.vis))
    assert relerr(got, want) < 1e-5
    # degridder: residual of a stored record = sum of the residuals of its members
    model = (rs.standard_normal((2, Gg, Gg)) + 1j * rs.standard_normal((2, Gg, Gg))).astype(np.complex64)
    d = grid.DegridderTemplate(ctx, ip.fixed, gp.fixed, {'arith': arith}).instantiate(q, ap, ip, gp, max_vis)
    d.ensure_all_bound()
    d.buffer('grid').set(q, model)
    resid = accel.DeviceArray(ctx, ch.vis.shape, np.complex64)
    ch.vis.copy_region(q, resid, np.s_[:ch.num_vis], np.s_[:ch.num_vis])
    d.bind(uv=ch.uv, w_plane=ch.w_plane, weights=ch.weights, vis=resid)
    d.num_vis = ch.num_vis
    d.locality_hint = ch.locality
    d()
    got_r = resid.get(q)[:ch.num_vis]
    ref_r = np.ascontiguousarray(a.vis).copy()
    orc.degrid(d.convolve_kernel.data, model, np.ascontiguousarray(a.uv), np.ascontiguousarray(a.sub_uv),
               np.ascontiguousarray(a.w_plane), np.ascontiguousarray(a.weights), ref_r)
    # sum the arrival-order residuals over equal coordinates, in the stored order
    key = lambda r: np.concatenate([r.uv, r.sub_uv, r.w_plane[:, None]], axis=1).astype(np.int64)
    ka, kb = key(a), key(b)
    packed = lambda k: ((((k[:, 0] + 32768) * 65536 + (k[:, 1] + 32768)) * 8 + k[:, 2]) * 8 + k[:, 3]) * 64 + k[:, 4]
    pa, pb = packed(ka), packed(kb)
    if merge:
        # (one record per coordinate, but for the runs cut at a multiple of STORE_MAX_RUN sorted
        # positions, which are stored as two: the sums per coordinate are compared)
        uniq_b, inv_b = np.unique(pb, return_inverse=True)
        assert len(pb) - len(uniq_b) <= len(pa) // orc.STORE_MAX_RUN + 1
        uniq, inv = np.unique(pa, return_inverse=True)
        summed = np.zeros((len(uniq), 2), np.complex128)
        np.add.at(summed, inv, ref_r.astype(np.complex128))
        got_sum = np.zeros((len(uniq_b), 2), np.complex128)
        np.add.at(got_sum, inv_b, got_r.astype(np.complex128))
        want_r = summed[np.searchsorted(uniq, uniq_b)]
        assert np.abs(got_sum - want_r).max() <= 1e-5 * np.abs(want_r).max()
    else:
        # a permutation: sort both by (coordinates, visibility bits) and compare record for record
        ia = np.lexsort((np.ascontiguousarray(a.vis[:, 0]).view(np.uint64), pa))
        ib = np.lexsort((np.ascontiguousarray(b.vis[:, 0]).view(np.uint64), pb))
        np.testing.assert_array_equal(pa[ia], pb[ib])
        assert np.abs(got_r[ib] - ref_r[ia]).max() <= 1e-5 * np.abs(ref_r).max()
    # density weights see the same sums per cell
    wt = weight.WeightsTemplate(ctx, weight.WeightType.UNIFORM, 2).instantiate(q, (2, Gg, Gg), max_vis)
    wt.ensure_all_bound()
    grids = []
    for rd in (plain, stored):
        wt.clear()
        for chunk in rd.iter_slice_device(0, 0, max_vis):
            wt.bind(uv=chunk.uv, weights=chunk.weights)
            wt.grid(chunk.num_vis)
        grids.append(wt.buffer('grid').get(q))
    assert relerr(grids[1], grids[0]) < 2e-6
