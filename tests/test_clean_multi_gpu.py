"""GPU parity of the multi-component CLEAN launch (csrc/clean_multi.hip, KIMG_CLEAN_FORM_MULTI)
through the C ABI: components (metric, position, flux), residual image, model image and the tile
arrays left behind are BIT-EXACT against the restated CleanHost (oracle/kimg_oracle.Clean,
clean.py:1060-1075), for every cap on the components per launch."""
import numpy as np
import pytest

from helpers import context_queue
from oracle import kimg_oracle as orc
from test_clean_multi_model import dominated_problem, fuzz_problem, reference_run, sources_problem

pytestmark = pytest.mark.gpu


def _clean(G, P, mode, border, loop_gain, dirty, psf, tuning):
    from katsdpimager_amd import clean, parameters
    ctx, q = context_queue()
    fixed = parameters.FixedImageParameters(list(range(P)), np.float32)
    ip = parameters.ImageParameters(fixed, 1.0, None, 0.2, None, pixel_size=1e-5, pixels=G)
    cp = parameters.CleanParameters(1000, loop_gain, 0.85, 5.0, mode, 0.01, 0.5, border)
    fn = clean.CleanTemplate(ctx, cp, np.float32, P, tuning).instantiate(q, ip)
    fn.ensure_all_bound()
    fn.buffer('dirty').set(q, dirty)
    fn.buffer('psf').set(q, psf)
    fn.buffer('model').zero(q)
    fn.reset()
    return fn, q


def _check(fn, q, got, want):
    log, img, model, tile_max, tile_pos = want
    assert len(got) == len(log)
    for a, b in zip(got, log):
        assert a[0] == b[0] and tuple(a[1]) == tuple(b[1]), (a, b)
        np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(fn.buffer('dirty').get(q), img)
    np.testing.assert_array_equal(fn.buffer('model').get(q), model)
    np.testing.assert_array_equal(fn.buffer('tile_max').get(q), tile_max)
    np.testing.assert_array_equal(fn.buffer('tile_pos').get(q), tile_pos)


@pytest.mark.parametrize('components', [0, 1, 2, 5])
@pytest.mark.parametrize('seed', range(8))
def test_multi_fuzz(seed, components):
    """The problems of test_clean_fuzz (odd shapes, 1-4 polarizations, both metrics, borders,
    patches from one pixel to larger than the image, thresholds that stop the loop)."""
    rs, P, mode, G, border, loop_gain, psf, dirty, patch, cycles = fuzz_problem(seed)
    fn, q = _clean(G, P, mode, border, loop_gain, dirty, psf, {'form': 'multi', 'components': components})
    first = float(np.max(fn.buffer('tile_max').get(q)))
    threshold = float(rs.choice([0.0, 0.3 * first, 2.0 * first]))
    want = reference_run(G, border, loop_gain, mode, dirty, psf, patch, threshold, cycles)
    got = fn.run_cycles(patch, threshold, cycles)
    _check(fn, q, got, want)


@pytest.mark.parametrize('G,P,mode,border,patch,n_sources,cycles,components', [
    (512, 1, 0, 0.02, (33, 47), 60, 300, 0),
    (512, 1, 0, 0.02, (33, 47), 60, 300, 3),
    (768, 1, 0, 0.02, (111, 133), 40, 250, 0),       # 5 x 6 lattice blocks: the bench's patch
    (640, 2, 1, 0.05, (65, 65), 50, 200, 0),         # 4 x 4 lattice blocks, sum of squares
    (512, 4, 1, 0.0, (31, 95), 60, 150, 0),
    (512, 4, 0, 0.1, (95, 31), 60, 150, 4),
    (1024, 1, 0, 0.02, (161, 191), 50, 200, 0),      # 7 x 7 = 49 blocks: four components per launch
    (1024, 1, 0, 0.02, (225, 321), 50, 120, 0),      # 9 x 12 = 108 blocks: two per launch
    (1536, 1, 0, 0.02, (351, 415), 60, 100, 0),      # 12 x 14 = 168 blocks: one per launch
    (1024, 1, 0, 0.013, (65, 97), 120, 400, 0),      # border 13
])
def test_multi_many_sources(G, P, mode, border, patch, n_sources, cycles, components):
    """Many sources of similar brightness: launches plan several components, some of them in
    vain (neighbouring sources, repeated peaks); the result is the reference's all the same."""
    rs, psf, dirty = sources_problem(G + P, G=G, P=P, n_sources=n_sources)
    fn, q = _clean(G, P, mode, border, 0.1, dirty, psf, {'form': 'multi', 'components': components})
    full = (P,) + patch
    want = reference_run(G, border, 0.1, mode, dirty, psf, full, 0.0, cycles)
    got = fn.run_cycles(full, 0.0, cycles)
    _check(fn, q, got, want)
    launches = fn.last_launches()
    assert launches is not None and launches >= 1
    if components != 1 and patch[0] * patch[1] < 100 * 140:
        assert launches < 0.7 * cycles, launches     # several components per launch did happen


@pytest.mark.parametrize('P,mode', [(1, 0), (2, 1)])
def test_multi_bench_image_vs_oracle(P, mode):
    """The bench's CLEAN problem at its true size -- 4096^2, 200 sources of similar brightness
    convolved with the PSF + noise, 133 x 111 patch, 1000 cycles, i.e. about 140 launches of 7
    components, a handful of them with the list of best tiles rebuilt -- and a bright source on top
    that takes the first few dozen cycles alone (every plan beyond it is mispredicted): components,
    residual image, model and tile arrays against the restated CleanHost, bit for bit."""
    G = 4096
    rs = np.random.RandomState(4)
    g1 = np.exp(-0.5 * ((np.arange(G) - G // 2) / 6.0) ** 2).astype(np.float32)
    psf = np.outer(g1, g1)[None].repeat(P, axis=0).astype(np.float32)
    psf += (0.002 * rs.standard_normal(psf.shape)).astype(np.float32)
    psf[:, G // 2, G // 2] = 1.0
    sky = (0.01 * rs.standard_normal((P, G, G))).astype(np.float32)
    for i in range(200):
        y, x = rs.randint(100, G - 100, 2)
        amp = rs.uniform(0.5, 2.0) if i else 30.0
        sky[:, y - 30:y + 31, x - 30:x + 31] += amp * psf[:, G // 2 - 30:G // 2 + 31, G // 2 - 30:G // 2 + 31]
    fn, q = _clean(G, P, mode, 0.02, 0.1, sky, psf, None)
    patch = (P, 111, 133)
    want = reference_run(G, 0.02, 0.1, mode, sky, psf, patch, 0.0, 1000)
    got = fn.run_cycles(patch, 0.0, 1000)
    _check(fn, q, got, want)
    launches = fn.last_launches()
    assert launches is not None and launches < 300, launches


def test_multi_continues_and_stops():
    """Consecutive calls continue where the last one stopped; thresholds and cycle limits stop the
    loop at exactly the reference's component, whatever had been planned beyond it."""
    G, P = 512, 1
    rs, psf, dirty = sources_problem(3, G=G, P=P, n_sources=50)
    fn, q = _clean(G, P, 0, 0.02, 0.1, dirty, psf, {'form': 'multi'})
    patch = (1, 47, 33)
    img, model = dirty.copy(), np.zeros_like(dirty)
    ref = orc.Clean(G, 0.02, 0.1, 0, img, psf, model)
    ref.reset()
    first = float(np.max(ref._tile_max))
    for cycles, threshold in ((1, 0.0), (2, 0.0), (3, 0.0), (7, 0.0), (64, 0.0), (500, 0.8 * first),
                              (500, 0.8 * first), (33, 0.5 * first), (500, 0.6 * first)):
        want = []
        for _ in range(cycles):
            v, pos, pix = ref(patch, threshold)
            if v is None:
                break
            want.append((v, ref.last_pos, np.array(pix)))
        got = fn.run_cycles(patch, threshold, cycles)
        _check(fn, q, got, (want, img, model, ref._tile_max, ref._tile_pos))


def test_multi_edge_images():
    """An all-zero image (the (x0, y0) start position of clean.py:950 wins every cycle), a constant
    one (every tile ties: the lowest tile index wins; more equal maxima than the keeper lists), and
    exact ties between distant pixels."""
    G = 160
    psf = np.zeros((1, G, G), np.float32)
    psf[0, G // 2 - 2:G // 2 + 3, G // 2 - 2:G // 2 + 3] = 0.5
    psf[0, G // 2, G // 2] = 1.0
    tie = np.zeros((1, G, G), np.float32)
    tie[0, 20, 130] = tie[0, 100, 30] = tie[0, 100, 31] = tie[0, 140, 140] = 3.0
    tie[0, 60, 60] = -3.0
    for dirty in (np.zeros((1, G, G), np.float32), np.full((1, G, G), 0.75, np.float32), tie):
        for border in (0.0, 0.05):
            fn, q = _clean(G, 1, 0, border, 0.3, dirty, psf, {'form': 'multi'})
            want = reference_run(G, border, 0.3, 0, dirty, psf, (1, 5, 5), 0.0, 40)
            got = fn.run_cycles((1, 5, 5), 0.0, 40)
            _check(fn, q, got, want)


def test_multi_is_what_auto_takes():
    """`auto` takes the multi-component form when the patch leaves room for two lattices."""
    G = 512
    rs, psf, dirty = sources_problem(11, G=G, P=1, n_sources=60)
    fn, q = _clean(G, 1, 0, 0.02, 0.1, dirty, psf, None)
    want = reference_run(G, 0.02, 0.1, 0, dirty, psf, (1, 33, 47), 0.0, 200)
    got = fn.run_cycles((1, 33, 47), 0.0, 200)
    _check(fn, q, got, want)
    assert fn.last_launches() is not None and fn.last_launches() < 140


@pytest.mark.parametrize('seed', range(8))
@pytest.mark.parametrize('repeats,always', [(0, True), (2, True), (0, False)])
def test_multi_fuzz_repeated_steps(seed, repeats, always):
    """The fuzz problems through the repeated-steps kernel (from the first launch on, or as the
    loop's own choice makes it alternate with the single-step kernel)."""
    rs, P, mode, G, border, loop_gain, psf, dirty, patch, cycles = fuzz_problem(40 + seed)
    fn, q = _clean(G, P, mode, border, loop_gain, dirty, psf,
                   {'form': 'multi', 'repeats': repeats, 'repeats_always': always})
    first = float(np.max(fn.buffer('tile_max').get(q)))
    threshold = float(rs.choice([0.0, 0.3 * first, 2.0 * first]))
    want = reference_run(G, border, loop_gain, mode, dirty, psf, patch, threshold, cycles)
    got = fn.run_cycles(patch, threshold, cycles)
    _check(fn, q, got, want)


@pytest.mark.parametrize('G,P,mode,patch,amplitudes,cycles', [
    (512, 1, 0, (33, 47), (30.0,), 150),
    (512, 1, 0, (33, 47), (30.0, 12.0, 11.0), 250),
    (768, 1, 0, (111, 133), (50.0, 20.0), 200),       # the bench's patch
    (512, 2, 1, (65, 65), (20.0, 15.0), 150),        # sum of squares: four steps per launch
    (512, 4, 0, (31, 95), (25.0,), 120),
    (1024, 1, 0, (161, 191), (40.0, 10.0), 150),     # 49 blocks: four lattices per launch
])
def test_multi_dominated_field(G, P, mode, patch, amplitudes, cycles):
    """A few sources far above the rest: a launch steps the same peaks several times (the value at a
    peak follows a scalar recursion that every workgroup evaluates), and far fewer launches commit
    the reference's components, bit for bit."""
    rs, psf, dirty = dominated_problem(G + P, G=G, P=P, n_sources=30, amplitudes=amplitudes)
    full = (P,) + patch
    want = reference_run(G, 0.02, 0.1, mode, dirty, psf, full, 0.0, cycles)
    launches = {}
    for name, tuning in (('single', {'form': 'multi', 'repeats': 1}),
                         ('repeated', {'form': 'multi', 'repeats_always': True}),
                         ('auto', {'form': 'multi'})):
        fn, q = _clean(G, P, mode, 0.02, 0.1, dirty, psf, tuning)
        got = fn.run_cycles(full, 0.0, cycles)
        _check(fn, q, got, want)
        launches[name] = fn.last_launches()
    assert launches['repeated'] < 0.75 * launches['single'], launches
    # (the loop's own choice starts with single steps and changes over only for a phase that lasts)
    assert launches['auto'] <= launches['single'] + 8, launches


@pytest.mark.parametrize('seed', range(4))
def test_multi_repeated_steps_that_fail(seed):
    """Bright single pixels under a PSF with a skirt: the steps planned at such a peak do not hold
    (its neighbours soon beat it); what was committed of them is written, the lattice evaluated
    again, and the result is the reference's."""
    rs, psf, dirty = dominated_problem(20 + seed, n_sources=[3, 10, 30][seed % 3],
                                       amplitudes=(5.0, 9.0)[:1 + seed % 2], shaped=False)
    gain = [0.1, 0.3][seed % 2]
    want = reference_run(384, 0.02, gain, 0, dirty, psf, (1, 33, 47), 0.0, 150)
    for tuning in ({'form': 'multi', 'repeats_always': True}, {'form': 'multi'}):
        fn, q = _clean(384, 1, 0, 0.02, gain, dirty, psf, tuning)
        got = fn.run_cycles((1, 33, 47), 0.0, 150)
        _check(fn, q, got, want)


def test_multi_repeated_steps_continue_and_stop():
    """Thresholds and cycle limits inside a run of repeated steps stop the loop at exactly the
    reference's component; the next call carries on."""
    G = 512
    rs, psf, dirty = dominated_problem(5, G=G, n_sources=20, amplitudes=(40.0, 25.0))
    fn, q = _clean(G, 1, 0, 0.02, 0.1, dirty, psf, {'form': 'multi', 'repeats_always': True})
    patch = (1, 47, 33)
    img, model = dirty.copy(), np.zeros_like(dirty)
    ref = orc.Clean(G, 0.02, 0.1, 0, img, psf, model)
    ref.reset()
    first = float(np.max(ref._tile_max))
    for cycles, threshold in ((1, 0.0), (3, 0.0), (5, 0.0), (70, 0.7 * first), (70, 0.7 * first),
                              (9, 0.0), (500, 0.2 * first), (100, 0.0)):
        want = []
        for _ in range(cycles):
            v, pos, pix = ref(patch, threshold)
            if v is None:
                break
            want.append((v, ref.last_pos, np.array(pix)))
        got = fn.run_cycles(patch, threshold, cycles)
        _check(fn, q, got, (want, img, model, ref._tile_max, ref._tile_pos))


@pytest.mark.parametrize('case', ['deep', 'stops after the first', 'noise decides', 'limit', 'one cycle'])
@pytest.mark.parametrize('mode,P', [(0, 1), (1, 1), (1, 3), (0, 2)])
def test_major_cycles_in_one_call(mode, P, case):
    """kimg_clean_major_cycles: the first cycle without a threshold and the others below
    max(noise threshold, (1 - major gain) x the first peak), worked out on the device -- against the
    two steps of the reference (frontend.py:560-585) on the restated CleanHost, with the threshold
    made on the host the way the frontend makes it."""
    from katsdpimager_amd import clean
    G = 512
    rs, psf, dirty = dominated_problem(90 + P + 4 * mode, G=G, P=P, n_sources=40,
                                       amplitudes=(6.0,) if case == 'stops after the first' else (3.0, 2.5))
    patch = (P, 47, 33)
    major_gain = 0.999 if case == 'deep' else 0.05 if case == 'stops after the first' else 0.85
    minor = {'limit': 40, 'one cycle': 1}.get(case, 400)
    img, model = dirty.copy(), np.zeros_like(dirty)
    ref = orc.Clean(G, 0.02, 0.1, mode, img, psf, model)
    ref.reset()
    v, pos, pix = ref(patch, 0.0)
    want = [(v, ref.last_pos, np.array(pix))]
    peak_power = clean.metric_to_power(mode, float(v))
    noise = (0.12 * peak_power if case == 'noise decides' else 1e-4 * peak_power)
    noise_threshold = noise * clean.noise_threshold_scale(mode, 5.0, P)
    threshold = max(noise_threshold, (1.0 - major_gain) * peak_power)
    if peak_power > threshold:
        metric = float(np.float32(clean.power_to_metric(mode, threshold)))
        for _ in range(minor - 1):
            v, pos, pix = ref(patch, metric)
            if v is None:
                break
            want.append((v, ref.last_pos, np.array(pix)))
    if case == 'stops after the first':
        assert len(want) == 1
    elif case == 'limit':
        assert len(want) == minor
    elif case == 'noise decides':
        assert 1 < len(want) < minor
    elif case == 'deep':
        assert len(want) > 100
    for tuning in ({'form': 'multi'}, {'form': 'auto', 'repeats_always': True}):
        fn, q = _clean(G, P, mode, 0.02, 0.1, dirty, psf, tuning)
        assert fn.run_major_cycles(patch, noise_threshold, 1.0 - major_gain, minor)
        values, positions, pixels = fn._collect_cycle_arrays()
        got = [(values[i], tuple(positions[i]), pixels[i]) for i in range(len(values))]
        _check(fn, q, got, (want, img, model, ref._tile_max, ref._tile_pos))


def test_major_cycles_thresholds_as_the_host_makes_them():
    """The threshold the device works out from the first peak is the float32 the host would pass:
    for many peaks and both metrics, stopping exactly where the host's threshold stops."""
    from katsdpimager_amd import clean
    G = 256
    rs = np.random.RandomState(7)
    g1 = np.exp(-0.5 * ((np.arange(G) - G // 2) / 2.0) ** 2)
    for trial in range(12):
        mode = trial % 2
        P = 1 + trial % 3 if mode else 1
        psf = np.repeat(np.outer(g1, g1)[None], P, axis=0).astype(np.float32)
        dirty = (0.001 * rs.standard_normal((P, G, G))).astype(np.float32)
        # a ladder of isolated sources: the loop stops between two rungs, where the threshold falls
        amps = np.sort(rs.uniform(0.2, 3.0, 30))[::-1]
        for k, amp in enumerate(amps):
            y, x = 20 + 40 * (k // 6), 20 + 40 * (k % 6)
            dirty[:, y, x] += np.float32(amp) * rs.uniform(0.5, 1.0, P).astype(np.float32)
        major_gain = float(rs.uniform(0.3, 0.95))
        img, model = dirty.copy(), np.zeros_like(dirty)
        ref = orc.Clean(G, 0.02, 1.0, mode, img, psf, model)
        ref.reset()
        patch = (P, 9, 9)
        v, pos, pix = ref(patch, 0.0)
        want = [(v, ref.last_pos, np.array(pix))]
        peak_power = clean.metric_to_power(mode, float(v))
        noise_threshold = float(rs.uniform(0.05, 0.5)) * clean.noise_threshold_scale(mode, 5.0, P)
        threshold = max(noise_threshold, (1.0 - major_gain) * peak_power)
        if peak_power > threshold:
            metric = float(np.float32(clean.power_to_metric(mode, threshold)))
            for _ in range(99):
                v, pos, pix = ref(patch, metric)
                if v is None:
                    break
                want.append((v, ref.last_pos, np.array(pix)))
        fn, q = _clean(G, P, mode, 0.02, 1.0, dirty, psf, {'form': 'multi'})
        assert fn.run_major_cycles(patch, noise_threshold, 1.0 - major_gain, 100)
        values, positions, pixels = fn._collect_cycle_arrays()
        got = [(values[i], tuple(positions[i]), pixels[i]) for i in range(len(values))]
        _check(fn, q, got, (want, img, model, ref._tile_max, ref._tile_pos))
