"""The executable specification of the multi-component CLEAN launch (oracle/clean_multi_model.py)
against the restated CleanHost (oracle/kimg_oracle.Clean, clean.py:1060-1075): the launch-level
algorithm -- plan, evaluate speculatively, verify, commit the verified prefix -- gives the
reference's components, images and tile arrays BIT FOR BIT.  CPU only; the HIP kernel that
implements the same algorithm is checked on the GPU by tests/test_clean_multi_gpu.py."""
import numpy as np
import pytest

from oracle import kimg_oracle as orc
from oracle.clean_multi_model import MultiClean


def fuzz_problem(seed):
    """The recipe of test_hip_parity.test_clean_fuzz."""
    rs = np.random.RandomState(5000 + seed)
    P = int(rs.randint(1, 5))
    mode = int(rs.randint(0, 2))
    G = int(rs.choice([96, 144, 200, 256]))
    border = float(rs.choice([0.0, 0.02, 0.1, 0.2]))
    loop_gain = float(rs.choice([0.05, 0.1, 0.5]))
    g1 = np.exp(-0.5 * ((np.arange(G) - G // 2) / rs.uniform(1.5, 8.0)) ** 2)
    psf = np.empty((P, G, G), np.float32)
    for p in range(P):
        psf[p] = np.outer(g1, g1) + 0.01 * rs.standard_normal((G, G))
    psf /= psf[:, G // 2, G // 2][:, None, None]
    dirty = (0.3 * rs.standard_normal((P, G, G))).astype(np.float32)
    for _ in range(10):
        y, x = rs.randint(0, G, 2)
        dirty[:, y, x] += rs.uniform(2.0, 10.0, P).astype(np.float32) * rs.choice([-1, 1])
    ph = int(rs.choice([1, 5, 33, 63, G - 1 if (G - 1) % 2 else G - 2, G + 1 if (G + 1) % 2 else G + 2]))
    pw = int(rs.choice([1, 7, 31, 65, 127]))
    patch = (P, min(ph, G), min(pw, G))
    cycles = int(rs.choice([1, 40, 150]))
    return rs, P, mode, G, border, loop_gain, psf, dirty, patch, cycles


def sources_problem(seed, G=512, P=1, n_sources=60, sigma=3.0):
    """Many point sources of similar brightness, far apart: what the planner is made for."""
    rs = np.random.RandomState(7000 + seed)
    g1 = np.exp(-0.5 * ((np.arange(G) - G // 2) / sigma) ** 2)
    psf = np.empty((P, G, G), np.float32)
    for p in range(P):
        psf[p] = np.outer(g1, g1) + 0.003 * rs.standard_normal((G, G))
    psf /= psf[:, G // 2, G // 2][:, None, None]
    dirty = (0.01 * rs.standard_normal((P, G, G))).astype(np.float32)
    h = 12
    for _ in range(n_sources):
        y, x = rs.randint(h, G - h, 2)
        amp = rs.uniform(0.5, 2.0, P) * rs.choice([-1, 1])
        for p in range(P):
            dirty[p, y - h:y + h + 1, x - h:x + h + 1] += (
                amp[p] * psf[p, G // 2 - h:G // 2 + h + 1, G // 2 - h:G // 2 + h + 1]).astype(np.float32)
    return rs, psf.astype(np.float32), dirty.astype(np.float32)


def reference_run(G, border, loop_gain, mode, dirty, psf, patch, threshold, cycles):
    img, model = dirty.copy(), np.zeros_like(dirty)
    ref = orc.Clean(G, border, loop_gain, mode, img, psf, model)
    ref.reset()
    log = []
    for _ in range(cycles):
        v, pos, pix = ref(patch, threshold)
        if v is None:
            break
        log.append((v, ref.last_pos, np.array(pix)))
    return log, img, model, ref._tile_max, ref._tile_pos


def check_model(G, border, loop_gain, mode, dirty, psf, patch, threshold, cycles, **kwargs):
    want, ref_img, ref_model, tile_max, tile_pos = reference_run(
        G, border, loop_gain, mode, dirty, psf, patch, threshold, cycles)
    img, model = dirty.copy(), np.zeros_like(dirty)
    mc = MultiClean(G, border, loop_gain, mode, img, psf, model, patch, threshold, cycles, **kwargs)
    got = mc.run()
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a[0] == b[0] and tuple(a[1]) == tuple(b[1])
        np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(img, ref_img)
    np.testing.assert_array_equal(model, ref_model)
    np.testing.assert_array_equal(mc.tile_max.reshape(tile_max.shape), tile_max)
    np.testing.assert_array_equal(mc.tile_pos.reshape(tile_pos.shape), tile_pos)
    return mc


@pytest.mark.parametrize('seed', range(12))
def test_model_fuzz(seed):
    rs, P, mode, G, border, loop_gain, psf, dirty, patch, cycles = fuzz_problem(seed)
    first = float(np.max(np.abs(dirty))) if mode == 0 else float(np.max(np.sum(dirty * dirty, axis=0)))
    threshold = float(rs.choice([0.0, 0.3 * first, 2.0 * first]))
    check_model(G, border, loop_gain, mode, dirty, psf, patch, threshold, cycles)
    # a keeper's list cut short at random (a shorter list with a higher floor is still exact)
    check_model(G, border, loop_gain, mode, dirty, psf, patch, threshold, cycles,
                rng=np.random.RandomState(seed))


@pytest.mark.parametrize('components', [1, 3, 8])
def test_model_plans_several_components(components):
    rs, psf, dirty = sources_problem(components, G=384, n_sources=40)
    mc = check_model(384, 0.02, 0.1, 0, dirty, psf, (1, 33, 47), 0.0, 120, max_components=components)
    if components > 1:
        assert mc.launches < 120 / 1.8


def test_model_edge_images():
    G = 96
    psf = np.zeros((1, G, G), np.float32)
    psf[0, G // 2 - 2:G // 2 + 3, G // 2 - 2:G // 2 + 3] = 0.5
    psf[0, G // 2, G // 2] = 1.0
    for dirty in (np.zeros((1, G, G), np.float32),              # the (x0, y0) quirk of clean.py:950
                  np.full((1, G, G), 0.75, np.float32)):        # every tile ties
        check_model(G, 0.05, 0.3, 0, dirty, psf, (1, 5, 5), 0.0, 25)


def dominated_problem(seed, G=384, P=1, n_sources=10, amplitudes=(8.0,), shaped=True):
    """A few sources far above the rest: PSF-shaped (the repeated steps of a launch hold) or single
    pixels (they do not: the PSF's skirt digs holes next to the peak that soon beat it)."""
    rs, psf, dirty = sources_problem(seed, G=G, P=P, n_sources=n_sources)
    h = 12
    for amp in amplitudes:
        y, x = rs.randint(40, G - 40, 2)
        if shaped:
            dirty[:, y - h:y + h + 1, x - h:x + h + 1] += (
                amp * psf[:, G // 2 - h:G // 2 + h + 1, G // 2 - h:G // 2 + h + 1]).astype(np.float32)
        else:
            dirty[:, y, x] += np.float32(amp)
    return rs, psf, dirty


@pytest.mark.parametrize('max_steps', [1, 2, 4, 8])
@pytest.mark.parametrize('P,mode', [(1, 0), (3, 1)])
def test_model_repeated_steps(max_steps, P, mode):
    """Several subtractions at one peak within a launch: the merge of the planned lattices' own
    sequences, cut where nothing is proven any more."""
    rs, psf, dirty = dominated_problem(11, P=P, amplitudes=(30.0, 12.0))
    mc = check_model(384, 0.02, 0.1, mode, dirty, psf, (P, 33, 47), 0.0, 120, max_steps=max_steps)
    if max_steps == 1:
        assert mc.repairs == 0
    else:
        assert mc.steps_planned >= 120      # (all of them committed, in far fewer launches)
    if max_steps == 8 and mode == 0:
        one = check_model(384, 0.02, 0.1, mode, dirty, psf, (P, 33, 47), 0.0, 120, max_steps=1)
        assert mc.launches < 0.75 * one.launches, (mc.launches, one.launches)


@pytest.mark.parametrize('seed', range(6))
def test_model_repeated_steps_that_fail(seed):
    """Peaks that do not stay the largest pixel of their lattice (single bright pixels under a PSF
    with a skirt): blocks report the step after which the peak was beaten, the steps up to it are
    committed, the lattice is evaluated again without steps -- and the plans take single steps for
    a while, so that what such a field costs stays bounded."""
    rs, psf, dirty = dominated_problem(20 + seed, n_sources=[3, 10, 30][seed % 3],
                                       amplitudes=(5.0, 9.0)[:1 + seed % 2], shaped=False)
    mc = check_model(384, 0.02, [0.1, 0.3][seed % 2], 0, dirty, psf, (1, 33, 47), 0.0, 150, max_steps=8)
    one = check_model(384, 0.02, [0.1, 0.3][seed % 2], 0, dirty, psf, (1, 33, 47), 0.0, 150, max_steps=1)
    assert mc.repairs > 0
    assert mc.launches <= 1.15 * one.launches + 4, (mc.launches, one.launches)


@pytest.mark.parametrize('seed', range(20, 32))
def test_model_fuzz_with_repeated_steps(seed):
    rs, P, mode, G, border, loop_gain, psf, dirty, patch, cycles = fuzz_problem(seed)
    first = float(np.max(np.abs(dirty))) if mode == 0 else float(np.max(np.sum(dirty * dirty, axis=0)))
    threshold = float(rs.choice([0.0, 0.3 * first, 2.0 * first]))
    for max_steps in (2, 8):
        check_model(G, border, loop_gain, mode, dirty, psf, patch, threshold, cycles, max_steps=max_steps,
                    max_components=int(rs.randint(1, 9)))
        check_model(G, border, loop_gain, mode, dirty, psf, patch, threshold, cycles, max_steps=max_steps,
                    refine=False, rng=np.random.RandomState(seed))
