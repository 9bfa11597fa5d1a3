"""Soak test of the multi-component CLEAN launch (not part of the suite: run on the GPU box when the
kernel changes): the fuzz problems of tests/test_clean_multi_model.py for many seeds and caps on the
components per launch, plus many-source images of random shape, each against the restated CleanHost,
bit for bit (components, residual image, model, tile arrays).

    python tools/soak_clean_multi.py [first seed] [seeds]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_clean_multi_model import fuzz_problem, reference_run, sources_problem      # noqa: E402
from katsdpimager_amd import accel, clean, parameters                                  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = accel.create_some_context()
q = ctx.create_command_queue()


def run(G, P, mode, border, loop_gain, dirty, psf, patch, threshold, cycles, cap, repeats=0, always=False):
    fixed = parameters.FixedImageParameters(list(range(P)), np.float32)
    ip = parameters.ImageParameters(fixed, 1.0, None, 0.2, None, pixel_size=1e-5, pixels=G)
    cp = parameters.CleanParameters(1000, loop_gain, 0.85, 5.0, mode, 0.01, 0.5, border)
    fn = clean.CleanTemplate(ctx, cp, np.float32, P, {'form': 'multi', 'components': cap, 'repeats': repeats, 'repeats_always': always}).instantiate(q, ip)
    fn.ensure_all_bound()
    fn.buffer('dirty').set(q, dirty)
    fn.buffer('psf').set(q, psf)
    fn.buffer('model').zero(q)
    fn.reset()
    want = reference_run(G, border, loop_gain, mode, dirty, psf, patch, threshold, cycles)
    got = fn.run_cycles(patch, threshold, cycles)
    log, img, model, tile_max, tile_pos = want
    assert len(got) == len(log), (len(got), len(log))
    for a, b in zip(got, log):
        assert a[0] == b[0] and tuple(a[1]) == tuple(b[1]), (a, b)
        np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(fn.buffer('dirty').get(q), img)
    np.testing.assert_array_equal(fn.buffer('model').get(q), model)
    np.testing.assert_array_equal(fn.buffer('tile_max').get(q), tile_max)
    np.testing.assert_array_equal(fn.buffer('tile_pos').get(q), tile_pos)
    return len(got), fn.last_launches()


total = launches = 0
for seed in range(first, first + count):
    rs, P, mode, G, border, loop_gain, psf, dirty, patch, cycles = fuzz_problem(seed)
    cap = int(rs.randint(0, 9))
    first_value = float(np.max(np.abs(dirty))) if mode == 0 else float(np.max(np.sum(dirty * dirty, axis=0)))
    threshold = float(rs.choice([0.0, 0.3 * first_value, 2.0 * first_value]))
    n, l = run(G, P, mode, border, loop_gain, dirty, psf, patch, threshold, cycles, cap, int(rs.randint(0, 9)), bool(seed % 3))
    total += n
    launches += l or 0
    # many sources, random geometry
    rs2 = np.random.RandomState(90000 + seed)
    G2 = int(rs2.choice([384, 512, 640, 800]))
    P2 = int(rs2.choice([1, 1, 2, 4]))
    mode2 = int(rs2.randint(0, 2))
    _, psf2, dirty2 = sources_problem(seed, G=G2, P=P2, n_sources=int(rs2.randint(5, 80)),
                                      sigma=float(rs2.uniform(1.5, 5.0)))
    patch2 = (P2, int(rs2.choice([15, 33, 65, 97, 161])), int(rs2.choice([15, 47, 65, 133, 191])))
    if seed % 2:
        # a few sources far above the rest: what repeated steps at one peak are for
        for _ in range(int(rs2.randint(1, 4))):
            y, x = rs2.randint(40, G2 - 40, 2)
            h = 12
            dirty2[:, y - h:y + h + 1, x - h:x + h + 1] += (rs2.uniform(5.0, 30.0) * psf2[
                :, G2 // 2 - h:G2 // 2 + h + 1, G2 // 2 - h:G2 // 2 + h + 1]).astype(np.float32)
    n, l = run(G2, P2, mode2, float(rs2.choice([0.0, 0.02, 0.07])), float(rs2.choice([0.05, 0.1, 0.3])),
               dirty2, psf2, patch2, 0.0, int(rs2.choice([50, 200, 400])), int(rs2.randint(0, 9)),
               int(rs2.choice([0, 0, 1, 2, 3, 8])), bool(seed % 4 < 2))
    total += n
    launches += l or 0
print('soak ok: seeds %d..%d, %d components in %d launches' % (first, first + count - 1, total, launches))
