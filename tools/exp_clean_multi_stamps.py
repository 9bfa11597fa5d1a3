"""In-kernel time stamps of the multi-component CLEAN launch (test build:
python tools/build_variant.py mcstamps clean_multi.hip -DKIMG_MC_STAMPS; KIMG_VARIANT_LIB=mcstamps).
Average shader-clock cycles (and us at the clock measured against the wall clock of the run) after a
workgroup's first instruction at which each stamped point is reached, for the keeper [K] and for the
workgroup of block (0, 0) of the first planned lattice [L].

    python tools/exp_clean_multi_stamps.py [components per launch] [cycles]

Environment: KIMG_REPEATS (cap on the steps per lattice; 1: the single-step kernel), KIMG_REPEATS_ALWAYS (the
repeated-steps kernel from the first launch on), KIMG_DOMINANT (amplitude of the first source)."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katsdpimager_amd import _lib as _kl
if os.environ.get('KIMG_VARIANT_LIB'):
    _kl.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build_variants',
                                'libkimg_%s.so' % os.environ['KIMG_VARIANT_LIB'])
from katsdpimager_amd import accel, clean, parameters

comps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
G, P = 4096, 1
ctx = accel.create_some_context()
q = ctx.create_command_queue()
rs = np.random.RandomState(4)
g1 = np.exp(-0.5 * ((np.arange(G) - G // 2) / 6.0) ** 2).astype(np.float32)
psf = np.outer(g1, g1)[None].astype(np.float32)
psf += (0.002 * rs.standard_normal(psf.shape)).astype(np.float32)
psf[:, G // 2, G // 2] = 1.0
sky = (0.01 * rs.standard_normal((P, G, G))).astype(np.float32)
dominant = float(os.environ.get('KIMG_DOMINANT', '0'))      # the first source at this amplitude (else as the others)
for i in range(200):
    y, x = rs.randint(100, G - 100, 2)
    sky[:, y - 30:y + 31, x - 30:x + 31] += (dominant if i == 0 and dominant else rs.uniform(0.5, 2.0)) * psf[:, G // 2 - 30:G // 2 + 31,
                                                                     G // 2 - 30:G // 2 + 31]
fixed = parameters.FixedImageParameters([0], np.float32)
ip = parameters.ImageParameters(fixed, 1.0, None, 0.2, None, pixel_size=1e-5, pixels=G)
cp = parameters.CleanParameters(cycles, 0.1, 0.85, 5.0, 0, 0.01, 0.5, 0.02)
op = clean.CleanTemplate(ctx, cp, np.float32, P, {'form': 'multi', 'components': comps, 'repeats': int(os.environ.get('KIMG_REPEATS', '0')), 'repeats_always': bool(os.environ.get('KIMG_REPEATS_ALWAYS'))}).instantiate(q, ip)
op.ensure_all_bound()
op.buffer('psf').set(q, psf)
patch = (1, 111, 133)
for rep in range(3):
    op.buffer('dirty').set(q, sky)
    op.buffer('model').zero(q)
    op.reset()
    q.finish()
    t0 = time.perf_counter()
    got = op.run_cycles(patch, 0.0, cycles)
    q.finish()
    dt = time.perf_counter() - t0
tiles = op.buffer('tile_max').shape
offset = (64 + 2 * 6464 + 2 * 256 * 32 + tiles[0] * tiles[1] * 16) // 4
raw = op._state.get(q)[offset:offset + 192].view(np.int64).reshape(3, 32)
launches = op.last_launches()
print('%d components in %d launches, %.1f us per launch, %.0f cycles/s' % (
    len(got), launches, dt / launches * 1e6, len(got) / dt))
names = {0: 'start', 1: 'records + first exchange', 2: 'verified', 3: 'second exchange', 4: 'planned',
         5: 'B: component extracted', 6: 'K: state written / B: pixels loaded', 7: 'B: record stored',
         12: 'L: old list merged', 13: 'L: tiles filtered (scan)', 8: 'L: first scan done', 9: 'L: list complete', 10: 'L: sorted + stored', 11: 'L: end'}
for row, label in ((0, 'K (keeper)'), (1, 'B (block (0, 0) of the first planned lattice)'), (2, 'L (lister)')):
    n = raw[row][16]
    if n == 0:
        continue
    print('%s: %d workgroups; planned %.2f, committed %.2f per launch; scans per launch %.2f (in %.2f of the launches), listed %.1f' % (
        label, n, raw[row][17] / n, raw[row][18] / n, raw[row][15] / n, raw[row][19] / n, raw[row][14] / max(n, 1)))
    if row == 2:
        print('   old list %.1f entries, merged %.1f of which %.1f from delta records; floor %.3f on average; %.1f lattices folded' % (
            raw[row][7] / n, raw[row][6] / n, raw[row][5] / n, raw[row][17] / n / 1000, raw[row][18] / n))
        raw[row][5] = raw[row][6] = raw[row][7] = 0
    plan_names = {8: 'B: pool sorted', 9: 'B: masks', 10: 'B: bound', 11: 'B: walked'}
    for i in ((0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7) if row == 1 else (0, 1, 2, 3, 4, 5, 6, 7, 12, 13, 8, 9, 10, 11)):
        if raw[row][i]:
            print('   %-40s %8.0f cycles' % ((plan_names if row == 1 and i in plan_names else names)[i], raw[row][i] / n))
