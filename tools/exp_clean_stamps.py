"""In-kernel time stamps of the one-launch CLEAN cycle (test build -DKIMG_CLEAN_STAMPS:
python tools/build_variant.py stamps clean.hip -DKIMG_CLEAN_STAMPS; KIMG_VARIANT_LIB=stamps).
Prints, averaged over cycles, when (us after the lattice workgroup's first instruction) each
stamped point is reached by lattice workgroup (0, 0) [L0..L5] and by the bookkeeping workgroup
[K0..K5], from the 100 MHz wall clock (10 ns resolution)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katsdpimager_amd import _lib as _kl
if os.environ.get('KIMG_VARIANT_LIB'):
    _kl.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build_variants',
                                'libkimg_%s.so' % os.environ['KIMG_VARIANT_LIB'])
from katsdpimager_amd import accel, clean, parameters

G, P = 4096, 1
ctx = accel.create_some_context()
q = ctx.create_command_queue()
rs = np.random.RandomState(4)
g1 = np.exp(-0.5 * ((np.arange(G) - G // 2) / 6.0) ** 2).astype(np.float32)
psf = np.outer(g1, g1)[None].astype(np.float32)
psf += (0.002 * rs.standard_normal(psf.shape)).astype(np.float32)
psf[:, G // 2, G // 2] = 1.0
sky = (0.01 * rs.standard_normal((P, G, G))).astype(np.float32)
for _ in range(200):
    y, x = rs.randint(100, G - 100, 2)
    sky[:, y - 30:y + 31, x - 30:x + 31] += rs.uniform(0.5, 2.0) * psf[:, G // 2 - 30:G // 2 + 31,
                                                                         G // 2 - 30:G // 2 + 31]
fixed = parameters.FixedImageParameters([0], np.float32)
ip = parameters.ImageParameters(fixed, 1.0, None, 0.2, None, pixel_size=1e-5, pixels=G)
cp = parameters.CleanParameters(1000, 0.1, 0.85, 5.0, 0, 0.01, 0.5, 0.02)
op = clean.CleanTemplate(ctx, cp, np.float32, P, {'form': 'one_launch'}).instantiate(q, ip)
op.ensure_all_bound()
op.buffer('psf').set(q, psf)
patch = (1, 111, 133)
acc = []
for n in range(10, 60):
    op.buffer('dirty').set(q, sky)
    op.buffer('model').zero(q)
    op.reset()
    op.run_cycles(patch, 0.0, 2 * n)            # even: the last launch wrote st[0]
    st = np.empty((16,), np.int32)
    op._state.get_region(q, st, np.s_[:16], np.s_[:])
    pad = st[4:16].astype(np.int64)
    if pad[0] and pad[6]:
        acc.append((pad - pad[0]) & 0xffffffff)
a = np.array(acc, np.float64)
a = np.where(a > 2 ** 31, a - 2 ** 32, a) * 0.01
print('lattice (0,0):', ' '.join('L%d %.2f' % (i, a[:, i].mean()) for i in range(6)))
print('bookkeeping  :', ' '.join('K%d %.2f' % (i, a[:, 6 + i].mean()) for i in range(6)))
