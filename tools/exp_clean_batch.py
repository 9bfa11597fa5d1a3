"""Aggregate CLEAN minor cycles per second with C channels per launch (kimg_clean_cycles_batch):
4096^2 images, the 111 x 133 patch of a measured PSF, 1000 cycles per channel.

    python tools/exp_clean_batch.py [C ...]        (default 1 2 4 8)

Under `rocprofv3 --kernel-trace --stats` the average duration of cycle_fused_batch_kernel next to the
wall time per cycle printed here separates the kernel from the boundary between two launches."""
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

G, P, cycles = 4096, 1, 1000
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
cp = parameters.CleanParameters(cycles, 0.1, 0.85, 5.0, 0, 0.01, 0.5, 0.02)
template = clean.CleanTemplate(ctx, cp, np.float32, P)
ops = []
OWN_QUEUES = '--queues' in sys.argv
if OWN_QUEUES:
    sys.argv.remove('--queues')
for c in range(8):
    op = template.instantiate(ctx.create_command_queue() if OWN_QUEUES and c else q, ip)
    op.ensure_all_bound()
    op.buffer('psf').set(q, psf)
    ops.append(op)
patch = (1, 111, 133)
for C in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    best = None
    for rep in range(3):
        for c in range(C):
            qc = ops[c].command_queue
            ops[c].buffer('dirty').set(qc, (sky * (1 + 0.07 * c)).astype(np.float32))
            ops[c].buffer('model').zero(qc)
            ops[c].reset()
            qc.finish()
        q.finish()
        t0 = time.perf_counter()
        if C == 1:
            runs = [ops[0].run_cycles(patch, 0.0, cycles)]
        else:
            clean.enqueue_cycles_batch(ops[:C], [patch] * C, [0.0] * C, [cycles] * C).finish()
            t1 = time.perf_counter()
            runs = [op._collect_cycles() for op in ops[:C]]
        q.finish()
        dt = time.perf_counter() - t0
        done = sum(len(r) for r in runs)
        best = dt if best is None else min(best, dt)
    extra = '' if C == 1 else '  (enqueue + device %.3f ms, read-back %.3f ms)' % ((t1 - t0) * 1e3, (dt - (t1 - t0)) * 1e3)
    print('C=%d: %d cycles in %.3f ms = %.2f us per launch, %.1f K cycles/s aggregate%s' % (
        C, done, best * 1e3, best * 1e6 / cycles, done / best / 1e3, extra), flush=True)
