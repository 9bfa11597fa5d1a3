"""Restoring-beam convolution of one image plane: the library's own three-launch transforms against
the FFT library's real <-> half-complex plans.    python tools/exp_convolve_beam.py [size ...]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katsdpimager_amd import accel, beam

ctx = accel.create_some_context()
q = ctx.create_command_queue()
for G in [int(a) for a in sys.argv[1:]] or [4096, 4800]:
    model = np.random.RandomState(1).standard_normal((G, G)).astype(np.float32)
    line = '%d^2:' % G
    for route, tuning in (('own', None), ('library', {'own_transform': False})):
        fn = beam.ConvolveBeamTemplate(ctx, (G, G), np.float32, tuning=tuning).instantiate(q)
        fn.beam = beam.Beam(1.0, 2.5, 1.8, 0.4)
        fn.ensure_all_bound()
        fn.buffer('image').set(q, model)
        fn()
        q.finish()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        q.finish()
        line += '  %s %.1f us' % (route, (time.perf_counter() - t0) / 20 * 1e6)
        del fn
    print(line)
