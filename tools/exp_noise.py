#!/usr/bin/env python3
"""Noise estimate on a noise-like image of the bench's size, timed; run it under
``rocprofv3 --kernel-trace --stats`` for the duration of each radix-select pass."""
import argparse
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katsdpimager_amd import accel, clean      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pixels', type=int, default=4096)
    ap.add_argument('--pols', type=int, default=1)
    ap.add_argument('--repeat', type=int, default=20)
    args = ap.parse_args()
    ctx = accel.create_some_context()
    q = ctx.create_command_queue()
    rs = np.random.RandomState(1)
    shape = (args.pols, args.pixels, args.pixels)
    img = rs.standard_normal(shape).astype(np.float32)
    op = clean.NoiseEstTemplate(ctx, np.float32, args.pols).instantiate(q, shape, 0.02)
    op.ensure_all_bound()
    op.buffer('dirty').set(q, img)
    got = op()
    bp = op.border_pixels
    want = np.median(np.abs(img[:, bp:-bp, bp:-bp])) * np.float32(clean._MEDIAN_TO_RMS)
    assert got == np.float32(want), (got, want)
    q.finish()
    t0 = time.perf_counter()
    for _ in range(args.repeat):
        op()
    dt = (time.perf_counter() - t0) / args.repeat
    print('noise_est %.1f us per call (%d x %d x %d)' % (dt * 1e6, *shape))


if __name__ == '__main__':
    main()
