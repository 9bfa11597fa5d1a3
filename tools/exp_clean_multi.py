"""Minor cycles per second of the multi-component CLEAN launch (KIMG_CLEAN_FORM_MULTI) against the
one-launch-per-cycle form, by the cap on the components per launch, on the bench's CLEAN image
(4096^2, 200 point sources (x) PSF + noise), with the launches taken and the components per launch.

    python tools/exp_clean_multi.py [patch height] [patch width] [cycles] [image size] [polarizations] [dominant]

`dominant` > 0: the first source gets that amplitude (the others: 0.5 .. 2), a field whose first
cycles all go to one peak -- what the repeated steps of a launch are for."""
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

ph = int(sys.argv[1]) if len(sys.argv) > 1 else 111
pw = int(sys.argv[2]) if len(sys.argv) > 2 else 133
cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
G = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
P = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dominant = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
ctx = accel.create_some_context()
q = ctx.create_command_queue()
rs = np.random.RandomState(4)
g1 = np.exp(-0.5 * ((np.arange(G) - G // 2) / 6.0) ** 2).astype(np.float32)
psf = np.outer(g1, g1)[None].repeat(P, axis=0).astype(np.float32)
psf += (0.002 * rs.standard_normal(psf.shape)).astype(np.float32)
psf[:, G // 2, G // 2] = 1.0
sky = (0.01 * rs.standard_normal((P, G, G))).astype(np.float32)
for i in range(200):
    y, x = rs.randint(100, G - 100, 2)
    amp = rs.uniform(0.5, 2.0)
    if i == 0 and dominant > 0:
        amp = dominant
    sky[:, y - 30:y + 31, x - 30:x + 31] += amp * psf[:, G // 2 - 30:G // 2 + 31,
                                                                     G // 2 - 30:G // 2 + 31]


class _IP:      # what CleanTemplate.instantiate reads of the image parameters
    pixels = G

    class fixed:
        polarizations = list(range(P))
        real_dtype = np.float32


cp = parameters.CleanParameters(cycles, 0.1, 0.85, 5.0, 0, 0.01, 0.5, 0.02)
patch = (P, ph, pw)
first = None
for form, comps, reps, always in [('one_launch', 0, 0, False), ('multi', 1, 1, False), ('multi', 4, 1, False),
                                  ('multi', 8, 1, False), ('multi', 8, 2, True), ('multi', 8, 4, True),
                                  ('multi', 8, 8, True), ('multi', 8, 0, False)]:
    cl = clean.CleanTemplate(ctx, cp, np.float32, P, {'form': form, 'components': comps, 'repeats': reps,
                                                      'repeats_always': always}).instantiate(q, _IP)
    cl.ensure_all_bound()
    cl.buffer('psf').set(q, psf)
    rates = []
    for rep in range(4):
        cl.buffer('dirty').set(q, sky)
        cl.buffer('model').zero(q)
        cl.reset()
        q.finish()
        t0 = time.perf_counter()
        cl.run_cycles(patch, 0.0, cycles, collect=False)
        arrays = cl._collect_cycle_arrays()         # (read-back included)
        q.finish()
        rates.append(len(arrays[0]) / (time.perf_counter() - t0))
        got = list(zip(arrays[0], [tuple(p) for p in arrays[1].tolist()], arrays[2]))
    launches = cl.last_launches()
    sig = [(round(float(v), 6), tuple(p)) for v, p, m in got]
    first = first or sig
    print('%-10s cap %d x %d%s: %9.0f cycles/s = %.2f us per cycle; launches %s (%.2f components each, %.2f us per launch)  %s' % (
        form, comps, reps, ' always' if always else ' (auto)' if form == 'multi' and reps != 1 else '', max(rates), 1e6 / max(rates), launches,
        len(got) / launches if launches else 1.0,
        1e6 * len(got) / max(rates) / (launches or len(got)),
        'same components' if sig == first else 'DIFFERENT'))
    del cl
