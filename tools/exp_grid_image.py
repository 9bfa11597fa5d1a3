"""grid -> image and image -> grid at w = 0: the three routes (own two-launch transforms, the FFT
library's complex-to-real plan, complex-to-complex), timed.

    python tools/exp_grid_image.py [layer size] [grid size] [repeat] [w of the slice]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katsdpimager_amd import _lib as _kl
if os.environ.get('KIMG_VARIANT_LIB'):
    _kl.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build_variants',
                                'libkimg_%s.so' % os.environ['KIMG_VARIANT_LIB'])
from katsdpimager_amd import accel, image

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Gg = int(sys.argv[2]) if len(sys.argv) > 2 else 1244
repeat = int(sys.argv[3]) if len(sys.argv) > 3 else 50
W_SLICE = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
ctx = accel.create_some_context()
q = ctx.create_command_queue()
rs = np.random.RandomState(1)
grid = (rs.standard_normal((1, Gg, Gg)) + 1j * rs.standard_normal((1, Gg, Gg))).astype(np.complex64)
k1d = rs.uniform(1.0, 2.0, G).astype(np.float32)
model = rs.uniform(-1, 1, (1, G, G)).astype(np.float32)
lm_scale = 0.3 / G
lm_bias = -0.5 * G * lm_scale
ROUTES = {'own': {}, 'library': {'own_transform': False}, 'c2c': {'real_transform': False}}
if os.environ.get('KIMG_VARIANT_LIB'):
    ROUTES = {'own': {}, 'c2c': ROUTES['c2c']}
ref = {}
for route, tuning in ROUTES.items():
    template = image.GridImageTemplate(ctx, np.float32, tuning)
    plan = template.make_fft_plan((G, G))
    g2i = template.instantiate_grid_to_image(q, (1, Gg, Gg), lm_scale, lm_bias, plan)
    g2i.ensure_all_bound()
    i2g = template.instantiate_image_to_grid(q, (1, Gg, Gg), lm_scale, lm_bias, plan)
    i2g.bind(layer=g2i.buffer('layer'), kernel1d=g2i.buffer('kernel1d'))
    i2g.ensure_all_bound()
    g2i.buffer('kernel1d').set(q, k1d)
    g2i.buffer('grid').set(q, grid)
    i2g.buffer('image').set(q, model)
    g2i.set_w(W_SLICE)
    i2g.set_w(W_SLICE)
    g2i.buffer('image').zero(q)
    g2i()
    i2g()
    out = (g2i.buffer('image').get(q), i2g.buffer('grid').get(q))
    ref.setdefault('c2c', out) if route == 'c2c' else None
    res = {}
    for name, op in (('grid_to_image', g2i), ('image_to_grid', i2g)):
        q.finish()
        t0 = time.perf_counter()
        for _ in range(repeat):
            op()
        q.finish()
        res[name] = (time.perf_counter() - t0) / repeat * 1e6
    ref[route] = out
    print('%-8s grid_to_image %7.1f us   image_to_grid %7.1f us' % (route, res['grid_to_image'], res['image_to_grid']))
for route in [r for r in ('own', 'library') if r in ROUTES]:
    print('%-8s max deviation from c2c: image %.2e of the peak, grid %.2e of the peak' % (
        route, np.abs(ref[route][0] - ref['c2c'][0]).max() / np.abs(ref['c2c'][0]).max(),
        np.abs(ref[route][1] - ref['c2c'][1]).max() / np.abs(ref['c2c'][1]).max()))
