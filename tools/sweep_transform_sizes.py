"""Every even layer size up to a limit that the library's own transforms take (no prime factor above
7), with a random grid size: grid -> image and image -> grid at w = 0 and at one w != 0 against the
complex-to-complex route on the FFT library.  Prints the worst deviation (of the peak).

    python tools/sweep_transform_sizes.py [largest size]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katsdpimager_amd import accel, image
from katsdpimager_amd._lib import lib

limit = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = accel.create_some_context()
q = ctx.create_command_queue()
rs = np.random.RandomState(5)
worst = (0.0, None)
count = 0
for G in range(16, limit + 1, 2):
    if not lib().kimg_grid_image_real_supported(G, 2):
        continue
    Gg = 2 * int(rs.randint(1, G // 2 + 1))
    grid = (rs.standard_normal((1, Gg, Gg)) + 1j * rs.standard_normal((1, Gg, Gg))).astype(np.complex64)
    k1d = rs.uniform(1.0, 2.0, G).astype(np.float32)
    model = rs.uniform(-1, 1, (1, G, G)).astype(np.float32)
    lm_scale = 0.3 / G
    lm_bias = -0.5 * G * lm_scale
    res = {}
    for route, tuning in (('own', {}), ('c2c', {'real_transform': False})):
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
        out = []
        for w in (0.0, 17.25):
            g2i.set_w(w)
            i2g.set_w(w)
            g2i.buffer('image').zero(q)
            g2i()
            i2g()
            out += [g2i.buffer('image').get(q), i2g.buffer('grid').get(q)]
        res[route] = out
        del g2i, i2g
    for a, b in zip(res['own'], res['c2c']):
        dev = float(np.abs(a - b).max() / np.abs(b).max())
        if dev > worst[0]:
            worst = (dev, (G, Gg))
    count += 1
print('%d sizes up to %d: worst deviation %.2e of the peak at layer / grid %s' % (count, limit, *worst))
assert worst[0] < 3e-6
