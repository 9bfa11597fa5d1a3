import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import context_queue
from katsdpimager_amd import clean, parameters
ctx, q = context_queue()
G, P, mode, border, patch = 4096, 1, 0, 0.02, (111, 133)
args = sys.argv[1:]
if '--patch' in args:
    i = args.index('--patch')
    patch = (int(args[i + 1]), int(args[i + 2]))
    del args[i:i + 3]
rs = np.random.RandomState(G + P)
g1 = np.exp(-0.5 * ((np.arange(G) - G // 2) / (4.0 if patch[0] < 300 else 60.0)) ** 2).astype(np.float32)
psf = np.repeat((np.outer(g1, g1))[None].astype(np.float32), P, axis=0)
dirty = (0.05 * rs.standard_normal((P, G, G))).astype(np.float32)
for _ in range(200):
    y, x = rs.randint(100, G - 100, 2)
    dirty[:, y, x] += rs.uniform(1.0, 5.0, P).astype(np.float32)
fixed = parameters.FixedImageParameters(list(range(P)), np.float32)
ip = parameters.ImageParameters(fixed, 1.0, None, 0.2, None, pixel_size=1e-5, pixels=G)
cp = parameters.CleanParameters(1000, 0.1, 0.85, 5.0, mode, 0.01, 0.5, border)
for form in (args or ('one_workgroup', 'one_launch')):
    fn = clean.CleanTemplate(ctx, cp, np.float32, P, {'form': form}).instantiate(q, ip)
    fn.ensure_all_bound()
    fn.buffer('psf').set(q, psf)
    for rep in range(3):
        fn.buffer('dirty').set(q, dirty); fn.buffer('model').zero(q); fn.reset(); q.finish()
        t0 = time.perf_counter(); n = len(fn.run_cycles((P,) + patch, 0.0, 1000)); q.finish()
        dt = time.perf_counter() - t0
    st = np.zeros(16, np.int32); fn._state.get_region(q, st, np.s_[:16], np.s_[:])
    print(form, n, 'cycles', round(dt / n * 1e6, 2), 'us/cycle', 'stamps (100 MHz ticks per cycle):', st[4:12].tolist())
