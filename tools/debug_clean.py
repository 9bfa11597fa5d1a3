import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import golden_inputs as gi
from oracle import kimg_oracle as orc
from test_hip_parity import _clean_op
for batched in (False, True):
    c = gi.CLEAN_CONFIGS['i']
    ci = gi.clean_inputs(c)
    fn, q = _clean_op(c, ci)
    fn.reset()
    dirty = ci['dirty'].copy(); model = np.zeros_like(dirty)
    ref = orc.Clean(c['pixels'], c['border'], c['loop_gain'], c['mode'], dirty, ci['psf'], model)
    ref.reset()
    print('tiles equal', np.array_equal(fn.buffer('tile_max').get(q), ref._tile_max))
    if batched:
        out = fn.run_cycles(ci['psf_patch'], 0.0, 30)
    for i in range(30):
        v, p, m = ref(ci['psf_patch'], 0.0)
        if batched:
            v2, p2, m2 = out[i]
        else:
            v2, p2, m2 = fn(ci['psf_patch'], 0.0)
            d = fn.buffer('dirty').get(q)
            neq = np.argwhere(d != dirty)
            tm = fn.buffer('tile_max').get(q)
            tneq = np.argwhere(tm != ref._tile_max)
            if len(neq) or len(tneq):
                print('cycle', i, 'dirty mismatches', len(neq), neq[:5], 'tile mismatches', len(tneq), tneq[:5])
                if len(neq):
                    idx = tuple(neq[0]); print(d[idx], dirty[idx], d[idx]-dirty[idx])
                break
        if p != p2 or v != v2:
            print('batched' if batched else 'single', 'first mismatch at', i, (v, p, m), (v2, p2, m2))
            break
    else:
        print('batched' if batched else 'single', 'all 30 equal')
