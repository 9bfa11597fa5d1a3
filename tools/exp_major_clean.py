"""The CLEAN share of the major-cycle loop of bench.py (config 5) by the tuning of the multi-component
form: the loop's own choice of kernel, single steps only, repeated steps from the first launch on.

    python tools/exp_major_clean.py"""
import os
import sys
import json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench                                    # noqa: E402
import synth                                    # noqa: E402
from katsdpimager_amd import accel, imaging     # noqa: E402

sys.argv = [sys.argv[0]]
args = bench.parse_args()
ctx = accel.create_some_context()
q = ctx.create_command_queue()
obs = synth.make_observation(args.pixels, args.vis, args.w_planes, args.polarizations, device=ctx.device,
                             cover=0.30, channel_scale=bench.channel_scale(bench.rank_channel(0, 1)), seed=2)
orig = imaging.ImagingTemplate.__init__
for name, clean_tuning in (('auto', None), ('single steps', {'form': 'multi', 'repeats': 1}),
                           ('repeated steps always', {'form': 'multi', 'repeats_always': True}),
                           ('auto', None)):
    def init(self, context, array_parameters, fixed_image_parameters, weight_parameters,
             fixed_grid_parameters, clean_parameters, tuning=None, _t=clean_tuning):
        tuning = dict(tuning or {})
        if _t is not None:
            tuning['clean'] = _t
        orig(self, context, array_parameters, fixed_image_parameters, weight_parameters,
             fixed_grid_parameters, clean_parameters, tuning)
    imaging.ImagingTemplate.__init__ = init
    best = None
    for rep in range(3):
        out = bench.major_cycle_loop(args, ctx, q, obs, add_sources=(name == 'auto' and best is None and rep == 0
                                                                      and not getattr(obs, '_sourced', False)))
        obs._sourced = True
        out.pop('store_driven', None)
        if best is None or out['clean_ms'] < best['clean_ms']:
            best = out
    print('%-24s clean %.3f ms, %d cycles/s, total %.3f ms' % (name, best['clean_ms'], best['clean_cycles_per_s'],
                                                               best['total_ms']))
