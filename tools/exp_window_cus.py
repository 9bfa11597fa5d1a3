"""The `--extras` block of bench.py (four channels with 1-4 in flight, the 12-channel stream) with the
window kernels of the channels in flight on another share of the CUs than frontend.WINDOW_CUS_SHARED.

    python tools/exp_window_cus.py CUS"""
import os
import sys
import json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench                                    # noqa: E402
import synth                                    # noqa: E402
from katsdpimager_amd import accel, frontend    # noqa: E402

frontend.WINDOW_CUS_SHARED = int(sys.argv[1])
sys.argv = [sys.argv[0], '--extras']
args = bench.parse_args()
ctx = accel.create_some_context()
q = ctx.create_command_queue()
obs = synth.make_observation(args.pixels, args.vis, args.w_planes, args.polarizations, device=ctx.device,
                             cover=0.30, channel_scale=bench.channel_scale(bench.rank_channel(0, 1)), seed=2)
out = bench.major_cycle_loop(args, ctx, q, obs, extras=True)
e = out['extras']
print(frontend.WINDOW_CUS_SHARED, json.dumps({k: v for k, v in e.items() if isinstance(v, (int, float)) and 'turns' not in k}))
