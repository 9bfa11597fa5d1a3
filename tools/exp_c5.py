"""Config 5 as bench.py runs it (the stage loop, then the channel from the resident store in arrival
order and in store order: `major_cycle_loop`), alone -- for a kernel trace whose last ~8 ms are the
store-order channel (tools/c5_timeline.py).    python tools/exp_c5.py"""
import os
import sys
import json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench                                    # noqa: E402
import synth                                    # noqa: E402
from katsdpimager_amd import accel              # noqa: E402

sys.argv = [sys.argv[0]]
args = bench.parse_args()
ctx = accel.create_some_context()
q = ctx.create_command_queue()
obs = synth.make_observation(args.pixels, args.vis, args.w_planes, args.polarizations, device=ctx.device,
                             cover=0.30, channel_scale=bench.channel_scale(bench.rank_channel(0, 1)), seed=2)
out = bench.major_cycle_loop(args, ctx, q, obs)
sd = out.pop('store_driven')
print(json.dumps({k: sd[k] for k in sd if k.startswith('store_order')}))
