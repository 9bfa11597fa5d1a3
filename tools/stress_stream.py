"""Several channels in flight against the same channels one after the other, repeatedly: the first peak,
the PSF patch and the weights' noise of every channel must come out the same (a race between the
streams would show as a channel without data, a NaN or a wrong peak).

    python tools/stress_stream.py [rounds] [short cuts 0/1]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch                                                            # noqa: E402
from katsdpimager_amd import _lib as _kl                                # noqa: E402
if os.environ.get('KIMG_VARIANT_LIB'):
    _kl.LIB_PATH = os.path.join(ROOT, 'build_variants', 'libkimg_%s.so' % os.environ['KIMG_VARIANT_LIB'])
import synth                                                            # noqa: E402
from katsdpimager_amd import accel, frontend, imaging, parallel, parameters, preprocess, weight   # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
short_cuts = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
imaging.Imaging.one_call_major_cycles = short_cuts
imaging.Imaging.device_psf_stage = short_cuts
if os.environ.get('KIMG_STRESS_OLD_WEIGHTS'):
    # (the two read-backs of weight.py:525-531 instead of kimg_density_weights_robust)
    def _finalize(self):
        self.ensure_all_bound()
        if self._mean_weight is not None:
            mean_weight = self._mean_weight()
            self._density_weights.a = (5 * 10**(-self.robustness))**2 / mean_weight
            self._density_weights.b = 1.0
        return self._density_weights()
    weight.Weights.finalize = _finalize
if os.environ.get('KIMG_STRESS_DIAG'):
    _orig_make_weights = frontend.make_weights

    def _make_weights(reader, rel_channel, imager, weight_type, vis_block):
        out = _orig_make_weights(reader, rel_channel, imager, weight_type, vis_block)
        if not (out[0] == out[0]):
            import threading
            w = imager._weights
            qq = imager.command_queue
            grid = w.buffer('grid').get(qq)
            chunk = next(iter(reader.iter_slice_device(rel_channel, 0, vis_block)))
            print('NaN weights in thread', threading.get_ident(), 'channel store', stores.index(reader),
                  ': density sums', w._density_weights.buffer('sums').get(qq),
                  'mean sums', w._mean_weight.buffer('sums').get(qq),
                  'grid nonzero', int(np.count_nonzero(grid)), 'nan', int(np.isnan(grid).sum()),
                  'chunk n', chunk.num_vis, 'weights sum', float(chunk.weights.tensor.double().sum()),
                  'uv range', int(chunk.uv.tensor.min()), int(chunk.uv.tensor.max()), flush=True)
            # once more, now that the other threads are further on
            out2 = _orig_make_weights(reader, rel_channel, imager, weight_type, vis_block)
            print('   again:', out2, flush=True)
        return out
    frontend.make_weights = _make_weights
ctx = accel.create_some_context()
q = ctx.create_command_queue()
big = bool(os.environ.get('KIMG_STRESS_BIG'))
channels, n_in, G = (8, 1_500_000, 4096) if big else (8, 600_000, 2048)
cp = parameters.CleanParameters(200 if big else 100, 0.1, 0.85, 5.0, 0, 0.01, 0.5, 0.02)
wparm = parameters.WeightParameters(weight.WeightType.ROBUST, 0.0)
stores, params, keep = [], [], []
for c in range(channels):
    obs = synth.make_observation(G, n_in, 32 if big else 16, 1, device=ctx.device, seed=20 + c,
                                 channel_scale=parallel.channel_frequency_scale(c, channels))
    synth.add_point_sources(obs, 40 if big else 30, seed=100 + c, noise=0.02)
    ipd, gpd, apd = synth.make_parameters(obs, 1, 28, degrid=True)
    arrays = (accel.DeviceArray(ctx, (n_in, 3), np.float32, tensor=obs.uvw),
              accel.DeviceArray(ctx, (1, n_in, 1), np.float32, tensor=obs.weights[None].contiguous()),
              accel.DeviceArray(ctx, (1, n_in, 1), np.complex64, tensor=obs.raw_vis[None].contiguous()))
    torch.cuda.synchronize()
    coll = preprocess.VisibilityCollectorDevice(q, [ipd], [gpd], 1 << 20)
    coll.add(arrays[0], arrays[1], arrays[2], None, None, np.identity(1, np.complex64), None)
    coll.close()
    q.finish()
    torch.cuda.synchronize()
    stores.append(coll.reader())
    params.append((ipd, gpd, apd))
    if not os.environ.get('KIMG_STRESS_DROP'):
        keep.append((obs, arrays, coll))
    del obs, arrays, coll
block = max(r.len(0, 0) for r in stores)
imagers = {}


def make_job(channel, worker=0, own_queue=True):
    ipd, gpd, apd = params[channel]
    key = (channel, own_queue)      # (the channels' parameters differ: an imager each)
    if key not in imagers:
        queue = ctx.create_command_queue() if own_queue else q
        im = imaging.ImagingTemplate(ctx, apd, ipd.fixed, wparm, gpd.fixed, cp).instantiate(
            queue, ipd, gpd, block, 0, 2)
        im.ensure_all_bound()
        imagers[key] = im
    return dict(reader=stores[channel], rel_channel=0, imager=imagers[key], image_p=ipd, grid_p=gpd,
                clean_p=cp, weight_type=wparm.weight_type, vis_block=block, major=2, degrid=True)


def key_of(stats):
    if stats is None:
        return None
    return (stats['psf_patch'], round(float(stats['peaks'][0]), 3), round(float(stats['weights_noise']), 6),
            stats['minor'])


if os.environ.get('KIMG_STRESS_PREMAKE'):
    for c_ in range(channels):
        make_job(c_, 0)
    torch.cuda.synchronize()
first = None
if os.environ.get('KIMG_STRESS_STREAM_FIRST'):
    # (imagers, plans and tables made for the first time by four threads at once)
    first = [key_of(s_) for s_ in frontend.process_channel_stream(
        lambda ch, worker: make_job(ch, worker), range(channels), workers=4)]
want = []
for c in range(channels):
    want.append(key_of(frontend.process_channel(**make_job(c, 0, own_queue=False))))
q.finish()
print('serial:', want[:2], '...')
bad = 0
if first is not None:
    for c in range(channels):
        if first[c] is None or first[c][0] != want[c][0] or abs(first[c][1] - want[c][1]) > 2e-3 * abs(want[c][1]):
            bad += 1
            print('first stream, channel', c, 'got', first[c], 'want', want[c])
for r in range(rounds):
    out = frontend.process_channel_stream(lambda ch, worker: make_job(ch, worker), range(channels), workers=4)
    got = [key_of(s) for s in out]
    for c in range(channels):
        ok = got[c] is not None and want[c] is not None and got[c][0] == want[c][0] \
            and abs(got[c][1] - want[c][1]) <= 2e-3 * abs(want[c][1]) and abs(got[c][2] - want[c][2]) <= 1e-5 * want[c][2]
        if not ok:
            bad += 1
            print('round', r, 'channel', c, 'got', got[c], 'want', want[c])
print('short cuts', short_cuts, ':', bad, 'bad channel results in', rounds, 'rounds')
