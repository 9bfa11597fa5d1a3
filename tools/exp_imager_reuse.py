"""Does an imager remember the channel it imaged before?  Channel A on a fresh imager, then channel B
and channel A again on that imager: the two results for A must agree to the last bits the float
atomics leave (first peak, weights noise, PSF patch, dirty image).

    python tools/exp_imager_reuse.py [image size]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch                                                            # noqa: E402
import synth                                                            # noqa: E402
from katsdpimager_amd import accel, frontend, imaging, parallel, parameters, preprocess, weight   # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_in = 1_500_000 if G >= 4096 else 600_000
ctx = accel.create_some_context()
q = ctx.create_command_queue()
cp = parameters.CleanParameters(200, 0.1, 0.85, 5.0, 0, 0.01, 0.5, 0.02)
wparm = parameters.WeightParameters(weight.WeightType.ROBUST, 0.0)
stores, params, keep = [], [], []
for c in (3, 0, 1, 2):
    obs = synth.make_observation(G, n_in, 32, 1, device=ctx.device, seed=20 + c,
                                 channel_scale=parallel.channel_frequency_scale(c, 8))
    synth.add_point_sources(obs, 40, seed=100 + c, noise=0.02)
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
    keep.append((obs, arrays, coll))
block = max(r.len(0, 0) for r in stores)


def fresh(which):
    ipd, gpd, apd = params[which]
    im = imaging.ImagingTemplate(ctx, apd, ipd.fixed, wparm, gpd.fixed, cp).instantiate(q, ipd, gpd, block, 0, 2)
    im.ensure_all_bound()
    return im


def run(im, which, majors=2):
    ipd, gpd, apd = params[which]
    stats = frontend.process_channel(stores[which], 0, im, ipd, gpd, cp, wparm.weight_type, block, majors, True)
    q.finish()
    return stats, im.get_buffer('dirty').copy(), im.get_buffer('psf').copy(), im.get_buffer('model').copy()


def show(label, a, b):
    sa, sb = a[0], b[0]
    print(label)
    for key in ('psf_patch', 'peaks', 'weights_noise', 'normalized_noise', 'noise', 'minor'):
        print('   %-18s %s | %s' % (key, sa[key], sb[key]))
    for name, i in (('dirty', 1), ('psf', 2), ('model', 3)):
        d = np.abs(a[i] - b[i]).max() / max(np.abs(a[i]).max(), 1e-30)
        print('   %-18s max difference %.3g of the peak' % (name, d))


im = fresh(0)
first = run(im, 0)
other = run(im, 1)
again = run(im, 0)
show('channel A: fresh imager | after another channel on the same imager', first, again)
im2 = fresh(0)
show('channel A: fresh imager | another fresh imager', first, run(im2, 0))
# an imager made for another channel's parameters, after three other channels (what a worker of the
# channel stream does with its imager)
im3 = fresh(1)
for which in (1, 2, 3):
    run(im3, which)
show('channel A: fresh imager | the fourth channel of an imager made for the first', first, run(im3, 0))
im4 = fresh(1)
show('channel A: fresh imager | a fresh imager made for another channel (parameters differ)', first, run(im4, 0))
