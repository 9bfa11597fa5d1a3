#!/usr/bin/env python3
"""End-to-end example on synthetic data: raw visibilities -> device preprocessing -> HBM-resident
store -> imaging weights, PSF, major/minor cycles -> restored image, all on one MI355X.

    python examples/image_channel.py [--pixels 2048] [--vis 4000000] [--major 3]

It follows the reference's per-channel flow (frontend.py:31-83 preprocess_visibilities,
:465-658 process_channel) with the loaders and FITS output left out: the sky is three point
sources; the restoring beam is fitted to the PSF.
"""
import argparse
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--pixels', type=int, default=2048)
    ap.add_argument('--vis', type=int, default=4_000_000)
    ap.add_argument('--major', type=int, default=3)
    ap.add_argument('--minor', type=int, default=500)
    ap.add_argument('--vis-block', type=int, default=1 << 20)
    ap.add_argument('--w-planes', type=int, default=32,
                    help='W planes per slice (more than 64: the kernel table is read from HBM)')
    ap.add_argument('--kernel-width', type=int, default=28)
    ap.add_argument('--output', help='write the restored image to this FITS file')
    args = ap.parse_args(argv)
    import torch
    import scipy.optimize       # noqa: F401  (used by beam.fit_beam; imported here, outside the timings)
    import synth
    from katsdpimager_amd import accel, beam, frontend, imaging, parameters, preprocess, weight

    ctx = accel.create_some_context()
    queue = ctx.create_command_queue()
    # array geometry, uvw tracks (metres) and the matching imaging parameters
    obs = synth.make_observation(args.pixels, args.vis, args.w_planes, 1, device=ctx.device)
    image_p, grid_p, array_p = synth.make_parameters(obs, 1, args.kernel_width, degrid=True)
    # three point sources -> raw visibilities (the loader's job in the reference)
    sources = [((40, -25), 1.0), ((-120, 60), 0.5), ((15, 200), 0.25)]      # (l, m) in pixels, Jy
    uvw_wl = obs.uvw.to(torch.float64) / obs.wavelength
    vis = torch.zeros(obs.n_vis, dtype=torch.complex128, device=ctx.device)
    for (lp, mp), flux in sources:
        l, m = lp * obs.pixel_size, mp * obs.pixel_size
        n = math.sqrt(1 - l * l - m * m)
        phase = uvw_wl[:, 0] * l + uvw_wl[:, 1] * m + uvw_wl[:, 2] * (n - 1)
        vis += flux / n * torch.exp(-2j * math.pi * phase)
    vis = vis.to(torch.complex64)[None, :, None].contiguous()
    weights = torch.ones((1, obs.n_vis, 1), dtype=torch.float32, device=ctx.device)

    torch.cuda.synchronize()        # the inputs above were produced on torch's own stream
    t0 = time.perf_counter()
    collector = preprocess.VisibilityCollectorDevice(queue, [image_p], [grid_p], args.vis_block)
    collector.add(accel.DeviceArray(ctx, (obs.n_vis, 3), np.float32, tensor=obs.uvw),
                  accel.DeviceArray(ctx, weights.shape, np.float32, tensor=weights),
                  accel.DeviceArray(ctx, vis.shape, np.complex64, tensor=vis),
                  None, None, np.ones((1, 1), np.complex64), None)
    collector.close()
    reader = collector.reader()
    queue.finish()
    t1 = time.perf_counter()
    print('preprocessed {} visibilities to {} in {:.1f} ms ({:.1f} MB resident)'.format(
        collector.num_input, collector.num_output, (t1 - t0) * 1e3, collector.nbytes() / 1e6))

    weight_p = parameters.WeightParameters(weight.WeightType.ROBUST, 0.0)
    clean_p = parameters.CleanParameters(args.minor, 0.1, 0.85, 5.0, 0, 0.01, 0.5, 0.02)
    template = imaging.ImagingTemplate(ctx, array_p, image_p.fixed, weight_p, grid_p.fixed, clean_p)
    imager = template.instantiate(queue, image_p, grid_p, args.vis_block, 0, args.major, streams=2)
    imager.ensure_all_bound()
    stats = frontend.process_channel(reader, 0, imager, image_p, grid_p, clean_p,
                                     weight_p.weight_type, args.vis_block, args.major, True,
                                     fit_beam=True)
    queue.finish()
    t2 = time.perf_counter()
    print('imaged in {:.1f} ms: {} major / {} minor cycles, PSF patch {}, noise {:.3g}'.format(
        (t2 - t1) * 1e3, stats['major'], stats['minor'], stats['psf_patch'], stats['noise']))

    print('restoring beam: {}'.format(stats['restoring_beam']))
    beam.restore(imager, stats['restoring_beam'])
    restored = imager.get_buffer('dirty')[0]
    G = args.pixels
    for (lp, mp), flux in sources:
        y, x = G // 2 + mp, G // 2 + lp
        box = restored[y - 3:y + 4, x - 3:x + 4]
        print('source at (l, m) = ({:5d}, {:5d}) px, {:.2f} Jy: restored peak {:.3f}'.format(
            lp, mp, flux, float(box.max())))
    if args.output:
        from katsdpimager_amd import io, polarization
        image_p.fixed.polarizations = [polarization.STOKES_I]
        io.write_fits_image(imager.get_buffer('dirty'), image_p, args.output, 0,
                            (0.0, math.radians(-45.0)), beam=stats['restoring_beam'])
        print('wrote', args.output)
    return restored, stats


if __name__ == '__main__':
    main()
