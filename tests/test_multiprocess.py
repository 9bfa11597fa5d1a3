"""world_size-2 tests of the channel-sharding layer on the gloo backend (CPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world_size, port, result_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    try:
        from katsdpimager_amd import parallel
        import synth
        assert parallel.world() == (rank, world_size)
        # 1. channel assignment partitions the channels
        mine = parallel.assign_channels(5, world_size, rank)
        # 2. shared tables: only rank 0 computes them, everybody ends up with the same bytes
        ref = torch.from_numpy(synth.baselines_equatorial())
        tables = {'baselines': ref.clone() if rank == 0 else torch.zeros_like(ref),
                  'taper': torch.arange(16, dtype=torch.float32) if rank == 0
                  else torch.empty(16, dtype=torch.float32)}
        parallel.broadcast_shared(tables, src=0)
        assert torch.equal(tables['baselines'], ref)
        assert torch.equal(tables['taper'], torch.arange(16, dtype=torch.float32))
        # 3. every rank builds its own channel from the shared table: different channels,
        #    deterministic per channel
        obs = synth.make_observation(256, 4096, 8, device='cpu', seed=2 + mine[0],
                                     channel_scale=parallel.channel_frequency_scale(mine[0], 5))
        checksum = float(obs.uv.to(torch.float64).abs().sum())
        # 3b. a rank with several channels hands exactly its own to the per-GPU channel pool
        seen = []

        def runner(make_job, channels, workers):
            seen.append((tuple(channels), workers))
            return [make_job(c)['channel'] * 10 for c in channels]
        got = parallel.image_assigned_channels(lambda c: dict(channel=c), 5, workers=3, runner=runner)
        assert got == {c: c * 10 for c in mine} and seen == [(tuple(mine), 3)]
        # 3c. every rank reports its own device; a collision is an error on every rank
        assert parallel.check_rank_devices(rank) == list(range(world_size))
        with pytest.raises(RuntimeError):
            parallel.check_rank_devices(0)
        # 4. timing reduction and statistics gather
        slowest = parallel.max_over_ranks(1.0 + rank)
        stats = parallel.gather_stats([float(rank), checksum, float(len(mine))])
        np.save(os.path.join(result_dir, 'rank%d.npy' % rank),
                np.array([slowest, checksum, len(mine)] + stats.flatten().tolist()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_channel_sharding_world2(tmp_path):
    world_size = 2
    mp.spawn(_worker, args=(world_size, _free_port(), str(tmp_path)), nprocs=world_size, join=True)
    r = [np.load(os.path.join(str(tmp_path), 'rank%d.npy' % i)) for i in range(world_size)]
    assert r[0][0] == r[1][0] == 2.0                        # max over ranks
    assert r[0][1] != r[1][1]                               # different channels
    assert r[0][2] + r[1][2] == 5                           # all channels covered once
    np.testing.assert_array_equal(r[0][3:], r[1][3:])       # identical gathered stats
    stats = r[0][3:].reshape(world_size, 3)
    np.testing.assert_array_equal(stats[:, 0], [0.0, 1.0])
    assert stats[0, 1] == r[0][1] and stats[1, 1] == r[1][1]


def test_sharding_helpers_single_process():
    sys.path.insert(0, ROOT)
    from katsdpimager_amd import parallel
    assert parallel.world() == (0, 1)
    assert parallel.assign_channels(8, 8, 3) == [3]
    assert parallel.assign_channels(3, 8, 5) == []
    covered = sorted(c for r in range(4) for c in parallel.assign_channels(10, 4, r))
    assert covered == list(range(10))
    with pytest.raises(ValueError):
        parallel.assign_channels(4, 2, 2)
    assert parallel.channel_frequency_scale(0, 1) == 1.0
    assert parallel.channel_frequency_scale(0, 8) == pytest.approx(0.97)
    assert parallel.channel_frequency_scale(7, 8) == pytest.approx(1.03)
    assert parallel.max_over_ranks(3.5) == 3.5
    assert parallel.gather_stats([1.0, 2.0]).tolist() == [[1.0, 2.0]]
    t = {'x': torch.ones(3)}
    assert parallel.broadcast_shared(t) is t


def test_synthetic_observation_matches_oracle_quantisation():
    """tools/synth.py quantises with torch; the oracle restates preprocess.cpp with numpy."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import synth
    from oracle import kimg_oracle as orc
    obs = synth.make_observation(512, 20000, 16, device='cpu')
    # rebuild float uvw in cells the way synth does, then quantise with the oracle
    bl = torch.from_numpy(synth.baselines_equatorial())
    assert obs.uv.shape == (20000, 4) and obs.uv.dtype == torch.int16
    uv = obs.uv.numpy()
    assert uv[:, 2:].min() >= 0 and uv[:, 2:].max() < 8
    assert obs.w_plane.min() >= 8 and obs.w_plane.max() < 16      # w >= 0 after the flip, 1 slice
    assert np.abs(uv[:, :2]).max() <= int(0.30 * 512) + 1
    # the oracle's splitter agrees with torch's on the same float32 inputs
    x = (np.random.RandomState(0).uniform(-150, 150, 5000)).astype(np.float32)
    pix, sub = orc.subpixel_coord(x, 8)
    xs = torch.floor(torch.from_numpy(x) * 8.0).to(torch.int32)
    tp = torch.div(xs, 8, rounding_mode='floor')
    np.testing.assert_array_equal(pix, tp.numpy())
    np.testing.assert_array_equal(sub, (xs - tp * 8).numpy())
    assert bl.shape == (2016, 3)


@pytest.mark.parametrize('world', [2, 8])
def test_bench_torchrun_rehearsal(world):
    """`bench.py --gpus N` under torchrun exactly as the driver launches it, with `--rehearse`
    (gloo, CPU tensors, no device work): rendezvous, broadcast, per-rank channels, barriers,
    max-over-ranks timing, per-rank statistics and ONE JSON line from rank 0; N = 2 and the
    driver's largest N = 8."""
    import json
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--steps', '3', '--warmup', '1',
           '--rehearse']
    env = dict(os.environ, OMP_NUM_THREADS='1')
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == world and out['steps'] == 3 and out['scaling'] == 'weak'
    want = [7, 3] if world == 2 else [7, 6, 5, 4, 3, 2, 1, 0]
    assert out['config']['band_channels'] == want       # rank 0 keeps the N = 1 channel
    assert out['ms_per_step'] >= float(world)           # the slowest rank (N ms per step) sets it
    assert [p['rank'] for p in out['per_rank']] == list(range(world))
    assert [p['band_channel'] for p in out['per_rank']] == want
    assert [p['device'] for p in out['per_rank']] == list(range(world))
    assert out['per_rank'][0]['ms_per_step'] <= out['per_rank'][-1]['ms_per_step']


def test_bench_channel_assignment_is_the_same_workload_for_every_n():
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 4, 8):
        chans = [bench.rank_channel(r, world) for r in range(world)]
        assert chans[0] == 7 and len(set(chans)) == world
        assert all(0.94 < bench.channel_scale(c) <= 1.0 for c in chans)
    assert bench.channel_scale(7) == 1.0
