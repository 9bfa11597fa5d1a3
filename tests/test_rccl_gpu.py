"""The RCCL code path on the one GPU a test box has: a `nccl` process group of ONE rank on cuda:0
and every collective helper of katsdpimager_amd.parallel on device tensors (broadcast of the
channel-independent inputs of SURVEY 8e, timing maximum, statistics gather, the rank / device
check), and bench.py --force-dist, which takes the same branch the driver's N > 1 launches take.
The N = 2 logic itself is covered on the CPU (gloo) by tests/test_multiprocess.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tools'))
import torch
import torch.distributed as dist
from katsdpimager_amd import parallel
import synth
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(parallel.free_port()), RANK='0',
                  WORLD_SIZE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', device_id=dev)
assert dist.get_backend() == 'nccl' and parallel.world() == (0, 1)
n = 1 << 20
shared = dict(uvw=synth.track_uvw(n, dev), taper1d=torch.linspace(0, 1, 4096, device=dev))
want = {{k: v.clone() for k, v in shared.items()}}
parallel.broadcast_shared(shared, src=0)
torch.cuda.synchronize()
assert all(torch.equal(shared[k], want[k]) for k in shared)
assert parallel.max_over_ranks(1.25, dev) == 1.25
stats = parallel.gather_stats([3.0, 4.5, float(torch.cuda.current_device())], dev)
assert stats.shape == (1, 3) and stats.is_cuda and stats.tolist() == [[3.0, 4.5, 0.0]]
assert parallel.check_rank_devices(torch.cuda.current_device(), dev) == [0]
dist.barrier(device_ids=[0])
dist.destroy_process_group()
print(json.dumps(dict(ok=True, bytes=sum(v.numel() * v.element_size() for v in shared.values()))))
'''


def test_rccl_helpers_on_one_gpu(tmp_path):
    script = tmp_path / 'rccl_worker.py'
    script.write_text(WORKER.format(root=ROOT))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    last = json.loads(out.stdout.strip().splitlines()[-1])
    assert last['ok'] and last['bytes'] == (1 << 20) * 12 + 4096 * 4


def test_bench_force_dist_takes_the_rccl_branch():
    """bench.py --force-dist: process group, the 8e broadcast (UVW in metres + taper), the
    device-side gathers and barriers, on a reduced workload; the line's value and per-rank block
    come out as at N = 1."""
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'bench.py'), '--force-dist', '--vis', '4000000', '--steps',
         '2', '--warmup', '1', '--cpu-sample', '0', '--no-secondary'],
        capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([x for x in out.stdout.splitlines() if x.startswith('{"metric"')][-1])
    cfg = line['config']
    assert line['n_gpus'] == 1 and cfg['process_group'].startswith('nccl')
    assert cfg['broadcast_MB'] == round((4000000 * 12 + 4096 * 4) / 1e6, 1) and cfg['broadcast_ms'] > 0
    assert line['value'] > 0 and [p['device'] for p in line['per_rank']] == [0]
