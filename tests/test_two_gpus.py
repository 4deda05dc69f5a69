"""The N > 1 path on REAL links: RCCL with two ranks on two GPUs.  The gpurun
box has ONE GPU, so these tests skip themselves there (`torch.cuda.
device_count() < 2`); on the first node with two MI355X they check what has
only ever run with one rank or over gloo: `bench.py --gpus 2` with the "nccl"
backend (every collective of the line on RCCL) and wb2_comm_init_rank /
wb2_time_mean_allreduce through the C ABI with two ranks.  No scaling curve
exists yet -- nothing here simulates one."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _two_gpus():
  import torch
  return torch.cuda.is_available() and torch.cuda.device_count() >= 2


def test_bench_two_ranks_over_rccl():
  if not _two_gpus():
    pytest.skip('needs two GPUs (RCCL ranks cannot share a device)')
  env = dict(os.environ)
  for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT',
            'WB2_BENCH_SAME_GPU', 'WB2_BENCH_DIST_BACKEND'):
    env.pop(k, None)
  out = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps',
       '5', '--warmup', '2', '--no-secondary', '--no-cpu-baseline'],
      env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
  assert out.returncode == 0, out.stderr[-2000:]
  line = json.loads(out.stdout.strip().splitlines()[-1])
  assert line['n_gpus'] == 2 and line['steps'] == 5
  assert line['ranks']['world_size_seen'] == 2
  assert line['ranks']['backend'].startswith('rccl')
  assert all(v == 'rccl' for v in line['ranks']['collectives'].values())
  assert len(line['ranks']['ms_per_step_per_rank']) == 2
  assert line['value'] > 0 and line['full_suite']['value'] > 0
  assert line['map_allreduce']['ms'] > 0


def test_strong_scaling_2920_units_over_two_ranks():
  """BASELINE configs[4] as written: a FIXED job of 2920 (init, lead) units in
  contiguous shards over the ranks (evaluation.shard_bounds: 1460 + 1460), 16
  units per step => K = ceil(1460 / 16) = 92 steps of the largest shard, the
  last one partial (4 units); `value` is the 2920-unit job over the max-over-
  ranks time, every collective on RCCL."""
  if not _two_gpus():
    pytest.skip('needs two GPUs (RCCL ranks cannot share a device)')
  from weatherbench2_amd.evaluation import shard_bounds
  assert [shard_bounds(2920, 2, r) for r in (0, 1)] == [(0, 1460), (1460, 2920)]
  env = dict(os.environ)
  for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT',
            'WB2_BENCH_SAME_GPU', 'WB2_BENCH_DIST_BACKEND'):
    env.pop(k, None)
  out = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
       '--total-units', '2920', '--warmup', '3', '--no-cpu-baseline'],
      env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
  assert out.returncode == 0, out.stderr[-2000:]
  last = out.stdout.strip().splitlines()[-1]
  assert len(last.encode()) < 4096
  line = json.loads(last)
  assert line['scaling'] == 'strong' and line['n_gpus'] == 2
  assert line['config']['total_units'] == 2920
  assert line['steps'] == 92   # ceil(1460 / 16)
  pts = 2920 * 13 * 721 * 1440
  assert abs(line['value'] - pts / (line['ms_per_step'] * line['steps'] * 1e-3)
             ) < 1e-6 * line['value']
  assert line['ranks']['world_size_seen'] == 2
  assert line['ranks']['backend'].startswith('rccl')
  assert all(v == 'rccl' for v in line['ranks']['collectives'].values())
  assert len(line['ranks']['ms_per_step_per_rank']) == 2
  assert line['full_suite']['scaling'] == 'strong'
  assert line['full_suite']['value'] > 0


def test_c_abi_allreduce_with_two_ranks(tmp_path):
  if not _two_gpus():
    pytest.skip('needs two GPUs (RCCL ranks cannot share a device)')
  script = tmp_path / 'rank.py'
  script.write_text(textwrap.dedent('''
      import os, sys, time
      import numpy as np, torch
      sys.path.insert(0, sys.argv[3])
      from weatherbench2_amd import engine
      rank, path = int(sys.argv[1]), sys.argv[2]
      torch.cuda.set_device(rank)
      if rank == 0:
        uid = engine.comm_unique_id()
        with open(path + '.tmp', 'wb') as f:
          f.write(uid)
        os.replace(path + '.tmp', path)
      else:
        for _ in range(600):
          if os.path.exists(path):
            break
          time.sleep(0.1)
        uid = open(path, 'rb').read()
      comm = engine.comm_init_rank(uid, 2, rank)
      dev = torch.device('cuda', rank)
      total = torch.arange(1000, dtype=torch.float64, device=dev) * (rank + 1)
      count = torch.full_like(total, float(rank + 2))
      engine.time_mean_allreduce(total, count, comm)
      torch.cuda.synchronize()
      want = torch.arange(1000, dtype=torch.float64, device=dev) * 3
      assert torch.equal(total, want), total[:4]
      assert torch.equal(count, torch.full_like(count, 5.0))
      engine.comm_destroy(comm)
      print('rank', rank, 'ok')
  '''))
  uid_path = str(tmp_path / 'uid')
  procs = [subprocess.Popen([sys.executable, str(script), str(r), uid_path,
                             ROOT], stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True)
           for r in range(2)]
  for r, p in enumerate(procs):
    out, err = p.communicate(timeout=600)
    assert p.returncode == 0, err[-2000:]
    assert f'rank {r} ok' in out
