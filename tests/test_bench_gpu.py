"""bench.py's N > 1 path on ONE GPU: `python bench.py --gpus 2` spawns its two
ranks itself (no torch.distributed.run wrapper); WB2_BENCH_SAME_GPU puts both on
device 0 and WB2_BENCH_DIST_BACKEND=gloo replaces RCCL (two ranks cannot share a
device in one RCCL communicator), so the sharding, the all-reduce of the
[sum, count] accumulators, the configs[4] leg and the JSON contract are
exercised end to end."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launched_two_ranks_on_one_gpu():
  env = dict(os.environ, WB2_BENCH_SAME_GPU='1', WB2_BENCH_DIST_BACKEND='gloo')
  for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
    env.pop(k, None)
  out = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps',
       '3', '--warmup', '1', '--units', '2', '--pool', '4', '--ramp-ms', '0',
       '--no-cpu-baseline'], env=env, cwd=ROOT, capture_output=True, text=True,
      timeout=900)
  assert out.returncode == 0, out.stderr[-2000:]
  line = json.loads(out.stdout.strip().splitlines()[-1])
  assert line['n_gpus'] == 2 and line['steps'] == 3 and line['warmup'] == 1
  assert line['ranks']['world_size_seen'] == 2
  assert len(line['ranks']['ms_per_step_per_rank']) == 2
  assert line['scaling'] == 'weak' and line['value'] > 0
  assert line['config']['launcher'] == 'self-spawned ranks'
  assert line['full_suite']['value'] > 0
  assert len(line['full_suite']['ms_per_step_per_rank']) == 2
  assert line['map_allreduce']['bytes_per_rank'] == 2 * 4 * 13 * 721 * 1440 * 8
  assert 'roofline' in line and line['roofline']['frac'] > 0
