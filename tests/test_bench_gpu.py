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


def test_torchrun_single_rank_uses_rccl():
  """The driver's launch line with ONE rank: the process group is RCCL
  (torch "nccl"), so init with device_id, barrier, the [sum, count] all-reduce,
  the MAX of the timings and the 864 MB map all-reduce all run through RCCL on
  the 1-GPU box."""
  import socket
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  env = dict(os.environ)
  for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT',
            'WB2_BENCH_SAME_GPU', 'WB2_BENCH_DIST_BACKEND'):
    env.pop(k, None)
  out = subprocess.run(
      [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
       '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port',
       str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3',
       '--warmup', '1', '--units', '2', '--pool', '4', '--ramp-ms', '0',
       '--no-cpu-baseline', '--no-api', '--no-secondary', '--no-pcie',
       '--no-pmc'], env=env, cwd=ROOT, capture_output=True,
      text=True, timeout=900)
  assert out.returncode == 0, out.stderr[-2000:]
  line = json.loads([l for l in out.stdout.strip().splitlines()
                     if l.startswith('{')][-1])
  assert line['n_gpus'] == 1 and line['ranks']['world_size_seen'] == 1
  assert line['ranks']['backend'].startswith('rccl')
  assert line['config']['launcher'] == 'torch.distributed.run'
  assert line['map_allreduce']['ms'] > 0
  assert line['full_suite']['value'] > 0 and line['value'] > 0
  assert all(v == 'rccl' for v in line['ranks']['collectives'].values())


def test_strong_scaling_shards_a_fixed_number_of_units():
  """--total-units (BASELINE configs[4] is a FIXED 2920-unit job): 5 units
  over 2 ranks = contiguous shards of 3 and 2, two units per step => K = 2
  steps (the second one partial on rank 0, empty... no: 1 + 0 units), value =
  5 units / time."""
  env = dict(os.environ, WB2_BENCH_SAME_GPU='1', WB2_BENCH_DIST_BACKEND='gloo')
  for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
    env.pop(k, None)
  out = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
       '--total-units', '5', '--warmup', '1', '--units', '2', '--pool', '4',
       '--ramp-ms', '0', '--no-cpu-baseline'], env=env, cwd=ROOT,
      capture_output=True, text=True, timeout=900)
  assert out.returncode == 0, out.stderr[-2000:]
  line = json.loads(out.stdout.strip().splitlines()[-1])
  assert line['scaling'] == 'strong' and line['n_gpus'] == 2
  assert line['steps'] == 2 and line['config']['total_units'] == 5
  pts = 5 * 13 * 721 * 1440
  assert abs(line['value'] - pts / (line['ms_per_step'] * 2e-3)) < 1e-6 * line[
      'value']
  assert line['full_suite']['scaling'] == 'strong'
  assert line['full_suite']['value'] > 0
  assert line['ranks']['world_size_seen'] == 2


def test_default_line_carries_every_baseline_config():
  """What the driver runs (`python bench.py --gpus 1`, here with few steps and
  without the CPU and PMC legs): configs[1] as `value`, configs[2] as
  `ensemble`, configs[3] as `spectrum` (+ its two sub-legs), configs[4] as
  `full_suite`, the K1 variants and the PCIe-inclusive legs -- each with its
  own roofline."""
  env = dict(os.environ)
  for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT',
            'WB2_BENCH_SAME_GPU', 'WB2_BENCH_DIST_BACKEND'):
    env.pop(k, None)
  out = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '10',
       '--warmup', '2', '--ramp-ms', '0', '--no-cpu-baseline', '--no-pmc'],
      env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
  assert out.returncode == 0, out.stderr[-2000:]
  line = json.loads(out.stdout.strip().splitlines()[-1])
  assert line['n_gpus'] == 1 and line['scaling'] == 'weak'
  for key in ('ensemble', 'spectrum'):
    leg = line[key]
    assert 'error' not in leg, leg
    r = leg['roofline']
    assert r['kernel_ms'] > 0 and 0 < r['frac'] < 1
    assert r['algorithmic_bytes_per_launch'] > 0
  for sub in ('materialized', 'time_mean'):
    assert 0 < line['spectrum'][sub]['roofline']['frac'] < 1
  assert 'configs[2]' in line['ensemble']['config']['workload']
  assert 'configs[3]' in line['spectrum']['config']['workload']
  variants = line['variants']
  assert sorted(variants) == ['det_no_acc', 'f64_inputs', 'lonlat',
                              'official16_landmask', 'skipna', 'wind']
  assert variants['official16_landmask']['regions'] == 16
  assert variants['official16_landmask']['weight_field'] is True
  for v in variants.values():
    assert v['kernel_ms'] > 0 and 0 < v['frac'] < 1
  pcie = line['pcie_inclusive']
  assert pcie['all_inputs_over_pcie']['h2d_GBps'] > 1
  assert pcie['forecast_over_pcie_truth_clim_resident']['value'] > pcie[
      'all_inputs_over_pcie']['value']
