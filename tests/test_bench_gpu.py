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


def _clean_env():
  env = dict(os.environ)
  for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT',
            'WB2_BENCH_SAME_GPU', 'WB2_BENCH_DIST_BACKEND'):
    env.pop(k, None)
  return env


def test_contract_line_is_one_small_json_object():
  """The driver's exact command shape (`python3 bench.py --gpus 1 --steps K
  --warmup W`, every leg on, few steps): the LAST stdout line is one JSON
  object of less than 4 kB that carries the contract's keys, `roofline` and
  `cpu_baseline`; everything else is in bench_detail.json beside bench.py."""
  detail_path = os.path.join(ROOT, 'bench_detail.json')
  if os.path.exists(detail_path):
    os.remove(detail_path)
  out = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1',
       '--steps', '8', '--warmup', '2'],
      env=_clean_env(), cwd=ROOT, capture_output=True, text=True, timeout=1500)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = out.stdout.strip().splitlines()
  last = lines[-1]
  assert len(last.encode()) < 4096, len(last)
  line = json.loads(last)
  assert json.loads(json.dumps(line)) == line
  for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup',
              'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
    assert key in line, key
  assert line['n_gpus'] == 1 and line['steps'] == 8 and line['warmup'] == 2
  assert line['scaling'] == 'weak' and line['vs_baseline'] is None
  assert 'configs[1]' in line['config']['workload']
  roof = line['roofline']
  assert roof['bound'] == 'hbm' and roof['peak'] == 8000.0
  assert 0 < roof['frac'] < 1 and roof['kernel_ms'] > 0
  assert roof['algorithmic_bytes_per_launch'] == 16 * 13 * 721 * 1440 * 12
  assert abs(roof['achieved'] - roof['algorithmic_bytes_per_launch'] /
             roof['kernel_ms'] / 1e6) < 1e-6 * roof['achieved']
  # traffic: live PMC bytes per launch (or an error string, never silence)
  assert roof['traffic'] or roof.get('traffic_live_error')
  if roof['traffic']:
    assert 0.95 < roof['traffic_over_algorithmic'] < 1.05
  cpu = line['cpu_baseline']
  assert cpu['value'] > 0 and cpu['cores'] >= 1 and cpu['kind'] == 'port'
  assert 'errors' not in line, line.get('errors')
  # one-number summaries of the other BASELINE configs
  assert 0 < line['ensemble']['frac'] < 1 and line['ensemble']['value'] > 0
  assert 0 < line['spectrum']['frac'] < 1
  assert 0 < line['spectrum']['materialized_frac'] < 1
  assert 0 < line['spectrum']['time_mean_frac'] < 1
  assert line['full_suite']['value'] > 0
  oc = line['api_official_chunk']
  assert oc['value'] > 0 and oc['chunk_by_chunk']['value'] > 0
  assert oc['host_fed']['h2d_GBps'] > 1
  # the full record
  assert line['detail'] == 'bench_detail.json'
  detail = json.load(open(detail_path))
  assert detail['value'] == line['value']
  for key in ('ensemble', 'spectrum'):
    r = detail[key]['roofline']
    assert r['kernel_ms'] > 0 and r['algorithmic_bytes_per_launch'] > 0
  assert 'configs[2]' in detail['ensemble']['config']['workload']
  assert 'configs[3]' in detail['spectrum']['config']['workload']
  pcie = detail['pcie_inclusive']
  assert pcie['all_inputs_over_pcie']['h2d_GBps'] > 1
  assert pcie['forecast_over_pcie_truth_clim_resident']['value'] > pcie[
      'all_inputs_over_pcie']['value']
  # the step is one C-ABI call: its timed ms tracks the kernel
  assert line['ms_per_step'] < 1.5 * roof['kernel_ms'] + 0.1


def test_detail_run_carries_every_instantiation():
  """`--detail`: K1's production instantiations (and the K3 / tier-2 sets) go
  to bench_detail.json; the last line stays the compact contract line."""
  out = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '6',
       '--warmup', '2', '--ramp-ms', '0', '--no-cpu-baseline', '--no-pmc',
       '--no-api', '--no-pcie', '--no-full-suite', '--detail'],
      env=_clean_env(), cwd=ROOT, capture_output=True, text=True, timeout=1500)
  assert out.returncode == 0, out.stderr[-2000:]
  last = out.stdout.strip().splitlines()[-1]
  assert len(last.encode()) < 4096
  detail = json.load(open(os.path.join(ROOT, 'bench_detail.json')))
  variants = detail['variants']
  assert sorted(variants) == ['det_no_acc', 'f64_inputs', 'lonlat',
                              'official16_landmask', 'skipna', 'wind']
  assert variants['official16_landmask']['regions'] == 16
  assert variants['official16_landmask']['weight_field'] is True
  for v in variants.values():
    assert v['kernel_ms'] > 0 and 0 < v['frac'] < 1
  assert 'error' not in detail['k3_variants'], detail['k3_variants']
  assert 'error' not in detail['tier2_variants'], detail['tier2_variants']
