"""Benchmark of the MI355X metric-evaluation hot path (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" = one fused pass of the deterministic suite (MSE + RMSE + MAE + Bias +
ACC for the 13 predefined slice regions of scripts/evaluate.py:345-374) over a
batch of `--units` (init, lead) units of 13 x 721 x 1440 float32 points each,
i.e. BASELINE config 2, followed by the running init-time mean.  Forecast,
truth and climatology are synthetic N(0,1) and already resident in HBM when the
timed region starts; truth and climatology slabs are gathered by index tables
exactly like the product path does for valid_time / (dayofyear, hour).  Units
are drawn round-robin from a pool much larger than the 256 MiB Infinity Cache.

For N > 1 (torchrun, one rank per GPU) every rank evaluates its own shard of
init-times (weak scaling) and the only exchange is one RCCL all-reduce of the
[sum, count] accumulators at the end (evaluation.py:740-744's xbeam.Mean).

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

N_LEV, N_LAT, N_LON = 13, 721, 1440
PTS_PER_UNIT = N_LEV * N_LAT * N_LON
BYTES_PER_PT = 12.0  # forecast 4 + truth 4 + climatology 4
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def predefined_regions():
  from weatherbench2_amd.regions import SliceRegion as R
  return {
      'global': R(),
      'tropics': R(lat_slice=slice(-20, 20)),
      'extra-tropics': R(lat_slice=[slice(None, -20), slice(20, None)]),
      'northern-hemisphere': R(lat_slice=slice(20, None)),
      'southern-hemisphere': R(lat_slice=slice(None, -20)),
      'europe': R(lat_slice=slice(35, 75),
                  lon_slice=[slice(360 - 12.5, None), slice(0, 42.5)]),
      'north-america': R(lat_slice=slice(25, 60),
                         lon_slice=slice(360 - 120, 360 - 75)),
      'north-atlantic': R(lat_slice=slice(25, 65),
                          lon_slice=slice(360 - 70, 360 - 10)),
      'north-pacific': R(lat_slice=slice(25, 60),
                         lon_slice=slice(145, 360 - 130)),
      'east-asia': R(lat_slice=slice(25, 60), lon_slice=slice(102.5, 150)),
      'ausnz': R(lat_slice=slice(-45, -12.5), lon_slice=slice(120, 175)),
      'arctic': R(lat_slice=slice(60, 90)),
      'antarctic': R(lat_slice=slice(-90, -60)),
  }


def measured_traffic(workload: str, **match):
  """HBM bytes per launch of the dominant kernel as measured with rocprofv3
  PMC counters for exactly this launch size (profiles/r01_pmc_traffic.*), or
  None when the run uses a different configuration."""
  try:
    table = json.load(open(os.path.join(ROOT, 'profiles',
                                        'r01_pmc_traffic.json')))
    entry = table[workload]
    if all(entry.get(k) == v for k, v in match.items()):
      return entry['traffic_bytes']
  except (OSError, KeyError, ValueError):
    pass
  return None


def ramp(step_fn, ms: float) -> None:
  """Untimed clock ramp: the same step, enqueued back to back for `ms` of wall
  time before the W warmup steps, WITHOUT draining the queue (the host only
  waits for events a few batches behind).  A per-launch trace shows why: after
  any idle gap the kernels run 5-8 % slower for ~15 ms, and a 20-step timed
  region lasts ~10 ms."""
  import torch
  if ms <= 0:
    return
  t0 = time.perf_counter()
  pending = []
  while (time.perf_counter() - t0) * 1e3 < ms:
    for _ in range(8):
      step_fn()
    ev = torch.cuda.Event()
    ev.record()
    pending.append(ev)
    if len(pending) > 4:  # stay <= 32 steps ahead of the GPU
      pending.pop(0).synchronize()


def cpu_baseline(seconds: float = 10.0, processes: int = 0) -> dict:
  """Times the NumPy oracle (the restated reference path: one metric x one
  region at a time, like evaluation.py:408-435) on this box's host cores:
  first one process, then `processes` at once ("(i) 1 process,
  (ii) nproc processes over init-time shards"); `value` is the aggregate of
  the multi-process leg.  Runs oracle/cpu_baseline.py in subprocesses (no
  torch, no product code in them)."""
  import subprocess
  import sys
  ncpu = os.cpu_count() or 1
  if processes <= 0:
    processes = max(1, min(ncpu // 2, 32))  # physical cores, memory-bound work
  cmd = [sys.executable, '-m', 'oracle.cpu_baseline', '--seconds', str(seconds)]
  env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1',
             MKL_NUM_THREADS='1')

  def launch(n):
    procs = [subprocess.Popen(cmd + ['--seed', str(i)], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, text=True)
             for i in range(n)]
    outs = []
    for pr in procs:
      stdout, _ = pr.communicate(timeout=seconds * 20 + 120)
      if pr.returncode != 0:
        raise RuntimeError('oracle.cpu_baseline failed')
      outs.append(json.loads(stdout.strip().splitlines()[-1]))
    return outs

  one = launch(1)[0]
  rate_1 = one['points'] / one['seconds']
  many = launch(processes) if processes > 1 else [one]
  rate_n = sum(o['points'] / o['seconds'] for o in many)
  return {
      'value': rate_n, 'unit': 'grid-point-evals/s', 'cores': len(many),
      'kind': 'port', 'value_1core': rate_1,
      'sample': (f'NumPy oracle (xarray-semantics restatement; the reference '
                 f'itself needs xarray, absent here): whole units of '
                 f'{N_LEV} levels x 721 x 1440 f32, {one["metrics"]} metrics x '
                 f'{one["regions"]} regions evaluated one (metric, region) at '
                 f'a time like evaluation.py:408-435; 1 process did '
                 f'{one["units"]} unit(s) in {one["seconds"]:.1f} s, then '
                 f'{len(many)} single-threaded processes at once did '
                 f'{sum(o["units"] for o in many)} units in '
                 f'{max(o["seconds"] for o in many):.1f} s (own data each, like '
                 f'Beam workers over init-time shards); host has {ncpu} '
                 f'logical cores'),
  }


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=200)
  ap.add_argument('--warmup', type=int, default=20)
  ap.add_argument('--ramp-ms', type=float, default=60.0,
                  help='before the W warmup steps, keep running the same step '
                       'untimed for this long so that the GPU has left its idle '
                       'clocks (a 20-step timed region is only ~10 ms; measured: '
                       '+7 %% throughput once ramped); 0 disables')
  ap.add_argument('--units', type=int, default=16,
                  help='(init, lead) units per step and per GPU')
  ap.add_argument('--pool', type=int, default=48,
                  help='distinct units resident in HBM per input')
  ap.add_argument('--rows-per-chunk', type=int, default=0,
                  help='0 = plan.auto_rows_per_chunk (32 at the default batch)')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--pcie', action='store_true',
                  help='also time the step with its inputs arriving from '
                       'pinned host memory (reported as pcie_inclusive, never '
                       'as value)')
  ap.add_argument('--workload', default='deterministic',
                  choices=['deterministic', 'ensemble', 'spectrum',
                           'spectrum_mean'],
                  help='deterministic = BASELINE configs[1] (the headline '
                       'metric); ensemble / spectrum = configs[2] / [3], '
                       'spectrum_mean = the time-mean pipeline of the spectrum '
                       'script with the mean fused; used for profiles/ and '
                       'DESIGN.md')
  ap.add_argument('--members', type=int, default=50)
  args = ap.parse_args()
  if args.workload != 'deterministic':
    return secondary(args)

  import torch
  import torch.distributed as dist
  from weatherbench2_amd import _lib, engine, plan as plan_lib

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if args.gpus != world:
    if world == 1 and args.gpus > 1:
      raise SystemExit('--gpus N > 1 must be launched with torch.distributed.run')
  if os.environ.get('WB2_BENCH_SAME_GPU'):  # smoke test of N > 1 on one GPU
    local_rank = 0
  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)
  # RCCL ("nccl") is the production backend; WB2_BENCH_DIST_BACKEND=gloo exists
  # only so the N > 1 control flow can be smoke-tested on a 1-GPU box.
  backend = os.environ.get('WB2_BENCH_DIST_BACKEND', 'nccl')
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend == 'nccl':
      dist.init_process_group('nccl', device_id=dev)
    else:
      dist.init_process_group(backend)

  def all_reduce(tensor, op=None):
    op = op or dist.ReduceOp.SUM
    if backend == 'nccl':
      dist.all_reduce(tensor, op=op)
      return tensor
    host = tensor.cpu()
    dist.all_reduce(host, op=op)
    return host.to(tensor.device)

  lat = np.linspace(-90, 90, N_LAT)
  lon = np.linspace(0, 360, N_LON, endpoint=False)
  units, pool = args.units, max(args.pool, args.units)
  if not args.rows_per_chunk:
    args.rows_per_chunk = plan_lib.auto_rows_per_chunk(N_LAT, units * N_LEV)
  pl = plan_lib.build_plan(lat, lon, plan_lib.LATLON, predefined_regions(), dev,
                           rows_per_chunk=args.rows_per_chunk)
  gen = torch.Generator(device=dev).manual_seed(1234 + rank)
  mk = lambda: torch.randn((pool * N_LEV, N_LAT, N_LON), generator=gen,
                           device=dev, dtype=torch.float32)
  fpool, tpool, cpool = mk(), mk(), mk()
  n_outer = units * N_LEV
  nr = pl.n_region
  total = torch.zeros((_lib.NMETRIC * nr, N_LEV), dtype=torch.float64, device=dev)
  count = torch.zeros_like(total)
  lev = torch.arange(N_LEV, device=dev, dtype=torch.int64)

  def tables(step):
    """Slab tables of this step: forecast units are consecutive pool entries,
    truth / climatology are gathered through (different) offsets."""
    u = (step * units + torch.arange(units, device=dev)) % pool
    fu = (u[:, None] * N_LEV + lev[None]).reshape(-1)
    tu = (((u + 7) % pool)[:, None] * N_LEV + lev[None]).reshape(-1)
    cu = (((u * 5 + 3) % pool)[:, None] * N_LEV + lev[None]).reshape(-1)
    return fu.contiguous(), tu.contiguous(), cu.contiguous()

  all_tables = [tables(s) for s in range(args.warmup + args.steps)]
  k1_events = []

  def step(i, timed):
    fu, tu, cu = all_tables[i]
    if timed:
      e0 = torch.cuda.Event(enable_timing=True)
      e1 = torch.cuda.Event(enable_timing=True)
      engine.K1_EVENTS = (e0, e1)
      k1_events.append((e0, e1))
    else:
      engine.K1_EVENTS = None
    metrics, _ = engine.stream_reduce(
        pl, _lib.MODE_DET_ACC, [fpool, tpool, cpool], [fu, tu, cu], n_outer,
        skipna=False)
    # running init-time mean: (metric*region, unit, level)
    engine.time_accumulate(metrics.view(_lib.NMETRIC * nr, units, N_LEV), 1,
                           False, total, count)

  # Touch every op of the timed region once: on a cold box the first use of a
  # torch kernel (the final division, the all-reduce) loads its code object,
  # which costs tens of ms and is not part of the hot path.
  step(0, False)
  _ = (total / count).sum().item()
  if world > 1:
    all_reduce(torch.stack([total, count]))
  # From here to the timed region the GPU never runs dry: ramp and warmup are
  # enqueued back to back and the only wait is the contract's synchronize.
  ramp(lambda: step(0, False), args.ramp_ms)
  for i in range(args.warmup):
    step(i, False)
  total.zero_()
  count.zero_()
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  g0 = torch.cuda.Event(enable_timing=True)
  g1 = torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  g0.record()
  for i in range(args.steps):
    step(args.warmup + i, True)
  g1.record()
  if world > 1:
    packed = all_reduce(torch.stack([total, count]))  # the path's only exchange
    total, count = packed[0], packed[1]
  mean = total / count
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  if world > 1:
    tmax = all_reduce(torch.tensor([dt], dtype=torch.float64, device=dev),
                      dist.ReduceOp.MAX)
    dt = float(tmax.item())
  assert torch.isfinite(mean).all()

  k1_ms = [a.elapsed_time(b) for a, b in k1_events]
  if os.environ.get('WB2_BENCH_TRACE'):  # per-launch durations (diagnostics)
    print('k1_ms', ' '.join(f'{x:.3f}' for x in k1_ms), file=sys.stderr)
  k1_avg_s = float(np.mean(k1_ms)) / 1e3
  pts_step = units * PTS_PER_UNIT
  achieved = pts_step * BYTES_PER_PT / k1_avg_s / 1e9
  out = {
      'metric': 'grid-point-evals/sec (721x1440x13)',
      'value': world * pts_step * args.steps / dt,
      'unit': 'grid-point-evals/s',
      'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
      'ms_per_step': dt / args.steps * 1e3,
      'gpu_ms_per_step': g0.elapsed_time(g1) / args.steps,
      'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
      'dtype': 'f32 (elementwise) + f64 (sums)', 'data': 'synthetic',
      'config': {
          'workload': ('BASELINE configs[1]: 721x1440x13 f32, deterministic '
                       'MSE+RMSE+MAE+Bias+ACC, 13 predefined slice regions, '
                       'running init-time mean'),
          'units_per_step_per_gpu': units, 'pool_units': pool,
          'regions': nr, 'rows_per_chunk': args.rows_per_chunk,
          'parallelism': f'init-time shards x{world}, 1 all-reduce of [sum,count]',
      },
      'roofline': {
          'bound': 'hbm', 'kernel': 'stream_partials_kernel<float,4,DET_ACC>',
          'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
          'frac': achieved / HBM_PEAK_GBPS,
          'frac_of_measured_copy_6290': achieved / 6290.0,
          'kernel_ms': k1_avg_s * 1e3,
          'algorithmic_bytes_per_launch': pts_step * BYTES_PER_PT,
          'traffic': measured_traffic('deterministic', units_per_launch=units,
                                      regions=nr),
      },
  }
  if rank == 0 and world == 1 and args.pcie:
    # The same step when the boundary hands over HOST buffers: forecast, truth
    # and climatology units cross PCIe (pinned memory, one copy stream) before
    # the fused pass.  Reported beside `value`, never as it.
    n_el = n_outer * N_LAT * N_LON
    host = [torch.empty((n_el,), dtype=torch.float32).pin_memory()
            for _ in range(3)]
    for h in host:
      h.normal_()
    stage = [torch.empty((n_outer, N_LAT, N_LON), dtype=torch.float32,
                         device=dev) for _ in range(3)]
    reps = 5

    def host_step():
      for h, d in zip(host, stage):
        d.view(-1).copy_(h, non_blocking=True)
      m, _ = engine.stream_reduce(pl, _lib.MODE_DET_ACC, stage,
                                  [None, None, None], n_outer, skipna=False)
      engine.time_accumulate(m.view(_lib.NMETRIC * nr, units, N_LEV), 1, False,
                             total, count)
    engine.K1_EVENTS = None
    host_step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(reps):
      host_step()
    torch.cuda.synchronize()
    dt_h = (time.perf_counter() - t1) / reps
    out['pcie_inclusive'] = {
        'value': pts_step / dt_h, 'unit': 'grid-point-evals/s',
        'ms_per_step': dt_h * 1e3,
        'h2d_GBps': pts_step * BYTES_PER_PT / dt_h / 1e9,
        'note': 'inputs start in pinned host memory; serial copy + compute',
    }
  if rank == 0:
    if world == 1 and not args.no_cpu_baseline:
      try:
        out['cpu_baseline'] = cpu_baseline()
      except Exception as e:  # never lose the GPU line to the baseline leg
        from oracle import cpu_baseline as ocb
        one = ocb.run(10.0)
        out['cpu_baseline'] = {
            'value': one['points'] / one['seconds'],
            'unit': 'grid-point-evals/s', 'cores': 1, 'kind': 'port',
            'sample': (f'NumPy oracle in-process, {one["units"]} unit(s) in '
                       f'{one["seconds"]:.1f} s (the multi-process leg failed: '
                       f'{type(e).__name__}: {e})')}
    print(json.dumps(out))
  if world > 1:
    dist.destroy_process_group()


def secondary(args):
  """BASELINE configs[2] (50-member ensemble) and configs[3] (zonal spectrum)
  on one GPU: same timing discipline, their own roofline."""
  import torch
  from weatherbench2_amd import _lib, engine, plan as plan_lib
  dev = torch.device('cuda', 0)
  torch.cuda.set_device(dev)
  lat = np.linspace(-90, 90, N_LAT)
  lon = np.linspace(0, 360, N_LON, endpoint=False)
  gen = torch.Generator(device=dev).manual_seed(99)
  events = []
  if args.workload == 'ensemble':
    m = args.members
    n_slab = 13            # one unit of 13 levels per step
    pool = 4               # 4 x 13 x 50 x 4.15 MB = 10.8 GB >> Infinity Cache
    pl = plan_lib.build_plan(lat, lon, plan_lib.LATLON, predefined_regions(),
                             dev, rows_per_chunk=(args.rows_per_chunk or
                                                  plan_lib.ENSEMBLE_ROWS_PER_CHUNK))
    ens = torch.randn((m, pool * n_slab, N_LAT, N_LON), generator=gen,
                      device=dev)
    truth = torch.randn((pool * n_slab, N_LAT, N_LON), generator=gen,
                        device=dev)
    stride = pool * n_slab * N_LAT * N_LON
    tabs = [(torch.arange(n_slab, device=dev) + (i % pool) * n_slab)
            for i in range(pool)]
    pts = n_slab * N_LAT * N_LON
    bytes_per_pt = (m + 1) * 4.0

    def step(i, timed):
      if timed:
        ev = (torch.cuda.Event(enable_timing=True),
              torch.cuda.Event(enable_timing=True))
        engine.K1_EVENTS = ev
        events.append(ev)
      else:
        engine.K1_EVENTS = None
      tab = tabs[i % pool]
      engine.ensemble_reduce(pl, ens, stride, m, tab, truth, tab, n_slab, False)
    kernel = f'ens_partials_kernel<float,64,{m if m == 50 else 0}>'
    workload = (f'BASELINE configs[2]: 721x1440x13 f32, {m}-member CRPS + '
                'spread/skill + ensemble-mean MSE + variance + debiased MSE, '
                '13 regions')
  else:
    units = 8
    pool = 6
    x = torch.randn((pool * units, N_LEV, N_LAT, N_LON), generator=gen,
                    device=dev)
    from weatherbench2_amd.derived_variables import ZonalEnergySpectrum
    circ = torch.as_tensor(ZonalEnergySpectrum._circumference(lat)).to(dev)
    w_lat = torch.as_tensor(plan_lib.get_lat_weights(lat)).to(dev)
    pts = units * PTS_PER_UNIT
    bytes_per_pt = 4.0 + (N_LON // 2 + 1) * 8.0 / N_LON
    if args.workload == 'spectrum_mean':
      bytes_per_pt = 4.0 + (N_LON // 2 + 1) * 8.0 / N_LON / units

    def step(i, timed):
      xs = x[(i % pool) * units:(i % pool + 1) * units]
      if timed:
        ev = (torch.cuda.Event(enable_timing=True),
              torch.cuda.Event(enable_timing=True))
        ev[0].record()
      if args.workload == 'spectrum_mean':
        # the script's pipeline (compute_zonal_energy_spectrum.py:234): the
        # time mean of the spectrum, fused -- the 8 units act as 8 times
        engine.zonal_spectrum(xs, circ, N_LAT, n_time=units)
        if timed:
          ev[1].record()
          events.append(ev)
        return
      spec = engine.zonal_spectrum(xs, circ, N_LAT)
      if timed:
        ev[1].record()
        events.append(ev)
      # configs[3] "+ lat-weighted reduce": area-weighted mean of the spectrum
      # over latitude (K7), [units * 13, 721 lat, 721 bins] -> [units * 13, 721]
      total, _, count = engine.axis_moments(
          spec.reshape(units * N_LEV, N_LAT, N_LON // 2 + 1), units * N_LEV,
          N_LAT, N_LON // 2 + 1, w_lat, False)
      lat_mean = total / count
    kernel = ('fused_spectrum_kernel<720,TIME> (LDS real FFT, time mean in '
              'registers)' if args.workload == 'spectrum_mean' else
              'fused_spectrum_kernel<720> (LDS real FFT + power epilogue); '
              'WB2HIP_SPECTRUM_BACKEND=rocfft selects rocFFT C2C + power_kernel')
    workload = ('BASELINE configs[3]: zonal energy spectrum of 8 units of '
                '13x721x1440 f32 per step, per-unit spectrum materialised, then '
                'its area-weighted latitude mean (K7; the roofline entry is the '
                'spectrum kernel alone)')
  ramp(lambda: step(0, False), args.ramp_ms)
  for i in range(args.warmup):
    step(i, False)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for i in range(args.steps):
    step(args.warmup + i, True)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  k_s = float(np.mean([a.elapsed_time(b) for a, b in events])) / 1e3
  achieved = pts * bytes_per_pt / k_s / 1e9
  print(json.dumps({
      'metric': 'grid-point-evals/sec (721x1440x13)',
      'value': pts * args.steps / dt, 'unit': 'grid-point-evals/s',
      'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
      'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
      'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
      'config': {'workload': workload},
      'roofline': {'bound': 'hbm', 'kernel': kernel, 'achieved': achieved,
                   'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                   'frac': achieved / HBM_PEAK_GBPS, 'kernel_ms': k_s * 1e3,
                   'algorithmic_bytes_per_launch': pts * bytes_per_pt,
                   'traffic': (measured_traffic('ensemble', slabs_per_launch=13,
                                                members=args.members)
                               if args.workload == 'ensemble' else
                               measured_traffic(args.workload,
                                                units_per_launch=8))}}))


if __name__ == '__main__':
  main()
