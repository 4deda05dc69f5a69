"""Benchmark of the MI355X metric-evaluation hot path (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" = one fused pass of the deterministic suite (MSE + RMSE + MAE + Bias +
ACC for the 13 predefined slice regions of scripts/evaluate.py:345-374) over a
batch of `--units` (init, lead) units of 13 x 721 x 1440 float32 points each,
i.e. BASELINE configs[1], followed by the running init-time mean.  Forecast,
truth and climatology are synthetic N(0,1) and already resident in HBM when the
timed region starts; truth and climatology slabs are gathered by index tables
exactly like the product path does for valid_time / (dayofyear, hour).  Units
are drawn round-robin from a pool much larger than the 256 MiB Infinity Cache.

N > 1: one rank per GPU, every rank evaluates its own shard of init-times (weak
scaling) and the only exchange is one RCCL all-reduce of the [sum, count]
accumulators at the end (evaluation.py:740-744's xbeam.Mean).  Launched by
`torch.distributed.run` (RANK / WORLD_SIZE in the environment) the script is
one rank; WITHOUT that environment `--gpus N` spawns the N ranks itself.

`value` is always the deterministic suite (so the N = 1, 2, 4, 8 values form one
scaling curve); the same JSON line carries, measured in the same process:
  full_suite      BASELINE configs[4]: deterministic + probabilistic (50-member
                  K3) suites over the same number of units per step, sharded
                  the same way, both [sum, count] sets all-reduced;
  map_allreduce   N > 1: the bandwidth-relevant all-reduce of Spatial* maps;
and on one GPU (N = 1), each with its own `roofline`:
  ensemble        BASELINE configs[2]: the standalone 50-member K3 launch;
  spectrum        BASELINE configs[3]: zonal spectrum + latitude mean fused
                  (LATSEG + combine), with the sub-legs `materialized`
                  (ZonalEnergySpectrum.compute) and `time_mean` (the script's
                  pipeline);
  variants        K1's production instantiations: the official 16 regions
                  incl. three land-sea-mask regions, skipna, float64 inputs,
                  wind vectors, the lon-lat layout, no ACC;
  k3_variants     K3's: the member counts with kernels of their own (10 ... 56),
                  skipna with and without NaN patches, land masks, float64
                  (tools/k3_variants.py; medians of 3 interleaved repetitions);
  tier2_variants  the tier-2 kernels (Spatial* maps and their temporal sums,
                  SEEPS, Gaussian CRPS / thresholds, ensemble thresholds, rank
                  histogram, axis means), each against the HBM roofline of its
                  own algorithmic bytes (tools/tier2_variants.py);
  api             the same 16-unit chunk through the drop-in API
                  (_metric_and_region_loop, 5 metrics x 13 regions);
  api_official_chunk  the drop-in API at the reference's production chunking
                  (init_time=1,lead_time=1 chunks of 13 variables, the 16
                  official regions, mse + wind vectors + acc + bias + mae)
                  through evaluation.evaluate_chunks (its default window -- 24
                  chunks -- and batch_chunks=1 / 16 / 32):
                  tools/official_chunk.py;
  api_probabilistic  the `probabilistic` config at the chunking of its command
                  lines (IFS ENS 240 x 121, 50 members, init_time=1,lead_time=1,
                  13 regions) through evaluation.evaluate_chunks, in its default
                  windows and chunk by chunk: tools/official_probabilistic.py;
  pcie_inclusive  inputs arriving from pinned host memory through the
                  pipelined feeder (never `value`);
  cpu_baseline    the NumPy oracle on this box's host cores;
and `roofline.traffic` is collected LIVE: rocprofv3 --pmc FETCH_SIZE /
WRITE_SIZE around the benched K1 launch (tools/live_traffic.py; --no-pmc
skips it and quotes profiles/ instead).

`--total-units N` switches to STRONG scaling (BASELINE configs[4] literally:
2920 units, contiguous shards over the ranks, K = the steps of the largest
shard).

Prints ONE JSON line (rank 0): the compact contract line (< 4 kB).  The full
record of every leg goes to bench_detail.json beside this file.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
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
SPECTRUM_UNITS = 16      # units per launch of the spectrum workloads (as the headline)


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


def coprime(k: int, n: int) -> int:
  """The smallest integer >= k that is coprime to n: a unit stride k would map
  consecutive units to fewer than min(units, n) distinct ones otherwise, and a
  launch that reads a slab twice measures cache hits, not bandwidth."""
  import math
  while math.gcd(k, n) != 1:
    k += 1
  return k


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


class KernelTimer:
  """HIP-event pairs around the dominant kernel, recorded on the launch stream
  through engine's profiling hook."""

  def __init__(self):
    self.pairs = []
    self.current = None

  def __call__(self, when, kernel):
    import torch
    if when == 'begin':
      self.current = (torch.cuda.Event(enable_timing=True),
                      torch.cuda.Event(enable_timing=True))
      self.current[0].record()
    else:
      self.current[1].record()
      self.pairs.append(self.current)

  def mean_ms(self):
    return float(np.mean([a.elapsed_time(b) for a, b in self.pairs]))


def cpu_baseline(seconds: float = 4.0) -> dict:
  """Times the NumPy oracle (the restated reference path: one metric x one
  region at a time, like evaluation.py:408-435) on this box's host cores: one
  process, then one process per PHYSICAL core, then one per logical core
  ("(i) 1 process, (ii) nproc processes over init-time shards");
  `value` is the best aggregate.  Runs oracle/cpu_baseline.py in subprocesses
  (no torch, no product code in them).  The process counts are capped by the
  host's free memory (~0.6 GB per process)."""
  ncpu = os.cpu_count() or 1
  avail = None
  try:
    for line in open('/proc/meminfo'):
      if line.startswith('MemAvailable:'):
        avail = int(line.split()[1]) * 1024
  except OSError:
    pass
  cap = ncpu if avail is None else max(1, int(0.5 * avail / 0.6e9))
  cmd = [sys.executable, '-m', 'oracle.cpu_baseline', '--seconds', str(seconds)]
  env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1',
             MKL_NUM_THREADS='1')

  def launch(n):
    procs = [subprocess.Popen(cmd + ['--seed', str(i)], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, text=True)
             for i in range(n)]
    outs = []
    for pr in procs:
      stdout, _ = pr.communicate(timeout=seconds * 30 + 180)
      if pr.returncode != 0:
        raise RuntimeError('oracle.cpu_baseline failed')
      outs.append(json.loads(stdout.strip().splitlines()[-1]))
    return outs

  # 1 process, 32 (where round 1 found the memory-bandwidth knee), one per
  # logical core (a leg at one per physical core sat BELOW both in every run --
  # 15-19 M evals/s against 47-50 M and 50-58 M -- and cost 10 s: dropped)
  counts = sorted({1, min(cap, 32, ncpu), min(cap, ncpu)})
  legs = []
  for n in counts:
    outs = launch(n)
    legs.append({'processes': n,
                 'value': sum(o['points'] / o['seconds'] for o in outs),
                 'units': sum(o['units'] for o in outs),
                 'seconds': max(o['seconds'] for o in outs),
                 'metrics': outs[0]['metrics'], 'regions': outs[0]['regions']})
  best = max(legs, key=lambda l: l['value'])
  one = legs[0]
  return {
      'value': best['value'], 'unit': 'grid-point-evals/s',
      'cores': best['processes'], 'kind': 'port', 'value_1core': one['value'],
      'legs': [{'processes': l['processes'], 'value': l['value']} for l in legs],
      'logical_cores': ncpu,
      'sample_short': (f'NumPy oracle (port of the xarray path), whole 13x721x'
                       f'1440 f32 units, {one["metrics"]} metrics x '
                       f'{one["regions"]} regions, ~{seconds:.0f} s per leg at '
                       + '/'.join(str(l['processes']) for l in legs)
                       + ' single-threaded processes; best leg quoted'),
      'sample': (f'NumPy oracle (xarray-semantics restatement; the reference '
                 f'itself needs xarray, absent here): whole units of '
                 f'{N_LEV} levels x 721 x 1440 f32, {one["metrics"]} metrics x '
                 f'{one["regions"]} regions evaluated one (metric, region) at '
                 f'a time like evaluation.py:408-435; single-threaded '
                 f'processes with their own data (like Beam workers over '
                 f'init-time shards) ran for ~{seconds:.0f} s each at '
                 + ', '.join(f'{l["processes"]} proc: {l["units"]:.2f} units '
                             f'in {l["seconds"]:.1f} s' for l in legs)
                 + f'; host has {ncpu} logical cores'
                 + ('' if cap >= ncpu else f' (capped at {cap} processes by '
                    f'free host memory)')),
  }


def parse_args():
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
  ap.add_argument('--no-full-suite', action='store_true',
                  help='skip the configs[4] (deterministic + probabilistic) leg')
  ap.add_argument('--no-api', action='store_true',
                  help='skip the drop-in API leg (N = 1 only)')
  ap.add_argument('--pcie', action='store_true',
                  help='(kept for compatibility: the pcie_inclusive leg is part '
                       'of the default N = 1 line now)')
  ap.add_argument('--no-pcie', action='store_true',
                  help='skip the pcie_inclusive leg (N = 1 only)')
  ap.add_argument('--no-secondary', action='store_true',
                  help='skip the ensemble / spectrum / variants legs (N = 1)')
  ap.add_argument('--no-pmc', action='store_true',
                  help='do not collect roofline.traffic live with rocprofv3 '
                       '(N = 1 only; the profiles/ figure is quoted instead)')
  ap.add_argument('--total-units', type=int, default=0,
                  help='STRONG scaling: this many (init, lead) units in total '
                       '(BASELINE configs[4]: 2920), sharded contiguously over '
                       'the ranks (evaluation.shard_bounds); --steps is then '
                       'derived (the steps of the largest shard)')
  ap.add_argument('--detail', action='store_true',
                  help='also run every other instantiation against its own '
                       'roofline (K1 / K3 / tier-2 variants, all window sizes '
                       'of the official-chunk leg, live traffic of every '
                       'benched kernel): minutes of GPU time; the record goes '
                       'to bench_detail.json, the last stdout line stays the '
                       'compact contract line')
  ap.add_argument('--print-detail', action='store_true',
                  help='also print the full record (bench_detail.json) on a '
                       'line BEFORE the contract line')
  ap.add_argument('--launch-timeout', type=float, default=900.0,
                  help='seconds after which a self-launched job (and the '
                       'rendezvous / collectives of every rank) gives up '
                       'instead of hanging')
  ap.add_argument('--traffic-probe', nargs='?', const='deterministic',
                  default=None,
                  choices=['deterministic', 'official16_landmask', 'skipna',
                           'all'],
                  help='(tools/live_traffic.py) only run a few launches of the '
                       'benched K1 configuration -- or of one of its variants '
                       '-- for the PMC passes')
  ap.add_argument('--variants-only', action='store_true',
                  help="only K1's production instantiations (the `variants` "
                       'object of the default line), printed as one JSON line: '
                       'for A/B runs of kernel changes')
  ap.add_argument('--workload', default='deterministic',
                  choices=['deterministic', 'ensemble', 'spectrum',
                           'spectrum_materialized', 'spectrum_mean',
                           'spectrum_materialized_f64', 'spectrum_mean_f64'],
                  help='deterministic = BASELINE configs[1] (the headline '
                       'metric); ensemble / spectrum = configs[2] / [3] '
                       '(spectrum + area-weighted latitude mean, fused); '
                       'spectrum_materialized = ZonalEnergySpectrum.compute '
                       '(per-latitude spectra written) followed by the same '
                       'latitude mean; spectrum_mean = the time-mean pipeline of '
                       'the spectrum script with the mean fused; used for '
                       'profiles/ and DESIGN.md')
  ap.add_argument('--members', type=int, default=50)
  ap.add_argument('--spectrum-units', type=int, default=SPECTRUM_UNITS,
                  help='units of 13 x 721 x 1440 per launch of the spectrum '
                       'workloads')
  return ap.parse_args()


def self_launch(args) -> int:
  """`python bench.py --gpus N` outside torch.distributed.run: start the N
  ranks (one process per GPU, RCCL rendezvous on 127.0.0.1) and relay rank 0's
  JSON line."""
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  procs = []
  for rank in range(args.gpus):
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank),
               WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(port), WB2_BENCH_SELF_LAUNCHED='1')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    procs.append(subprocess.Popen(
        [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
        stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, text=True))
  # a rank that dies would leave the others waiting in a collective: watch all
  # of them and take the job down with the first failure
  import threading
  out_box = []
  reader = threading.Thread(
      target=lambda: out_box.append(procs[0].stdout.read()), daemon=True)
  reader.start()
  rc = 0
  deadline = time.monotonic() + args.launch_timeout
  while True:
    codes = [pr.poll() for pr in procs]
    failed = [c for c in codes if c not in (None, 0)]
    if failed or time.monotonic() > deadline:
      # a dead rank (or a rendezvous / collective that never completes) must
      # not hang the caller: take the whole job down
      rc = failed[0] if failed else 124
      if not failed:
        print(f'bench.py: ranks still running after {args.launch_timeout:.0f} s '
              '(--launch-timeout): killing the job', file=sys.stderr)
      for pr in procs:
        if pr.poll() is None:
          pr.kill()
      break
    if all(c == 0 for c in codes):
      break
    time.sleep(0.2)
  for pr in procs:
    pr.wait()
  reader.join(timeout=10)
  sys.stdout.write(out_box[0] if out_box else '')
  sys.stdout.flush()
  return rc


def main():
  args = parse_args()
  if args.workload != 'deterministic':
    print(json.dumps(secondary(args.workload, args.steps, args.warmup,
                               args.ramp_ms, args.members,
                               args.rows_per_chunk, args.spectrum_units)))
    return
  if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
    sys.exit(self_launch(args))

  import torch
  import torch.distributed as dist
  from weatherbench2_amd import _lib, engine, plan as plan_lib

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if args.gpus != world:
    raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
  if os.environ.get('WB2_BENCH_SAME_GPU'):  # smoke test of N > 1 on one GPU
    local_rank = 0
  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)
  # RCCL ("nccl") is the production backend; WB2_BENCH_DIST_BACKEND=gloo exists
  # only so the N > 1 control flow can be smoke-tested on a 1-GPU box.
  backend = os.environ.get('WB2_BENCH_DIST_BACKEND', 'nccl')
  # a process group exists whenever a launcher set the rendezvous up -- also
  # for ONE rank (python -m torch.distributed.run --nproc-per-node 1): the RCCL
  # calls of the N > 1 path then run on a 1-GPU box too
  ddp = world > 1 or ('WORLD_SIZE' in os.environ and 'MASTER_PORT' in os.environ)
  if ddp:
    import datetime
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    # rendezvous and every collective give up after --launch-timeout instead of
    # waiting forever for a rank that died
    limit = datetime.timedelta(seconds=args.launch_timeout)
    if backend == 'nccl':
      dist.init_process_group('nccl', device_id=dev, timeout=limit)
    else:
      dist.init_process_group(backend, timeout=limit)
    if dist.get_world_size() != args.gpus:
      raise SystemExit(f'--gpus {args.gpus} but the process group has '
                       f'{dist.get_world_size()} ranks')

  def all_reduce(tensor, op=None):
    op = op or dist.ReduceOp.SUM
    if backend == 'nccl':
      dist.all_reduce(tensor, op=op)
      return tensor
    host = tensor.cpu()
    dist.all_reduce(host, op=op)
    return host.to(tensor.device)

  def per_rank(value: float) -> list:
    """[value of rank 0, ..., value of rank world-1] on every rank."""
    if not ddp:
      return [value]
    mine = torch.zeros(world, dtype=torch.float64, device=dev)
    mine[rank] = value
    return [float(x) for x in all_reduce(mine).tolist()]

  lat = np.linspace(-90, 90, N_LAT)
  lon = np.linspace(0, 360, N_LON, endpoint=False)
  units, pool = args.units, max(args.pool, args.units)
  if not args.rows_per_chunk:
    args.rows_per_chunk = plan_lib.auto_rows_per_chunk(N_LAT, units * N_LEV)
  # STRONG scaling (--total-units): this rank's contiguous shard of the units,
  # `units` per step, the last step partial; K = the steps of the largest shard
  strong = args.total_units > 0
  unit_lo, n_mine = 0, None
  if strong:
    from weatherbench2_amd.evaluation import shard_bounds
    if args.total_units < world:
      raise SystemExit(f'--total-units {args.total_units} < {world} ranks')
    unit_lo, unit_hi = shard_bounds(args.total_units, world, rank)
    n_mine = unit_hi - unit_lo
    largest = shard_bounds(args.total_units, world, 0)
    args.steps = -(-(largest[1] - largest[0]) // units)

  def units_of_step(i):
    """Units the timed step i handles on this rank."""
    if not strong:
      return units
    return max(0, min(units, n_mine - i * units))
  regions = predefined_regions()
  pl = plan_lib.build_plan(lat, lon, plan_lib.LATLON, regions, dev,
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

  def tables(first_unit, n_units):
    """Slab tables of one step: forecast units are consecutive pool entries,
    truth / climatology are gathered through (different) offsets."""
    u = (first_unit + torch.arange(n_units, device=dev)) % pool
    fu = (u[:, None] * N_LEV + lev[None]).reshape(-1)
    tu = (((u + 7) % pool)[:, None] * N_LEV + lev[None]).reshape(-1)
    cu = (((u * coprime(5, pool) + 3) % pool)[:, None] * N_LEV
          + lev[None]).reshape(-1)
    return fu.contiguous(), tu.contiguous(), cu.contiguous()

  # warmup steps are always full; timed step i covers units_of_step(i)
  all_tables = [(tables(s * units, units), units) for s in range(args.warmup)]
  all_tables += [(tables(unit_lo + (args.warmup + s) * units, units_of_step(s)),
                  units_of_step(s)) for s in range(args.steps)]
  # One C-ABI call per step (wb2_det_suite_step: K1 -> K2 -> running init-time
  # mean over (metric*region, unit, level)), its arguments marshalled once per
  # table set: the host's share of a step is one foreign call.
  suites = {}
  for n_u in sorted({n for _, n in all_tables if n}):
    st = engine.SuiteStep(pl, _lib.MODE_DET_ACC, torch.float32, False,
                          n_u * N_LEV)
    st.accumulate_into(total, count, (_lib.NMETRIC * nr, n_u, N_LEV))
    suites[n_u] = st
  calls = [suites[n_u].bind([fpool, tpool, cpool], list(tabs)) if n_u else None
           for tabs, n_u in all_tables]
  k1_timer = KernelTimer()

  def step(i, timed=False):
    call = calls[i]
    if call is not None:  # strong scaling: None = this rank's shard is done
      call()

  def sampled_step(i):
    """The same step through the three separate entry points with HIP events
    around K1 alone (engine's launch hook): the kernel-time sampling loop that
    runs AFTER the timed region -- event records are barrier packets with
    timestamps and cost a step 10-18 % on some boxes, so none is inside it."""
    (fu, tu, cu), n_u = all_tables[i]
    if n_u != units:
      return
    engine.set_launch_hook(k1_timer)
    metrics, _ = engine.stream_reduce(
        pl, _lib.MODE_DET_ACC, [fpool, tpool, cpool], [fu, tu, cu], n_u * N_LEV,
        skipna=False)
    engine.set_launch_hook(None)
    engine.time_accumulate(metrics.view(_lib.NMETRIC * nr, n_u, N_LEV), 1,
                           False, total, count)

  # Touch every op of the timed region once: on a cold box the first use of a
  # torch kernel (the final division, the all-reduce) loads its code object,
  # which costs tens of ms and is not part of the hot path.
  step(0, False)
  sampled_step(0)
  k1_timer.pairs.clear()
  _ = (total / count).sum().item()
  if ddp:
    all_reduce(torch.cat([total.reshape(-1), count.reshape(-1)]))
  if args.variants_only:
    print(json.dumps(k1_variants(dev, fpool, tpool, cpool, units, pool,
                                 with_headline=True,
                                 rows=args.rows_per_chunk)))
    return
  if args.traffic_probe:
    # tools/live_traffic.py: a few launches of exactly the benched K1
    # configuration (or one variant of it) under rocprofv3 --pmc, nothing else
    if args.traffic_probe in ('deterministic', 'all'):
      for i in range(6):
        step(args.warmup + i % max(args.steps, 1), False)
      if args.traffic_probe == 'all':
        # the dominant kernels of the other legs, same process, same passes
        for workload in ('ensemble', 'spectrum', 'spectrum_materialized',
                         'spectrum_mean'):
          secondary(workload, 5, 1, 0.0, args.members, 0)
          torch.cuda.empty_cache()
    else:
      k1_variants(dev, fpool, tpool, cpool, units, pool, steps=6,
                  only=args.traffic_probe, reps=1)
    torch.cuda.synchronize()
    return

  def timed_region(step_fn, n_steps, accumulators):
    """The contract's bracket: barrier + synchronize on both sides, exactly
    n_steps steps and the path's one all-reduce inside; returns (max-over-ranks
    seconds, this rank's seconds, GPU ms between the bracketing events)."""
    for a in accumulators:
      a.zero_()
    torch.cuda.synchronize()
    if ddp:
      dist.barrier()
    torch.cuda.synchronize()
    g0 = torch.cuda.Event(enable_timing=True)
    g1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    g0.record()
    for i in range(n_steps):
      step_fn(i)
    g1.record()
    means = []
    if ddp:  # the path's only exchange: every [sum, count] pair, once
      shapes = [a.shape for a in accumulators]
      packed = all_reduce(torch.cat([a.reshape(-1) for a in accumulators]))
      accumulators, off = [], 0
      for sh in shapes:
        accumulators.append(packed[off:off + sh.numel()].reshape(sh))
        off += sh.numel()
    for s_, c_ in zip(accumulators[0::2], accumulators[1::2]):
      means.append(s_ / c_)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0
    if ddp:
      dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if ddp:
      dt = float(all_reduce(torch.tensor([dt], dtype=torch.float64, device=dev),
                            dist.ReduceOp.MAX).item())
    for m in means:
      assert torch.isfinite(m).all()
    return dt, own, g0.elapsed_time(g1)

  # From here to the timed region the GPU never runs dry: ramp and warmup are
  # enqueued back to back and the only wait is the contract's synchronize.
  ramp(lambda: step(0, False), args.ramp_ms)
  for i in range(args.warmup):
    step(i, False)
  dt, own_dt, gpu_ms = timed_region(lambda i: step(args.warmup + i, True),
                                    args.steps, [total, count])
  rank_ms = per_rank(own_dt / args.steps * 1e3)

  # ---- K1's duration: HIP events around the kernel alone, on the launch
  # stream, over the same table sets as the timed steps, in a loop of its own
  # right behind the timed region (queue kept full by a short un-evented lead-in)
  n_samp = min(args.steps, 24)
  for i in range(min(8, args.steps)):
    step(args.warmup + i)
  for i in range(n_samp):
    sampled_step(args.warmup + i)
  torch.cuda.synchronize()
  k1_ms = [a.elapsed_time(b) for a, b in k1_timer.pairs]
  if os.environ.get('WB2_BENCH_TRACE'):  # per-launch durations (diagnostics)
    print('k1_ms', ' '.join(f'{x:.3f}' for x in k1_ms), file=sys.stderr)
  k1_avg_s = (float(np.mean(k1_ms)) if k1_ms else float('nan')) / 1e3
  pts_step = units * PTS_PER_UNIT
  achieved = pts_step * BYTES_PER_PT / k1_avg_s / 1e9
  job_pts = (args.total_units * PTS_PER_UNIT if strong
             else world * pts_step * args.steps)
  out = {
      'metric': 'grid-point-evals/sec (721x1440x13)',
      'value': job_pts / dt,
      'unit': 'grid-point-evals/s',
      'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
      'ms_per_step': dt / args.steps * 1e3,
      'gpu_ms_per_step': gpu_ms / args.steps,
      'higher_is_better': True, 'scaling': 'strong' if strong else 'weak',
      'vs_baseline': None,
      'dtype': 'f32 (elementwise) + f64 (sums)', 'data': 'synthetic',
      'config': {
          'workload': ('BASELINE configs[1]: 721x1440x13 f32, deterministic '
                       'MSE+RMSE+MAE+Bias+ACC, 13 predefined slice regions, '
                       'running init-time mean'),
          'units_per_step_per_gpu': units, 'pool_units': pool,
          'total_units': args.total_units if strong else None,
          'regions': nr, 'rows_per_chunk': args.rows_per_chunk,
          'step': 'one wb2_det_suite_step call (K1 + K2 + time accumulate)',
          'parallelism': f'init-time shards x{world}, 1 all-reduce of [sum,count]',
          'launcher': ('self-spawned ranks' if os.environ.get(
              'WB2_BENCH_SELF_LAUNCHED') else
                       'torch.distributed.run' if ddp else 'single process'),
      },
      'ranks': {
          'world_size_seen': dist.get_world_size() if ddp else 1,
          'backend': ('rccl (torch "nccl")' if backend == 'nccl' else backend)
                     if ddp else None,
          'ms_per_step_per_rank': rank_ms,
          # which exchange steps of this line ran on RCCL over the GPUs' links
          'collectives': ({
              'time_mean_allreduce ([sum,count], in every timed region)':
                  'rccl' if backend == 'nccl' else f'{backend} via host copies',
              'map_allreduce': 'rccl' if backend == 'nccl' else
                               f'{backend} via host copies',
              'timing max-over-ranks / barriers':
                  'rccl' if backend == 'nccl' else backend} if ddp else None),
      },
      'roofline': {
          'bound': 'hbm', 'kernel': 'stream_partials_kernel<float,4,DET_ACC>',
          'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
          'frac': achieved / HBM_PEAK_GBPS,
          'frac_of_measured_copy_6290': achieved / 6290.0,
          'kernel_ms': k1_avg_s * 1e3,
          'kernel_ms_samples': len(k1_ms),
          'kernel_ms_from': ('HIP events around K1 alone on the launch stream, '
                             f'{len(k1_ms)} launches of the timed table sets in '
                             'a loop right behind the timed region'),
          'algorithmic_bytes_per_launch': pts_step * BYTES_PER_PT,
          # whole-step view: algorithmic bytes / the timed ms_per_step (K1 + K2
          # + accumulate + whatever the host adds) against the same peak
          'step_frac': (pts_step * BYTES_PER_PT / (dt / args.steps) / 1e9 /
                        HBM_PEAK_GBPS) if not strong else None,
          # HBM bytes per launch from PMC counters: collected below, live
          'traffic': None,
      },
  }

  # ---- the un-ramped figure: the same K steps right after an idle queue ------
  torch.cuda.synchronize()
  time.sleep(0.05)
  dt_cold, _, _ = timed_region(lambda i: step(args.warmup + i, False),
                               args.steps, [total, count])
  out['unramped'] = {
      'value': job_pts / dt_cold,
      'ms_per_step': dt_cold / args.steps * 1e3,
      'note': 'same K steps started from an idle queue (no ramp, no warmup)'}

  solo = rank == 0 and world == 1

  # ---- the WHOLE configs[1] job on one GPU: 730 init times x 4 lead times =
  # 2 920 units, 183 steps of `units` (the last one partial) through the same
  # bound calls in one timed region -- that the 20-step headline rate holds for
  # the 78 ms the job takes
  if solo and not strong:
    try:
      job_units = 2920
      job_steps = -(-job_units // units)
      sizes = [min(units, job_units - s_ * units) for s_ in range(job_steps)]
      for n_u in sorted(set(sizes) - set(suites)):
        st = engine.SuiteStep(pl, _lib.MODE_DET_ACC, torch.float32, False,
                              n_u * N_LEV)
        st.accumulate_into(total, count, (_lib.NMETRIC * nr, n_u, N_LEV))
        suites[n_u] = st
      job_tabs = [tables(s_ * units, n_u) for s_, n_u in enumerate(sizes)]
      job_calls = [suites[n_u].bind([fpool, tpool, cpool], list(tabs))
                   for tabs, n_u in zip(job_tabs, sizes)]
      ramp(lambda: step(0, False), args.ramp_ms)
      dt_job, _, gpu_job = timed_region(lambda i: job_calls[i](), job_steps,
                                        [total, count])
      out['config1_full_job'] = {
          'value': job_units * PTS_PER_UNIT / dt_job,
          'unit': 'grid-point-evals/s', 'units': job_units,
          'steps': job_steps, 'ms': dt_job * 1e3, 'gpu_ms': gpu_job,
          'vs_headline': job_units * PTS_PER_UNIT / dt_job / out['value'],
          'what': ('BASELINE configs[1] in full: 730 init x 4 lead = 2920 '
                   f'units of 13 x 721 x 1440 f32, {job_steps} suite steps of '
                   f'{units} units (the last of {sizes[-1]}), one timed '
                   'region, one running mean')}
      del job_calls, job_tabs
    except Exception as e:
      out['config1_full_job'] = {'error': f'{type(e).__name__}: {e}'}

  def leg(name, fn, *a, **k):
    """A secondary leg never costs the run its headline: errors are recorded
    under the leg's key (and show in the contract line's `errors` list)."""
    try:
      out[name] = fn(*a, **k)
    except Exception as e:
      out[name] = {'error': f'{type(e).__name__}: {e}'}
    torch.cuda.empty_cache()

  # ---- BASELINE configs[4]: deterministic + probabilistic suites, sharded ----
  if not args.no_full_suite:
    out['full_suite'] = full_suite(args, dev, pl, step, (total, count),
                                   timed_region, per_rank, world, rank)
  if ddp:
    out['map_allreduce'] = map_allreduce(dev, all_reduce, world, backend)
  if solo and not args.no_api:
    leg('api', api_leg, dev, regions, units)
    # ---- the boundary at the reference's PRODUCTION chunking: init_time=1,
    # lead_time=1 chunks of 13 variables, 16 regions, through evaluate_chunks;
    # device-resident chunks and chunks handed over as pageable NumPy arrays
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import official_chunk
    leg('api_official_chunk', official_chunk.run, dev,
        batches=(1, 16, 32, None) if args.detail else (1, None),
        host_fed='both' if args.detail else True)
    # ---- the `probabilistic` config at the chunking of ITS command lines
    # (IFS ENS, 240 x 121, 50 members, init_time=1,lead_time=1): chunk by
    # chunk and in evaluate_chunks' default windows
    import official_probabilistic
    leg('api_probabilistic', official_probabilistic.run, dev)
  if solo and not args.no_secondary:
    # ---- BASELINE configs[2] / configs[3], each with its own roofline;
    # bounded step counts keep the whole run in minutes
    legs = (('ensemble', 'ensemble', 20), ('spectrum', 'spectrum', 40),
            ('spectrum/materialized', 'spectrum_materialized', 30),
            ('spectrum/time_mean', 'spectrum_mean', 30))
    # three repetitions, the legs interleaved; the line carries the repetition
    # with the median kernel time and the min / max fraction over all three
    runs = {key: [] for key, _, _ in legs}
    for _ in range(3):
      for key, workload, n in legs:
        try:
          one = secondary(workload, n, 5, 20.0, args.members, 0)
        except Exception as e:  # never lose the GPU line to a secondary leg
          one = {'error': f'{type(e).__name__}: {e}'}
        runs[key].append(one)
        torch.cuda.empty_cache()
    for key, _, _ in legs:
      good = sorted((r for r in runs[key] if 'roofline' in r),
                    key=lambda r: r['roofline']['kernel_ms'])
      one = good[len(good) // 2] if good else runs[key][-1]
      one = {k: one[k] for k in ('value', 'unit', 'steps', 'ms_per_step',
                                 'config', 'roofline', 'error') if k in one}
      if good:
        one['roofline'].update(
            frac_min=good[-1]['roofline']['frac'],
            frac_max=good[0]['roofline']['frac'], repetitions=len(good))
      if '/' in key:
        out.setdefault('spectrum', {})[key.split('/')[1]] = one
      else:
        out[key] = one
  if solo and args.detail and not args.no_secondary:
    # ---- every other instantiation against its own roofline (--detail only:
    # minutes of GPU time that the contract run does not need)
    for name in ('spectrum_materialized_f64', 'spectrum_mean_f64'):
      leg(name, lambda name=name: {
          k: v for k, v in secondary(name, 20, 5, 20.0, args.members,
                                     0).items()
          if k in ('value', 'unit', 'ms_per_step', 'config', 'roofline',
                   'dtype')})
    leg('variants', k1_variants, dev, fpool, tpool, cpool, units, pool)
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import k3_variants
    import tier2_variants
    leg('k3_variants', k3_variants.variants, dev)
    leg('tier2_variants', tier2_variants.variants, dev)
  if solo and not args.no_pcie:
    leg('pcie_inclusive', pcie_leg, dev, pl, units, nr, total, count)
  if solo and not args.no_pmc and not strong:
    # ---- roofline.traffic, live: the PMC passes run in child processes under
    # rocprofv3 (their own 7.8 GB pools; 288 GB of HBM hold both).  The
    # contract run collects K1's; --detail every benched kernel's.
    live = live_traffic(units, pool, args.rows_per_chunk,
                        'all' if args.detail else 'deterministic')
    source = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, '
              '--kernel-trace only) around this launch configuration, '
              'collected by this run: tools/live_traffic.py')
    legs = {'deterministic': out.get('roofline'),
            'ensemble': (out.get('ensemble') or {}).get('roofline'),
            'spectrum': (out.get('spectrum') or {}).get('roofline'),
            'spectrum_materialized': ((out.get('spectrum') or {}).get(
                'materialized') or {}).get('roofline'),
            'spectrum_mean': ((out.get('spectrum') or {}).get(
                'time_mean') or {}).get('roofline')}
    for name, roof in legs.items():
      if roof is None:
        continue
      got = live.get(name) if 'error' not in live else None
      if got and got.get('traffic_bytes'):
        roof.update(traffic=got['traffic_bytes'],
                    traffic_over_algorithmic=got['traffic_bytes'] /
                    roof['algorithmic_bytes_per_launch'],
                    traffic_source=source, traffic_detail=got)
      elif name == 'deterministic' or args.detail:
        roof['traffic_live_error'] = live.get('error', 'unavailable')
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
    emit(out, args)
  if ddp:
    dist.barrier()
    dist.destroy_process_group()


DETAIL_FILE = 'bench_detail.json'
LINE_LIMIT = 4096  # bytes: the contract line must stay readable by the driver


def _pick(d, *keys):
  return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact(out: dict) -> dict:
  """The CONTRACT line: the headline with its roofline and cpu_baseline, and
  one-number summaries of the other legs.  Everything else lives in
  bench_detail.json (printed on the line before)."""
  line = _pick(out, 'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup',
               'ms_per_step', 'gpu_ms_per_step', 'higher_is_better', 'scaling',
               'vs_baseline', 'dtype', 'data')
  line['config'] = _pick(out['config'], 'workload', 'units_per_step_per_gpu',
                         'pool_units', 'total_units', 'regions',
                         'rows_per_chunk', 'step', 'parallelism', 'launcher')
  line['roofline'] = _pick(
      out['roofline'], 'bound', 'kernel', 'achieved', 'peak', 'unit', 'frac',
      'kernel_ms', 'kernel_ms_samples', 'algorithmic_bytes_per_launch',
      'step_frac', 'traffic', 'traffic_over_algorithmic', 'traffic_live_error')
  if 'cpu_baseline' in out:
    cb = out['cpu_baseline']
    line['cpu_baseline'] = _pick(cb, 'value', 'unit', 'cores', 'kind',
                                 'value_1core', 'logical_cores')
    line['cpu_baseline']['sample'] = cb.get('sample_short') or str(
        cb.get('sample', ''))[:200]
  line['unramped'] = _pick(out.get('unramped', {}), 'value', 'ms_per_step')
  if 'config1_full_job' in out:
    line['config1_full_job'] = _pick(out['config1_full_job'], 'value', 'units',
                                     'steps', 'ms', 'vs_headline', 'error')
  if out['n_gpus'] > 1 or out['ranks']['backend']:
    line['ranks'] = out['ranks']
  if 'map_allreduce' in out:
    line['map_allreduce'] = _pick(out['map_allreduce'], 'bytes_per_rank', 'ms',
                                  'busbw_GBps', 'backend')
  if 'full_suite' in out:
    line['full_suite'] = _pick(out['full_suite'], 'value', 'ms_per_step',
                               'scaling', 'steps', 'ms_per_step_per_rank',
                               'error')
    k3 = out['full_suite'].get('ensemble_kernel') or {}
    if 'frac' in k3:
      line['full_suite']['k3_frac'] = k3['frac']

  def roof_of(leg_):
    r = (leg_ or {}).get('roofline') or {}
    return _pick(r, 'frac', 'kernel_ms', 'traffic_over_algorithmic')
  if 'ensemble' in out:
    line['ensemble'] = dict(_pick(out['ensemble'], 'value', 'error'),
                            **roof_of(out['ensemble']))
  if 'spectrum' in out:
    sp = out['spectrum']
    line['spectrum'] = dict(_pick(sp, 'value', 'error'), **roof_of(sp))
    for sub in ('materialized', 'time_mean'):
      if sub in sp:
        line['spectrum'][sub + '_frac'] = roof_of(sp[sub]).get('frac')
  if 'api' in out:
    line['api'] = _pick(out['api'], 'value', 'ms_per_step', 'error')
  if 'api_official_chunk' in out:
    oc = out['api_official_chunk']
    line['api_official_chunk'] = _pick(oc, 'value', 'batch_chunks',
                                       'wall_ms_per_chunk', 'chunks', 'error')
    roof = oc.get('roofline') or {}
    if roof:
      line['api_official_chunk']['roofline'] = _pick(
          roof, 'frac', 'k1_ms_per_chunk', 'traffic_over_algorithmic')
    by = oc.get('by_batch_chunks') or {}
    if '1' in by:
      line['api_official_chunk']['chunk_by_chunk'] = dict(
          _pick(by['1'], 'value', 'host_ms_per_chunk'),
          k1_ms_per_chunk=(by['1'].get('roofline') or {}).get(
              'k1_ms_per_chunk'))
    if 'host_fed' in oc:
      line['api_official_chunk']['host_fed'] = _pick(
          oc['host_fed'], 'value', 'h2d_GBps', 'wall_ms_per_chunk', 'error')
    if 'host_fed_both_configs' in oc:
      line['api_official_chunk']['host_fed_both_configs'] = _pick(
          oc['host_fed_both_configs'], 'value', 'wall_ms_per_chunk', 'error')
    for key in ('deterministic_temporal', 'deterministic_and_temporal'):
      dt = oc.get(key) or {}
      if dt:
        line['api_official_chunk'][key] = (
            _pick(dt, 'error') if 'error' in dt else
            {k: _pick(v, 'value', 'host_ms_per_chunk') for k, v in dt.items()})
    if 'deterministic_spatial' in oc:
      ds = oc['deterministic_spatial']
      line['api_official_chunk']['deterministic_spatial'] = _pick(
          ds, 'value', 'steady_ms_per_chunk', 'chunks_per_lead_and_launch',
          'error')
      if 'roofline' in ds:
        line['api_official_chunk']['deterministic_spatial']['frac'] = (
            ds['roofline'].get('frac'))
      k1 = (ds.get('by_window') or {}).get('chunk_by_chunk') or {}
      if 'value' in k1:
        line['api_official_chunk']['deterministic_spatial'][
            'chunk_by_chunk'] = k1['value']
  if 'api_probabilistic' in out:
    ap = out['api_probabilistic']
    line['api_probabilistic'] = _pick(ap, 'value', 'ms_per_chunk', 'hbm_frac',
                                      'grid', 'error')
    if 'programs_1' in ap:
      line['api_probabilistic']['chunk_by_chunk'] = ap['programs_1']['value']
  if 'pcie_inclusive' in out:
    pc = out['pcie_inclusive']
    line['pcie_inclusive'] = {
        k: _pick(v, 'value', 'h2d_GBps') for k, v in pc.items()
        if isinstance(v, dict)} or _pick(pc, 'error')
  errors = [k for k, v in out.items() if isinstance(v, dict) and 'error' in v]
  if errors:
    line['errors'] = errors
  line['detail'] = DETAIL_FILE
  return line


def emit(out: dict, args) -> None:
  """Writes the full record to bench_detail.json (repo root, and gpurun_out/
  when that exists) and prints the contract line -- ONE JSON object of less
  than LINE_LIMIT bytes, the last (by default the only) stdout line."""
  detail = json.dumps(out)
  for d in (ROOT, os.path.join(ROOT, 'gpurun_out')):
    if os.path.isdir(d):
      try:
        with open(os.path.join(d, DETAIL_FILE), 'w') as f:
          f.write(detail + '\n')
      except OSError:
        pass
  if args.print_detail:
    print('bench_detail ' + detail)
  line = compact(out)
  text = json.dumps(line)
  if len(text) >= LINE_LIMIT:  # shed the optional summaries, never the contract
    for key in ('pcie_inclusive', 'api', 'unramped', 'api_probabilistic',
                'api_official_chunk', 'spectrum', 'ensemble', 'full_suite', 'map_allreduce'):
      line.pop(key, None)
      text = json.dumps(line)
      if len(text) < LINE_LIMIT:
        break
  sys.stdout.flush()
  print(text)
  sys.stdout.flush()


def full_suite(args, dev, pl_det, det_step, det_acc, timed_region, per_rank,
               world, rank) -> dict:
  """BASELINE configs[4]: every step pushes `units` (init, lead) units through
  the deterministic suite AND `units` 50-member ensemble units through the
  probabilistic suite (scripts/evaluate.py:496-520: crps, crps_spread,
  crps_skill, ensemble_mean_mse, debiased, variance -- one K3 pass), each rank
  on its own init-time shard; both [sum, count] pairs meet in ONE all-reduce."""
  import torch
  from weatherbench2_amd import _lib, engine, plan as plan_lib
  m = args.members
  units = args.units
  ens_pool = 3  # 3 x 13 x 50 x 4.15 MB = 8.1 GB >> Infinity Cache
  pl_ens = plan_lib.build_plan(
      np.linspace(-90, 90, N_LAT), np.linspace(0, 360, N_LON, endpoint=False),
      plan_lib.LATLON, predefined_regions(), dev,
      rows_per_chunk=plan_lib.ENSEMBLE_ROWS_PER_CHUNK)
  gen = torch.Generator(device=dev).manual_seed(4321 + rank)
  ens = torch.randn((m, ens_pool * N_LEV, N_LAT, N_LON), generator=gen,
                    device=dev)
  etruth = torch.randn((ens_pool * N_LEV, N_LAT, N_LON), generator=gen,
                       device=dev)
  stride = ens_pool * N_LEV * N_LAT * N_LON
  lev = torch.arange(N_LEV, device=dev, dtype=torch.int64)
  nr = pl_ens.n_region
  etotal = torch.zeros((_lib.NMETRIC_ENS * nr, N_LEV), dtype=torch.float64,
                       device=dev)
  ecount = torch.zeros_like(etotal)
  strong = args.total_units > 0
  n_steps = args.steps if strong else min(args.steps, 20)
  if strong:  # the same contiguous shard of units as the deterministic suite
    from weatherbench2_amd.evaluation import shard_bounds
    lo, hi = shard_bounds(args.total_units, world, rank)
  etabs = []
  for s in range(n_steps + 2):
    n_u = units
    if strong and s < n_steps:
      n_u = max(0, min(units, hi - lo - s * units))
    u = (s * units + torch.arange(n_u, device=dev)) % ens_pool
    etabs.append(((u[:, None] * N_LEV + lev[None]).reshape(-1).contiguous(),
                  (((u + 1) % ens_pool)[:, None] * N_LEV + lev[None]
                   ).reshape(-1).contiguous(), n_u))
  k3_timer = KernelTimer()

  def ens_step(i, timed=True):
    et, tt, n_u = etabs[i]
    if n_u == 0:
      return
    engine.set_launch_hook(
        k3_timer if timed and n_u == units and i % 3 == 0 else None)
    metrics, _ = engine.ensemble_reduce(pl_ens, ens, stride, m, et, etruth, tt,
                                        n_u * N_LEV, False)
    engine.set_launch_hook(None)
    engine.time_accumulate(metrics.view(_lib.NMETRIC_ENS * nr, n_u, N_LEV), 1,
                           False, etotal, ecount)

  def both(i):
    det_step(args.warmup + i, False)
    ens_step(i)

  ens_step(n_steps, False)
  ens_step(n_steps + 1, False)
  dt, own, _ = timed_region(both, n_steps, [det_acc[0], det_acc[1], etotal,
                                            ecount])
  pts_step = units * PTS_PER_UNIT
  k3_s = k3_timer.mean_ms() / 1e3
  bytes_k3 = pts_step * (m + 1) * 4.0
  return {
      'workload': ('BASELINE configs[4]: per step and per GPU '
                   f'{units} units through the deterministic suite + {units} '
                   f'{m}-member units through the probabilistic suite (K3), 13 '
                   'regions, init-time shards, one all-reduce of both '
                   '[sum,count] pairs'),
      'value': (args.total_units * PTS_PER_UNIT if strong
                else world * pts_step * n_steps) / dt,
      'unit': 'grid-point-evals/s',
      'scaling': 'strong' if strong else 'weak',
      'steps': n_steps, 'ms_per_step': dt / n_steps * 1e3,
      'ms_per_step_per_rank': per_rank(own / n_steps * 1e3),
      'ensemble_kernel': {
          'kernel': f'ens_partials_kernel<float,64,{m if m == 50 else 0}>',
          'kernel_ms': k3_s * 1e3, 'achieved': bytes_k3 / k3_s / 1e9,
          'frac': bytes_k3 / k3_s / 1e9 / HBM_PEAK_GBPS, 'unit': 'GB/s',
          'algorithmic_bytes_per_launch': bytes_k3},
  }


def map_allreduce(dev, all_reduce, world, backend) -> dict:
  """the only bandwidth-relevant exchange of the path, the
  all-reduce of the (sum, count) accumulators of Spatial* maps: one variable x
  4 leads x 13 levels x 721 x 1440 fp64 x 2 = 864 MB per rank."""
  import torch
  import torch.distributed as dist
  n = 2 * 4 * N_LEV * N_LAT * N_LON
  maps = torch.ones((n,), dtype=torch.float64, device=dev)
  all_reduce(maps)  # warm-up (communicator set-up, code objects)
  torch.cuda.synchronize()
  dist.barrier()
  reps = 3
  t0 = time.perf_counter()
  for _ in range(reps):
    maps = all_reduce(maps)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / reps
  nbytes = n * 8
  return {'bytes_per_rank': nbytes, 'ms': dt * 1e3,
          'algbw_GBps': nbytes / dt / 1e9,
          'busbw_GBps': nbytes / dt / 1e9 * 2 * (world - 1) / world,
          'backend': backend,
          'what': 'all-reduce of SpatialMSE/MAE/Bias-style (sum, count) maps: '
                  '4 leads x 13 levels x 721 x 1440 fp64 x 2'}


def api_leg(dev, regions, units) -> dict:
  """What a caller of the DROP-IN API gets: a device-resident chunk of `units`
  (init, lead) units pushed through evaluation._metric_and_region_loop
  (evaluation.py:388-438 signature; 5 metrics x 13 regions) per call, fresh
  Dataset objects every call (no cross-call result reuse)."""
  import torch
  from weatherbench2_amd import config, evaluation, metrics as gm
  from weatherbench2_amd import xarray_lite as xl
  n_lead = 4
  n_time = max(1, units // n_lead)
  lat = np.linspace(-90, 90, N_LAT)
  lon = np.linspace(0, 360, N_LON, endpoint=False)
  times = np.datetime64('2020-01-01T00') + np.arange(n_time) * np.timedelta64(
      12, 'h')
  leads = np.arange(n_lead) * np.timedelta64(6, 'h')
  g = torch.Generator(device=dev).manual_seed(0)
  dims = ('time', 'prediction_timedelta', 'level', 'latitude', 'longitude')
  coords = {'time': times.astype('datetime64[ns]'),
            'prediction_timedelta': leads.astype('timedelta64[ns]'),
            'level': np.arange(N_LEV), 'latitude': lat, 'longitude': lon}
  shape = (n_time, n_lead, N_LEV, N_LAT, N_LON)
  n_var = 4  # distinct forecast chunks >> Infinity Cache
  fs = [torch.randn(shape, device=dev, generator=g) for _ in range(n_var)]
  t = torch.randn(shape, device=dev, generator=g)
  truth = xl.Dataset({'z': xl.DataArray(t, dims)}, coords)
  clim = xl.Dataset(
      {'z': xl.DataArray(torch.randn((4, 3, N_LEV, N_LAT, N_LON), device=dev,
                                     generator=g),
                         ('hour', 'dayofyear', 'level', 'latitude',
                          'longitude'))},
      {'hour': np.array([0, 6, 12, 18]), 'dayofyear': np.array([1, 2, 3]),
       'level': np.arange(N_LEV), 'latitude': lat, 'longitude': lon})
  cfg = config.Eval(metrics={'mse': gm.MSE(), 'acc': gm.ACC(climatology=clim),
                             'bias': gm.Bias(), 'mae': gm.MAE(),
                             'rmse': gm.RMSESqrtBeforeTimeAvg()},
                    regions=regions)
  mean = evaluation.RunningMean('time', False, dev)

  def call(i):
    # a new chunk every call: fresh Dataset objects AND no result reuse (the
    # n_var forecast tensors come round again; results are cached per chunk)
    gm.clear_caches()
    f = xl.Dataset({'z': xl.DataArray(fs[i % n_var], dims)}, coords)
    mean.add(evaluation._metric_and_region_loop(f, truth, cfg, False,
                                                compute_chunk=True))

  for i in range(6):
    call(i)
  torch.cuda.synchronize()
  reps = 40
  t0 = time.perf_counter()
  for i in range(reps):
    call(i)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / reps
  pts = n_time * n_lead * PTS_PER_UNIT
  return {'value': pts / dt, 'unit': 'grid-point-evals/s',
          'ms_per_step': dt * 1e3, 'units_per_call': n_time * n_lead,
          'what': ('evaluation._metric_and_region_loop(forecast, truth, Eval('
                   '5 metrics, 13 regions), compute_chunk=True) + RunningMean.add '
                   'on a device-resident chunk, fresh Datasets per call')}


def pcie_leg(dev, pl, units, nr, total, count) -> dict:
  """The same step when the boundary hands over HOST buffers, through the
  pipelined feeder (pinned ring + copy stream: the transfer of chunk i + 1
  overlaps the pass over chunk i): (i) forecast, truth and climatology all
  cross PCIe; (ii) only the forecast does, truth / climatology stay resident in
  HBM and are gathered through slab tables.  Reported beside `value`."""
  import torch
  from weatherbench2_amd import _lib, engine, feeder
  n_outer = units * N_LEV
  shape = (n_outer, N_LAT, N_LON)
  n_el = n_outer * N_LAT * N_LON
  host = [torch.empty((n_el,), dtype=torch.float32).pin_memory()
          for _ in range(3)]
  for h in host:
    h.normal_()
  resident = [torch.randn(shape, device=dev) for _ in range(2)]
  pts_step = units * PTS_PER_UNIT
  result = {}
  for name, n_streams in (('all_inputs_over_pcie', 3),
                          ('forecast_over_pcie_truth_clim_resident', 1)):
    feeders = [feeder.ChunkFeeder(shape, torch.float32, dev, depth=2)
               for _ in range(n_streams)]
    reps = 6
    for f_, h in zip(feeders, host):
      f_.submit(h)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(reps):
      xs = []
      for f_, h in zip(feeders, host):
        # chunk i + 1 starts to move ...  (also behind the last step: every
        # timed step then holds exactly one transfer per stream -- rounds 1-5
        # left the last one out and still divided the bytes of `reps`
        # transfers by the time of reps - 1: h2d_GBps was 6/5 of the truth)
        f_.submit(h)
      for f_ in feeders:
        xs.append(f_.acquire())
      xs += resident[:3 - n_streams]
      m, _ = engine.stream_reduce(pl, _lib.MODE_DET_ACC, xs,
                                  [None, None, None], n_outer, skipna=False)
      engine.time_accumulate(m.view(_lib.NMETRIC * nr, units, N_LEV), 1, False,
                             total, count)
      for f_ in feeders:
        f_.release()            # ... while this pass runs
    torch.cuda.synchronize()
    dt_h = (time.perf_counter() - t1) / reps
    result[name] = {
        'value': pts_step / dt_h, 'unit': 'grid-point-evals/s',
        'ms_per_step': dt_h * 1e3,
        'h2d_GBps': pts_step * 4.0 * n_streams / dt_h / 1e9}
    del feeders
  result['note'] = ('inputs start in pinned host memory; double-buffered '
                    'ChunkFeeder on a copy stream (weatherbench2_amd/feeder.py)')
  return result


def secondary(workload_name, steps, warmup, ramp_ms, members=50,
              rows_per_chunk=0, spectrum_units=SPECTRUM_UNITS) -> dict:
  """BASELINE configs[2] (50-member ensemble) and configs[3] (zonal spectrum)
  on one GPU: same timing discipline, their own roofline.  Returns the JSON
  line of `--workload NAME`; the default run embeds the same dicts."""
  import types
  import torch
  from weatherbench2_amd import _lib, engine, plan as plan_lib
  args = types.SimpleNamespace(workload=workload_name, steps=steps,
                               warmup=warmup, ramp_ms=ramp_ms, members=members,
                               rows_per_chunk=rows_per_chunk,
                               spectrum_units=spectrum_units)
  dev = torch.device('cuda', torch.cuda.current_device())
  lat = np.linspace(-90, 90, N_LAT)
  lon = np.linspace(0, 360, N_LON, endpoint=False)
  gen = torch.Generator(device=dev).manual_seed(99)
  events = []
  timer = KernelTimer()
  f64 = False
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
      engine.set_launch_hook(
          timer if timed and (i - args.warmup) % 3 == 0 else None)
      tab = tabs[i % pool]
      engine.ensemble_reduce(pl, ens, stride, m, tab, truth, tab, n_slab, False)
    kernel = f'ens_partials_kernel<float,64,{m if m == 50 else 0}>'
    workload = (f'BASELINE configs[2]: 721x1440x13 f32, {m}-member CRPS + '
                'spread/skill + ensemble-mean MSE + variance + debiased MSE, '
                '13 regions')
  else:
    # `_f64`: float64 rows (complex128 transform, fft_core.hpp templated on
    # the scalar type): 8 B/pt read
    f64 = args.workload.endswith('_f64')
    if f64:
      args.workload = args.workload[:-4]
    units = args.spectrum_units
    pool = max(2, 48 // units)   # 2.6 GB of distinct input >> Infinity Cache
    x = torch.randn((pool * units, N_LEV, N_LAT, N_LON), generator=gen,
                    device=dev, dtype=torch.float64 if f64 else torch.float32)
    from weatherbench2_amd.derived_variables import ZonalEnergySpectrum
    circ = torch.as_tensor(ZonalEnergySpectrum._circumference(lat)).to(dev)
    w_host = plan_lib.get_lat_weights(lat)
    w_lat = torch.as_tensor(w_host).to(dev)
    w_sum = float(np.sum(w_host))
    w_row = (w_lat * circ).contiguous()
    pts = units * PTS_PER_UNIT
    n_bins = N_LON // 2 + 1
    in_bytes = 8.0 if f64 else 4.0
    bytes_per_pt = in_bytes + n_bins * 8.0 / N_LON
    if args.workload == 'spectrum_mean':
      bytes_per_pt = in_bytes + n_bins * 8.0 / N_LON / units
    if args.workload == 'spectrum':
      # fused latitude mean, SURVEY 8d strictly: 4 B/pt read + the ONE reduced
      # 721-bin float64 spectrum per field.  The per-segment partial spectra
      # the two-step reduction writes and re-reads are scratch: they show up
      # in `traffic` (1.04 x), not here
      bytes_per_pt = in_bytes + n_bins * 8.0 / (N_LAT * N_LON)

    def step(i, timed):
      xs = x[(i % pool) * units:(i % pool + 1) * units]
      # events on every third timed step (see step() of the headline)
      timed = timed and (i - args.warmup) % 3 == 0
      if timed:
        ev = (torch.cuda.Event(enable_timing=True),
              torch.cuda.Event(enable_timing=True))
        ev[0].record()
      if args.workload == 'spectrum_mean':
        # the script's pipeline (compute_zonal_energy_spectrum.py:234): the
        # time mean of the spectrum, fused -- the units act as times
        engine.zonal_spectrum(xs, circ, N_LAT, n_time=units)
        if timed:
          ev[1].record()
          events.append(ev)
        return
      if args.workload == 'spectrum':
        # configs[3]: spectrum + area-weighted latitude mean, per-latitude
        # spectra never written: [units, 13, 721, 1440] -> [units, 13, 721 bins]
        engine.zonal_spectrum_lat_mean(xs, circ, w_lat, N_LAT,
                                       weight_sum=w_sum, row_weight=w_row)
        if timed:
          ev[1].record()
          events.append(ev)
        return
      spec = engine.zonal_spectrum(xs, circ, N_LAT)
      if timed:
        ev[1].record()
        events.append(ev)
      # the same latitude mean from the materialised spectrum (K7):
      # [units * 13, 721 lat, 721 bins] -> [units * 13, 721]
      total, _, count = engine.axis_moments(
          spec.reshape(units * N_LEV, N_LAT, n_bins), units * N_LEV,
          N_LAT, n_bins, w_lat, False)
      lat_mean = total / count
    kernel = {
        'spectrum_mean': 'fused_spectrum_kernel<720,TIME_MEAN> (LDS real FFT, '
                         'time mean in registers)',
        'spectrum': 'fused_spectrum_kernel<720,LATSEG> + latseg_combine_kernel '
                    '(LDS real FFT, weighted latitude sums in registers; the '
                    'event pair brackets both)',
        'spectrum_materialized': 'fused_spectrum_kernel<720,MATERIALISE> (LDS '
                                 'real FFT + power epilogue); '
                                 'WB2HIP_SPECTRUM_BACKEND=rocfft selects rocFFT '
                                 'C2C + power_kernel'}[args.workload]
    workload = {
        'spectrum_mean': 'the time-mean pipeline of scripts/compute_zonal_energy_'
                         f'spectrum.py:234 on {units} units of 13x721x1440 f32 per '
                         'step: spectrum and its mean over the units in ONE '
                         'kernel (4 B/pt read, one 721-bin spectrum per row '
                         'written)',
        'spectrum': f'BASELINE configs[3]: zonal energy spectrum of {units} units of '
                    '13x721x1440 f32 per step + its area-weighted latitude '
                    'mean, fused (per-latitude spectra never written)',
        'spectrum_materialized': f'ZonalEnergySpectrum.compute on {units} units of '
                                 '13x721x1440 f32 per step (per-latitude spectra '
                                 'materialised, 8 B/pt), then their area-weighted '
                                 'latitude mean (K7; the roofline entry is the '
                                 'spectrum kernel alone)'}[args.workload]
  if f64:
    kernel = kernel.replace('<720,', '<720 complex128 points,')
    workload = workload.replace(' f32', ' f64 (complex128 transform)')
  ramp(lambda: step(0, False), args.ramp_ms)
  for i in range(args.warmup):
    step(i, False)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for i in range(args.steps):
    step(args.warmup + i, True)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  engine.set_launch_hook(None)
  pairs = events or timer.pairs
  k_s = float(np.mean([a.elapsed_time(b) for a, b in pairs])) / 1e3
  achieved = pts * bytes_per_pt / k_s / 1e9
  return {
      'metric': 'grid-point-evals/sec (721x1440x13)',
      'value': pts * args.steps / dt, 'unit': 'grid-point-evals/s',
      'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
      'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
      'scaling': 'weak', 'vs_baseline': None,
      'dtype': 'f64' if f64 else 'f32', 'data': 'synthetic',
      'config': {'workload': workload},
      'roofline': {'bound': 'hbm', 'kernel': kernel, 'achieved': achieved,
                   'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                   'frac': achieved / HBM_PEAK_GBPS, 'kernel_ms': k_s * 1e3,
                   'algorithmic_bytes_per_launch': pts * bytes_per_pt,
                   # filled in by the default line from PMC counters
                   # collected in the same run (tools/live_traffic.py)
                   'traffic': None}}


def land_sea_mask(rs, lat, lon):
  """A synthetic land-sea mask in [0, 1] (smooth blobs, ~30 % land, fractional
  coast cells, float32 values like ERA5's `land_sea_mask`)."""
  yy, xx = np.meshgrid(np.deg2rad(lat), np.deg2rad(lon), indexing='ij')
  field = np.zeros_like(yy)
  for _ in range(24):
    a, b, c, d = rs.uniform(-1, 1), rs.randint(1, 5), rs.randint(1, 4), (
        rs.uniform(0, 2 * np.pi))
    field += a * np.cos(b * xx + d) * np.cos(c * yy + d)
  return np.clip((field - np.quantile(field, 0.6)) * 2.0, 0.0, 1.0).astype(
      np.float32)


def official_regions(n_lat: int = 0, n_lon: int = 0):
  """The 16 regions of the reference's `deterministic` config with a land-sea
  mask present (scripts/evaluate.py:345-395): the 13 slice regions +
  global_land, extra-tropics_land, tropics_land."""
  from weatherbench2_amd import regions as R
  from weatherbench2_amd import xarray_lite as xl
  lat = np.linspace(-90, 90, n_lat or N_LAT)
  lon = np.linspace(0, 360, n_lon or N_LON, endpoint=False)
  lsm = xl.DataArray(land_sea_mask(np.random.RandomState(7), lat, lon),
                     ('latitude', 'longitude'),
                     {'latitude': lat, 'longitude': lon})
  regions = predefined_regions()
  regions['global_land'] = R.LandRegion(land_sea_mask=lsm)
  regions['extra-tropics_land'] = R.CombinedRegion(regions=[
      R.SliceRegion(lat_slice=[slice(None, -20), slice(20, None)]),
      R.LandRegion(land_sea_mask=lsm)])
  regions['tropics_land'] = R.CombinedRegion(regions=[
      R.SliceRegion(lat_slice=slice(-20, 20)), R.LandRegion(land_sea_mask=lsm)])
  return regions


def k1_variants(dev, fpool, tpool, cpool, units, pool, steps=30,
                only=None, with_headline=False, rows=0, reps=3) -> dict:
  """Kernel time + fraction of the HBM peak of K1's OTHER production
  instantiations (the headline is MODE_DET_ACC / float32 / 13 slice regions /
  no skipna), same launch size (16 units of 13 x 721 x 1440), same pools:
    official16_landmask  the 16 regions of the official `deterministic` config
                         (scripts/evaluate.py:345-395): three of them carry the
                         2-D land-sea mask -> the WF = true instantiation (a
                         second set of fp64 accumulators + the mask field)
    skipna               notnull-weighted extra slots (K 6 -> 10)
    f64_inputs           float64 forecast / truth / climatology (24 B/pt)
    wind                 MODE_WIND: u, v of forecast and truth (16 B/pt)
    det_no_acc           MODE_DET: no climatology (8 B/pt)
    lonlat               (..., longitude, latitude) slabs: rows = longitude,
                         the latitude weights applied per column
  Algorithmic bytes exclude the mask / weight tables (cache-resident)."""
  import torch
  from weatherbench2_amd import _lib, engine, plan as plan_lib
  lat = np.linspace(-90, 90, N_LAT)
  lon = np.linspace(0, 360, N_LON, endpoint=False)
  lev = torch.arange(N_LEV, device=dev, dtype=torch.int64)
  n_outer = units * N_LEV
  rows = rows or plan_lib.auto_rows_per_chunk(N_LAT, n_outer)

  def tabs(step, pool_units, k):
    u = (step * units + torch.arange(units, device=dev)) % pool_units
    return [(((u * coprime(2 * j + 1, pool_units) + 3 * j) % pool_units)[:, None] * N_LEV
             + lev[None]).reshape(-1).contiguous() for j in range(k)]

  def run(pl, mode, inputs, pool_units, skipna):
    timer = KernelTimer()
    k = len(inputs)
    tables = [tabs(s_, pool_units, k) for s_ in range(steps + 3)]
    # warm-up: code object, then the same untimed clock ramp as every other leg
    # (this leg follows host-heavy ones: api, PCIe)
    for i in range(3):
      engine.stream_reduce(pl, mode, inputs, tables[i], n_outer, skipna)
    ramp(lambda: engine.stream_reduce(pl, mode, inputs, tables[0], n_outer,
                                      skipna), 20.0)
    engine.set_launch_hook(timer)
    for i in range(steps):
      engine.stream_reduce(pl, mode, inputs, tables[3 + i], n_outer, skipna)
    engine.set_launch_hook(None)
    torch.cuda.synchronize()
    return timer.mean_ms()

  pl13 = plan_lib.build_plan(lat, lon, plan_lib.LATLON, predefined_regions(),
                             dev, rows_per_chunk=rows)
  pl16 = plan_lib.build_plan(lat, lon, plan_lib.LATLON, official_regions(), dev,
                             rows_per_chunk=rows)
  rows_ll = plan_lib.auto_rows_per_chunk(N_LON, n_outer)
  pl_ll = plan_lib.build_plan(lat, lon, plan_lib.LONLAT, predefined_regions(),
                              dev, rows_per_chunk=rows_ll)
  f32 = [fpool, tpool, cpool]
  # lon-lat layout: the same bytes viewed as (slab, longitude, latitude)
  ll = [x.view(-1, N_LON, N_LAT) for x in f32]
  # the fourth input of the wind mode: a pool of its own (reusing one of the
  # three would let a launch read some slabs twice: cache hits, not bandwidth)
  wpool = torch.randn_like(fpool) if only in (None, 'wind') else None
  f64 = None
  pool64 = units + 8
  if only in (None, 'f64_inputs'):
    gen = torch.Generator(device=dev).manual_seed(77)
    f64 = [torch.randn((pool64 * N_LEV, N_LAT, N_LON), generator=gen,
                       device=dev, dtype=torch.float64) for _ in range(3)]
  # (name, plan, mode, inputs, pool units, skipna, B/pt, what)
  specs = [
      ('headline', pl13, _lib.MODE_DET_ACC, f32, pool, False, 12.0,
       'MODE_DET_ACC f32, 13 regions (the benched one)'),
      ('official16_landmask', pl16, _lib.MODE_DET_ACC, f32, pool, False, 12.0,
       'MODE_DET_ACC f32, the 16 regions of scripts/evaluate.py:345-395 incl. '
       'global_land / extra-tropics_land / tropics_land (2-D mask: WF = true)'),
      ('skipna', pl13, _lib.MODE_DET_ACC, f32, pool, True, 12.0,
       'MODE_DET_ACC f32, 13 regions, skipna = True (K = 10 slots)'),
      ('det_no_acc', pl13, _lib.MODE_DET, f32[:2], pool, False, 8.0,
       'MODE_DET f32 (MSE / RMSE / MAE / Bias without a climatology), 13 '
       'regions'),
      ('wind', pl13, _lib.MODE_WIND, [fpool, tpool, cpool, wpool], pool, False,
       16.0, 'MODE_WIND f32: u, v of forecast and truth (4 inputs), 13 regions'),
      ('lonlat', pl_ll, _lib.MODE_DET_ACC, ll, pool, False, 12.0,
       'MODE_DET_ACC f32 on (..., longitude, latitude) slabs (721 columns: '
       'rows are not 16-byte aligned), 13 regions'),
      ('f64_inputs', pl13, _lib.MODE_DET_ACC, f64, pool64, False, 24.0,
       'MODE_DET_ACC float64 inputs (24 B/pt), 13 regions'),
  ]
  specs = [sp for sp in specs if sp[3] is not None and (
      sp[0] == only if only is not None
      else (sp[0] != 'headline' or with_headline))]
  # `reps` repetitions, the variants interleaved: one sample per run is not a
  # measurement (the register-heavy instantiations spread by 10 % between
  # boxes and moments); the line carries the median with min / max
  samples = {sp[0]: [] for sp in specs}
  for _ in range(reps):
    for name, pl, mode, inputs, pool_units, skipna, _, _ in specs:
      samples[name].append(run(pl, mode, inputs, pool_units, skipna))
  out = {}
  for name, pl, mode, inputs, pool_units, skipna, bytes_per_pt, what in specs:
    ms = sorted(samples[name])
    med = ms[len(ms) // 2]
    nbytes = units * PTS_PER_UNIT * bytes_per_pt
    frac = lambda t: nbytes / t / 1e6 / HBM_PEAK_GBPS
    out[name] = {'what': what, 'kernel_ms': med,
                 'algorithmic_bytes_per_launch': nbytes,
                 'achieved': nbytes / med / 1e6, 'unit': 'GB/s',
                 'frac': frac(med), 'frac_min': frac(ms[-1]),
                 'frac_max': frac(ms[0]), 'repetitions': len(ms),
                 'regions': pl.n_region,
                 'rows_per_chunk': rows_ll if name == 'lonlat' else rows,
                 'weight_field': pl.wfield is not None}
  del f64, wpool
  return out


def live_traffic(units, pool, rows_per_chunk, workload='all') -> dict:
  """HBM bytes per launch of the benched kernels (K1, K3, the three modes of the
  fused spectrum kernel) from PMC counters, collected NOW (tools/
  live_traffic.py: two rocprofv3 passes around `bench.py --traffic-probe all`):
  {workload: {traffic_bytes, algorithmic_bytes, ratio, ...}}."""
  tool = os.path.join(ROOT, 'tools', 'live_traffic.py')
  try:
    res = subprocess.run(
        [sys.executable, tool, '--workload', workload, '--units', str(units),
         '--pool', str(pool), '--rows-per-chunk', str(rows_per_chunk)],
        cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
        timeout=500)
    if res.returncode != 0:
      return {'error': (res.stderr or res.stdout).strip()[-300:]}
    got = json.loads(res.stdout.strip().splitlines()[-1])
    # `--workload deterministic` prints K1's object flat
    return {'deterministic': got} if workload == 'deterministic' else got
  except Exception as e:  # rocprofv3 missing, timeout, ...
    return {'error': f'{type(e).__name__}: {e}'}


if __name__ == '__main__':
  main()
