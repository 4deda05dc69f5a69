"""The drop-in boundary at the reference's PRODUCTION chunking, measured.

  python tools/official_chunk.py [--chunks 128] [--batch 1,16,32,default] [--profile]

The official 0.25-degree deterministic run (docs/source/official-evaluation.md:
537-556) feeds `_evaluate_chunk` (evaluation.py:583-599) one chunk per
`--input_chunks=init_time=1,lead_time=1` with all 13 variables in it
(`split_vars=False`, evaluation.py:693-705): six 13-level fields (geopotential,
temperature, u / v wind, specific humidity, wind speed) and seven surface
fields = 85 slabs of 721 x 1440 per chunk, against the 16 regions of
scripts/evaluate.py:345-395 (13 slices + 3 land-sea-mask regions) with the
metrics of its `deterministic` config (:420-425): mse (+ the two wind-vector
pairs, :279-311), acc, bias, mae.

This leg streams such chunks -- device-resident, float32, synthetic N(0, 1) --
through `evaluation.evaluate_chunks(..., batch_chunks=k)` and reports, per k:
grid-point-evals/s (one eval = one point of one variable through the whole
metric set), wall / host time per chunk, K1 launches per chunk and the K1
roofline fraction at that launch size (HIP events on the launch stream).
`bench.py` puts the result into its default line as `api_official_chunk`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

N_LEV, N_LAT, N_LON = 13, 721, 1440
# WB2HIP_OFFICIAL_GRID=240x121: the same legs on the 1.5-degree grid (what most
# WeatherBench 2 scorecards are computed on)
if os.environ.get('WB2HIP_OFFICIAL_GRID'):
  N_LON, N_LAT = (int(v) for v in os.environ['WB2HIP_OFFICIAL_GRID'].split('x'))
VARS_3D = ('geopotential', 'temperature', 'u_component_of_wind',
           'v_component_of_wind', 'specific_humidity', 'wind_speed')
VARS_2D = ('2m_temperature', '10m_u_component_of_wind',
           '10m_v_component_of_wind', 'mean_sea_level_pressure',
           'total_precipitation_6hr', 'total_precipitation_24hr',
           '10m_wind_speed')
WIND = (('u_component_of_wind', 'v_component_of_wind', 'wind_vector'),
        ('10m_u_component_of_wind', '10m_v_component_of_wind',
         '10m_wind_vector'))
SLABS_PER_CHUNK = len(VARS_3D) * N_LEV + len(VARS_2D)           # 85
WIND_SLABS_PER_CHUNK = N_LEV + 1                                # 14 (u, v) pairs
SEEPS = (('seeps_24hr', 'total_precipitation_24hr', 0.25),
         ('seeps_6hr', 'total_precipitation_6hr', 0.1))
PTS_PER_CHUNK = SLABS_PER_CHUNK * N_LAT * N_LON
# ALGORITHMIC bytes, strictly (every metric shares one read of its
# inputs): 12 B per point of every variable (forecast + truth + climatology)
# + the two SEEPS wet-threshold fields (4 B per point each).  What the passes
# read on top of that is re-read traffic, not algorithm: the wind-vector pass
# reads u, v of forecast and truth again (16 B per point of 14 pairs), the two
# SEEPS passes their forecast / truth precipitation (8 B per point each).
DET_BYTES_PER_CHUNK = PTS_PER_CHUNK * 12.0
SEEPS_BYTES_PER_CHUNK = len(SEEPS) * N_LAT * N_LON * 4.0
WIND_BYTES_PER_CHUNK = WIND_SLABS_PER_CHUNK * N_LAT * N_LON * 16.0
SEEPS_REREAD_PER_CHUNK = len(SEEPS) * N_LAT * N_LON * 8.0
HBM_PEAK_GBPS = 8000.0
CLIM_DAYS = 128


class _Events:
  """HIP events around every K1 launch (engine's launch hook) + a count."""

  def __init__(self, timed: bool, kernel: str = 'stream_partials'):
    self.timed = timed
    self.kernel = kernel
    self.pairs: list = []
    self.launches = 0
    self.cur = None

  def __call__(self, when, kernel):
    import torch
    if kernel != self.kernel:
      return
    if when == 'begin':
      self.launches += 1
      if self.timed:
        self.cur = (torch.cuda.Event(enable_timing=True),
                    torch.cuda.Event(enable_timing=True))
        self.cur[0].record()
    elif self.timed:
      self.cur[1].record()
      self.pairs.append(self.cur)


def build(dev, n_chunks: int, pool: int, n_lead: int = 4, seeps: bool = True):
  """(chunks, eval config): `n_chunks` (init_time=1, lead_time=1) chunk pairs
  in init-major order over a pool of `pool` distinct device-resident chunks;
  `seeps`: the two SEEPS metrics of `--compute_seeps=True`
  (scripts/evaluate.py:436-445) with their climatological wet thresholds and
  dry fractions."""
  import torch
  import bench
  from weatherbench2_amd import config, metrics as gm
  from weatherbench2_amd import xarray_lite as xl
  lat = np.linspace(-90, 90, N_LAT)
  lon = np.linspace(0, 360, N_LON, endpoint=False)
  level = np.array([50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925,
                    1000])
  n_init = -(-n_chunks // n_lead)
  # daily inits x 6-hourly leads below one day: every chunk of the list has a
  # valid time of its own, so no window reads a climatology slab twice (a
  # launch that does measures cache hits, not bandwidth)
  assert n_lead <= 4
  # (the climatology holds CLIM_DAYS days of the year -- 4 hours x 85 slabs x
  # 4.15 MB a day: 182 GB for 128 --; longer lists go on with the same days of
  # the NEXT year: init times stay distinct, a climatology slab comes back
  # after 4 x CLIM_DAYS chunks, long after it has left every cache)
  k = np.arange(n_init)
  years = np.array([np.datetime64(f'{2021 + y}-01-01T00', 'ns')
                    for y in range(int(k.max()) // CLIM_DAYS + 1)])
  init = years[k // CLIM_DAYS] + (k % CLIM_DAYS) * np.timedelta64(24, 'h')
  lead = (np.arange(n_lead) * np.timedelta64(6, 'h')).astype('timedelta64[ns]')
  n_day = min(n_init, CLIM_DAYS) + 1
  g = torch.Generator(device=dev).manual_seed(11)

  def randn(*shape):
    return torch.randn(shape, device=dev, generator=g)
  d3 = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
  d2 = ('init_time', 'lead_time', 'latitude', 'longitude')
  pooled = []
  for _ in range(pool):
    f = {k: randn(1, 1, N_LEV, N_LAT, N_LON) for k in VARS_3D}
    f.update({k: randn(1, 1, N_LAT, N_LON) for k in VARS_2D})
    t = {k: randn(1, 1, N_LEV, N_LAT, N_LON) for k in VARS_3D}
    t.update({k: randn(1, 1, N_LAT, N_LON) for k in VARS_2D})
    for _, precip, _ in SEEPS:   # precipitation is not negative
      f[precip].abs_()
      t[precip].abs_()
    pooled.append((f, t))
  ccoords = {'hour': np.array([0, 6, 12, 18]),
             'dayofyear': 1 + np.arange(n_day), 'level': level,
             'latitude': lat, 'longitude': lon}
  clim = xl.Dataset(coords=ccoords)
  for k in VARS_3D:
    clim[k] = xl.DataArray(randn(4, n_day, N_LEV, N_LAT, N_LON),
                           ('hour', 'dayofyear', 'level', 'latitude',
                            'longitude'))
  for k in VARS_2D:
    clim[k] = xl.DataArray(randn(4, n_day, N_LAT, N_LON),
                           ('hour', 'dayofyear', 'latitude', 'longitude'))
  if seeps:
    for _, precip, _ in SEEPS:
      clim[f'{precip}_seeps_threshold'] = xl.DataArray(
          randn(4, n_day, N_LAT, N_LON).abs_() * 0.7 + 0.1,
          ('hour', 'dayofyear', 'latitude', 'longitude'))
      clim[f'{precip}_seeps_dry_fraction'] = xl.DataArray(
          torch.rand((4, n_day, N_LAT, N_LON), device=dev, generator=g) * 0.9
          + 0.02, ('hour', 'dayofyear', 'latitude', 'longitude'))
  chunks = []
  for j in range(n_chunks):
    i, l = divmod(j, n_lead)
    valid = init[i:i + 1, None] + lead[None, l:l + 1]
    coords = {'init_time': init[i:i + 1], 'lead_time': lead[l:l + 1],
              'level': level, 'latitude': lat, 'longitude': lon,
              'valid_time': xl.DataArray(valid, ('init_time', 'lead_time'))}
    f, t = pooled[j % pool]
    fd = xl.Dataset({k: xl.DataArray(v, d3 if v.dim() == 5 else d2)
                     for k, v in f.items()}, coords)
    td = xl.Dataset({k: xl.DataArray(v, d3 if v.dim() == 5 else d2)
                     for k, v in t.items()}, dict(coords))
    chunks.append((fd, td))
  wv = [gm.WindVectorMSE(u_name=u, v_name=v, vector_name=n)
        for u, v, n in WIND]
  metrics = {'mse': gm.MSE(wind_vector_mse=wv),
             'acc': gm.ACC(climatology=clim), 'bias': gm.Bias(),
             'mae': gm.MAE()}
  if seeps:
    for key, precip, dry in SEEPS:
      metrics[key] = gm.SEEPS(climatology=clim, precip_name=precip,
                              dry_threshold_mm=dry)
  cfg = config.Eval(metrics=metrics, regions=bench.official_regions(N_LAT, N_LON))
  return chunks, cfg


def temporal_config(cfg):
  """The `deterministic_temporal` config of the documented command line
  (scripts/evaluate.py:479-487): the deterministic metrics + RMSE with the
  square root per chunk, `temporal_mean=False` -- every chunk's values are
  kept under their (init_time, lead_time) labels."""
  import dataclasses
  from weatherbench2_amd import metrics as gm
  wv = [gm.WindVectorRMSESqrtBeforeTimeAvg(u_name=u, v_name=v, vector_name=n)
        for u, v, n in WIND]
  metrics = dict(cfg.metrics)
  metrics['rmse_sqrt_before_time_avg'] = gm.RMSESqrtBeforeTimeAvg(
      wind_vector_rmse=wv)
  return dataclasses.replace(cfg, metrics=metrics, temporal_mean=False)


def spatial_config(cfg=None):
  """The `deterministic_spatial` config of the documented command line
  (scripts/evaluate.py:431-457, 471-478): bias, mse and mae maps of every
  variable + (`--compute_seeps=True`, given `cfg` with the scalar SEEPS pair)
  the two SpatialSEEPS maps under the reference's keys; no regions, temporal
  mean."""
  from weatherbench2_amd import config, metrics as gm
  metrics = {'bias': gm.SpatialBias(), 'mse': gm.SpatialMSE(),
             'mae': gm.SpatialMAE()}
  if cfg is not None:
    for (key, precip, dry), out_key in zip(SEEPS, ('seeps_24hr', 'seels_6hr')):
      if key in cfg.metrics:
        metrics[out_key] = gm.SpatialSEEPS(
            climatology=cfg.metrics[key].climatology, precip_name=precip,
            dry_threshold_mm=dry)
  return config.Eval(metrics=metrics)


# read forecast + truth, read and write three float64 running sums
SPATIAL_BYTES_PER_POINT = 8.0 + 3 * 16.0


def measure_spatial(chunks, cfg=None, short: int = 64, long: int = 256,
                    n_lead: int = 4) -> dict:
  """`deterministic_spatial` through evaluate_chunks: in windows of 8 chunks
  per lead label (the headline: one accumulate launch adds the k = 8 time
  steps of a lead, the running sums cross HBM once per 8 chunks) and chunk by
  chunk (k = 1)."""
  legs = {'window': _measure_spatial(chunks, cfg, short, long, 8 * n_lead, 8),
          'chunk_by_chunk': _measure_spatial(chunks, cfg, short, long, 1, 1)}
  out = dict(legs['window'])
  out['by_window'] = legs
  return out


def _measure_spatial(chunks, cfg, short, long, batch, k) -> dict:
  """One window size.  The first chunk of a structure takes the generic path
  and result() brings 8.5 GB of mean maps to the host (through the pinned
  ring: `result`) -- one-offs that a production run spreads over ~10^4
  chunks: reported are the fused kernel
  (map_suite.py) under HIP events, the host time per chunk from two list
  lengths (up to the moment result() is called), and the walls."""
  import torch
  from weatherbench2_amd import engine, evaluation
  cfg = spatial_config(cfg)
  # (4 lead rows of float64 sums + counts of 85 slabs x 3 maps: 17 GB; inside
  # bench.py the other legs' cached blocks go first)
  torch.cuda.empty_cache()
  evaluation.evaluate_chunks(chunks[:4], cfg, False, prefetch=0, batch_chunks=1)
  walls, hosts = {}, {}
  marks = {}
  real_result = evaluation.RunningMean.result
  real_window = evaluation._evaluate_map_window

  def result(self):
    marks['enqueued'] = time.perf_counter()
    return real_result(self)

  def window(pieces, *a, **k):   # (start of every window, chunks before it)
    marks['windows'].append((time.perf_counter(), marks['chunks']))
    marks['chunks'] += len(pieces)
    return real_window(pieces, *a, **k)
  ev = None
  for n in (short, long):
    ev = _Events(True, 'spatial_accumulate')
    old = engine.set_launch_hook(ev)
    evaluation.RunningMean.result = result
    evaluation._evaluate_map_window = window
    marks['windows'], marks['chunks'] = [], 0
    try:
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      out = evaluation.evaluate_chunks(chunks[:n], cfg, False, prefetch=0,
                                       batch_chunks=batch)
      torch.cuda.synchronize()
      walls[n] = time.perf_counter() - t0
      hosts[n] = marks['enqueued'] - t0
      # result(): the queue drains, then the mean maps leave for the host
      # through the pinned ring (feeder.download)
      marks['result_s'] = time.perf_counter() - marks['enqueued']
      marks['result_bytes'] = sum(
          v.data.nbytes for v in out.data_vars.values())
    finally:
      evaluation.RunningMean.result = real_result
      evaluation._evaluate_map_window = real_window
      engine.set_launch_hook(old)
    del out
  torch.cuda.empty_cache()
  # host time per chunk of the LONG run from its second window on (the first
  # one holds the generic first chunk and the build of the suite: one-offs
  # whose run-to-run spread used to dominate a difference of two runs)
  after = [w for w in marks['windows'] if w[1] > 0]
  if after and long > after[0][1]:
    host_ms = (marks['enqueued'] - after[0][0]) / (long - after[0][1]) * 1e3
  else:
    host_ms = (hosts[long] - hosts[short]) / (long - short) * 1e3
  ms = [a.elapsed_time(b) for a, b in ev.pairs]
  # (the first chunk of the list goes through the generic path: the launches
  # cover the other long - 1)
  kernel_ms = sum(ms) / max(long - 1, 1)
  # per chunk and grid point: 8 B of forecast + truth; the three float64
  # running sums read and written once per k chunks
  per_point = 8.0 + 48.0 / k
  bytes_per_chunk = PTS_PER_CHUNK * per_point
  steady_ms = max(kernel_ms, host_ms)
  return {
      'batch_chunks': batch, 'chunks_per_lead_and_launch': k,
      'value': PTS_PER_CHUNK / steady_ms * 1e3,
      'unit': 'grid-point-evals/s',
      'steady_ms_per_chunk': steady_ms, 'host_ms_per_chunk': host_ms,
      'wall_s': {str(n): v for n, v in walls.items()},
      'result': {'bytes': marks['result_bytes'], 's': marks['result_s'],
                 'GBps_incl_queue_drain':
                     marks['result_bytes'] / marks['result_s'] / 1e9},
      'fused_launches': len(ms),
      'roofline': {
          'bound': 'hbm', 'unit': 'GB/s', 'peak': HBM_PEAK_GBPS,
          'kernel': 'spatial_accumulate_addr_kernel<float,4,false> over '
                    f'{SLABS_PER_CHUNK} destinations x {k} steps',
          'kernel_ms_per_chunk': kernel_ms,
          'algorithmic_bytes_per_point': per_point,
          'algorithmic_bytes_per_chunk': bytes_per_chunk,
          'achieved': bytes_per_chunk / kernel_ms / 1e6,
          'frac': bytes_per_chunk / kernel_ms / 1e6 / HBM_PEAK_GBPS,
      },
      'metrics': list(cfg.metrics),
      'what': 'bias + mse + mae maps of 13 variables (85 slabs) per chunk '
              'into the float64 running means: 8 B/pt read per chunk + 3 x 16 '
              'B/pt read-modify-write of the sums per launch of k chunks of '
              'one lead (the fused kernel; the two SpatialSEEPS maps of '
              '--compute_seeps are computed and accumulated per slab beside '
              'it); `value` = points per chunk / max(kernel, '
              'host) time per chunk (the walls include the generic first '
              'chunk and the 8.5 GB result copy)',
  }


def measure(chunks, cfg, batch, timed_events: bool = True) -> dict:
  """`batch`: chunks per window, or None = evaluate_chunks' default (as many
  as hold 16 GiB of input: 24 of these chunks)."""
  import torch
  from weatherbench2_amd import engine, evaluation
  marks = {}
  sinks = (evaluation.RunningMean, evaluation.RunningConcat)
  real = [k.result for k in sinks]

  def hooked(real_result):
    def result(self):  # the host is done with the chunks
      marks.setdefault('enqueued', time.perf_counter())
      return real_result(self)
    return result
  for k, r in zip(sinks, real):
    k.result = hooked(r)

  def one_pass(hook):
    old = engine.set_launch_hook(hook)
    try:
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      out = evaluation.evaluate_chunks(
          chunks, cfg, False, prefetch=0,
          **({} if batch is None else {'batch_chunks': batch}))
      torch.cuda.synchronize()
      return t0, time.perf_counter(), out
    finally:
      engine.set_launch_hook(old)
  try:
    # throughput, wall and host time from a pass WITHOUT events in the queue
    # (a pair of timing events per replay costs the chunk-by-chunk run ~10 %);
    # the kernel time from a second pass with them
    ev = _Events(False)
    t0, t1, out = one_pass(ev)
    if timed_events:
      enqueued = marks.pop('enqueued')
      del out
      timed = _Events(True)
      _, _, out = one_pass(timed)
      marks['enqueued'] = enqueued
      ev.pairs = timed.pairs
  finally:
    for k, r in zip(sinks, real):
      k.result = r
  n = len(chunks)
  import gc
  from weatherbench2_amd import program as program_lib
  gc.collect()
  replay_stats = list(program_lib.REPLAY_STATS)
  del program_lib.REPLAY_STATS[:]
  if batch is None:
    first = sum(evaluation._input_bytes(ds) for ds in chunks[0])
    batch = int(min(evaluation.AUTO_BATCH_MAX,
                    max(1, evaluation.AUTO_BATCH_BYTES // first)))
    auto = True
  else:
    auto = False
  leg = {
      'batch_chunks': batch, 'default_window': auto, 'chunks': n,
      'value': n * PTS_PER_CHUNK / (t1 - t0), 'unit': 'grid-point-evals/s',
      'wall_ms_per_chunk': (t1 - t0) / n * 1e3,
      'host_ms_per_chunk': (marks['enqueued'] - t0) / n * 1e3,
      'k1_launches_per_chunk': ev.launches / n,
  }
  if replay_stats:   # host ms per wb2_program_replay call, by phase
    leg['replay_host_ms'] = {k: round(v, 4) for k, v in max(
        replay_stats, key=lambda d: d['replays']).items()}
  if timed_events and ev.pairs:
    ms = [a.elapsed_time(b) for a, b in ev.pairs]
    # every event pair brackets one fused launch (+ its K2): the deterministic
    # suite over all variables, the wind vectors, the SEEPS passes
    total_ms = sum(ms)
    strict = DET_BYTES_PER_CHUNK + SEEPS_BYTES_PER_CHUNK
    # wind vectors: from the read of the per-variable metrics (the pair
    # kernel) unless WB2HIP_WIND_PAIRS=0 brings the MODE_WIND launch back
    paired = os.environ.get('WB2HIP_WIND_PAIRS', '1') != '0'
    read = strict + (0.0 if paired else WIND_BYTES_PER_CHUNK) + (
        SEEPS_REREAD_PER_CHUNK)
    n_pairs = WIND_SLABS_PER_CHUNK * batch
    det = ('stream_partials_kernel<float,4,DET_ACC,WF> over '
           f'{(SLABS_PER_CHUNK - 2 * WIND_SLABS_PER_CHUNK) * batch} slabs + '
           f'stream_pair_kernel<float,4,ACC,WF> over {n_pairs} (u, v) pairs'
           if paired else
           'stream_partials_kernel<float,4,DET_ACC,WF> over '
           f'{SLABS_PER_CHUNK * batch} slabs + <float,4,WIND,WF> over '
           f'{n_pairs}')
    leg['roofline'] = {
        'bound': 'hbm', 'unit': 'GB/s', 'peak': HBM_PEAK_GBPS,
        'kernel': f'{det} + 2 x <float,4,SEEPS> over '
                  f'{batch} per window (K2 inside the brackets)',
        'k1_ms_per_chunk': total_ms / n,
        'algorithmic_bytes_per_chunk': strict,
        'bytes_read_per_chunk': read,
        # the wind-vector and SEEPS passes read inputs the first pass has read
        'traffic_over_algorithmic': read / strict,
        'achieved': n * strict / total_ms / 1e6,
        'frac': n * strict / total_ms / 1e6 / HBM_PEAK_GBPS,
        'frac_of_bytes_read': n * read / total_ms / 1e6 / HBM_PEAK_GBPS,
    }
  del out
  return leg


def host_chunks(chunks, pool: int):
  """The same chunk list with every FORECAST variable as a pageable NumPy
  array (what xbeam.DatasetToChunks hands _evaluate_chunk, evaluation.py:
  693-705); truth chunks stay device-resident views, the climatology resident.
  `pool` distinct host chunks (353 MB each) come round."""
  from weatherbench2_amd import xarray_lite as xl
  host_pool = []
  for f, _ in chunks[:pool]:
    host_pool.append({k: v.data.cpu().numpy() for k, v in f.data_vars.items()})
  out = []
  for j, (f, t) in enumerate(chunks):
    arrays = host_pool[j % pool]
    fd = xl.Dataset({k: xl.DataArray(arrays[k], f.data_vars[k].dims)
                     for k in f.data_vars}, dict(f.coords))
    out.append((fd, t))
  return out


def measure_host_fed(chunks, cfg, n_chunks: int = 144, pool: int = 6,
                     prefetch: int = 2) -> dict:
  """`api_official_chunk.host_fed`: forecast chunks arrive as pageable NumPy
  arrays and cross PCIe inside evaluate_chunks (the fetch thread stages them
  through the uploader while the main thread evaluates the previous window);
  reports evals/s, the H2D rate achieved and the wall time per chunk."""
  import torch
  from weatherbench2_amd import evaluation, feeder
  fed = host_chunks(chunks[:n_chunks], pool)
  nbytes = sum(v.data.nbytes for v in fed[0][0].data_vars.values())
  legs = {}
  for name, kwargs in (('default_window', {}), ('chunk_by_chunk',
                                                {'batch_chunks': 1})):
    # uploader, ring, plans -- and the allocator: a window in flight plus the
    # next one being staged hold ~2 x 24 chunks of device memory, which the
    # first pass over the list has to hipMalloc
    evaluation.evaluate_chunks(fed, cfg, False, prefetch=prefetch, **kwargs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evaluation.evaluate_chunks(fed, cfg, False, prefetch=prefetch, **kwargs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    legs[name] = {
        'value': len(fed) * PTS_PER_CHUNK / dt, 'unit': 'grid-point-evals/s',
        'h2d_GBps': len(fed) * nbytes / dt / 1e9,
        'wall_ms_per_chunk': dt / len(fed) * 1e3, 'chunks': len(fed)}
  # the uploader alone: one chunk's arrays, back to back (no evaluation)
  arrays = [v.data for v in fed[0][0].data_vars.values()]
  for a in arrays:
    feeder.upload(a, dev_of(chunks))
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(4):
    for a in arrays:
      feeder.upload(a, dev_of(chunks))
  torch.cuda.synchronize()
  up = 4 * nbytes / (time.perf_counter() - t0) / 1e9
  out = dict(legs['default_window'])
  out.update(
      by_window=legs, uploader_alone_GBps=up,
      copy_threads=feeder.copy_threads(), bytes_per_chunk=nbytes,
      what=('forecast chunks as pageable NumPy arrays (13 variables, 353 MB '
            'per chunk) through evaluate_chunks(prefetch=%d): staged by the '
            'fetch thread (wb2_uploader_upload: %d copy threads into a pinned '
            'ring, DMA overlapped), truth / climatology resident' % (
                prefetch, feeder.copy_threads())))
  return out


def dev_of(chunks):
  return next(iter(chunks[0][1].data_vars.values())).data.device


def run(dev, n_chunks: int = 4608, pool: int = 32,
        batches=(1, 16, 32, None), headline_batch=None,
        host_fed: bool = False) -> dict:
  """The `api_official_chunk` object of the bench record."""
  import torch
  from weatherbench2_amd import metrics as gm
  chunks, cfg = build(dev, n_chunks, pool)
  measure(chunks[:8], cfg, 4, timed_events=False)  # plans, tables, allocator
  legs = {}
  for b in batches:
    gm.clear_caches()
    measure(chunks[:max(2 * (b or 24), 8)], cfg, b, timed_events=False)  # warm
    # the whole list: the first chunk (window) of a structure runs the generic
    # path twice to build its program (program.py) -- a one-off that a
    # production run spreads over ~10^4-10^5 chunks, here over `n_chunks`
    name = 'default' if b is None else str(b)
    legs[name] = measure(chunks, cfg, b)
  head_name = 'default' if headline_batch is None else str(headline_batch)
  head = legs[head_name] if head_name in legs else legs[list(legs)[-1]]
  out = dict(head)
  out['by_batch_chunks'] = legs
  # the same chunks through the `deterministic_temporal` config: every launch
  # of `deterministic` (rmse is one more row of the same fold), the results
  # kept per (init_time, lead_time) instead of averaged
  try:
    cfg_t = temporal_config(cfg)
    temporal = {}
    for b in (1, None):
      gm.clear_caches()
      measure(chunks[:max(2 * (b or 24), 8)], cfg_t, b, timed_events=False)
      temporal['default' if b is None else str(b)] = measure(chunks, cfg_t, b)
    out['deterministic_temporal'] = temporal
  except Exception as e:
    out['deterministic_temporal'] = {'error': f'{type(e).__name__}: {e}'}
  # both configs of the documented command line
  # (--eval_configs=deterministic,deterministic_temporal) in ONE call: every
  # chunk is read once, the launches serve both
  try:
    pair = {'deterministic': cfg, 'deterministic_temporal': temporal_config(cfg)}
    together = {}
    for b in (1, None):
      gm.clear_caches()
      measure(chunks[:max(2 * (b or 24), 8)], pair, b, timed_events=False)
      together['default' if b is None else str(b)] = measure(chunks, pair, b)
    out['deterministic_and_temporal'] = together
  except Exception as e:
    out['deterministic_and_temporal'] = {'error': f'{type(e).__name__}: {e}'}
  try:
    out['deterministic_spatial'] = measure_spatial(chunks, cfg)
  except Exception as e:
    out['deterministic_spatial'] = {'error': f'{type(e).__name__}: {e}'}
  if host_fed:
    try:
      out['host_fed'] = measure_host_fed(chunks, cfg)
    except Exception as e:
      out['host_fed'] = {'error': f'{type(e).__name__}: {e}'}
    if host_fed == 'both':
      try:  # both configs of the command line from ONE upload of every chunk
        out['host_fed_both_configs'] = measure_host_fed(
            chunks, {'deterministic': cfg,
                     'deterministic_temporal': temporal_config(cfg)})
      except Exception as e:
        out['host_fed_both_configs'] = {'error': f'{type(e).__name__}: {e}'}
  out['config'] = {
      'workload': ('official 0.25-degree deterministic chunking: '
                   'init_time=1,lead_time=1 chunks of 13 variables (6 x 13 '
                   f'levels + 7 surface = {SLABS_PER_CHUNK} slabs of '
                   f'{N_LAT}x{N_LON} f32), 16 regions incl. 3 land-sea-mask '
                   'regions, mse (+ 2 wind vectors) + acc + bias + mae + '
                   'seeps_24hr + seeps_6hr (--compute_seeps=True), '
                   'evaluation.evaluate_chunks from device-resident chunks; '
                   '`deterministic_temporal`: the same + '
                   'rmse_sqrt_before_time_avg with temporal_mean=False'),
      'pool_chunks': pool, 'points_per_chunk': PTS_PER_CHUNK}
  del chunks, cfg
  torch.cuda.empty_cache()
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--chunks', type=int, default=4608,
                  help='chunks per leg (a multiple of the 16- / 24- / 32-chunk '
                       'windows; every evaluate_chunks call pays ~20 ms of '
                       'one-offs -- the generic first window, the program build '
                       '--: 6 %% of a 1 536-chunk windowed leg, 2 %% of this one; '
                       'a year of 0.25-degree forecasts is ~29 000 chunks)')
  ap.add_argument('--pool', type=int, default=32,
                  help='distinct device-resident chunks (>= the largest window)')
  ap.add_argument('--batch', default='1,16,32,default')
  ap.add_argument('--host-fed', action='store_true',
                  help='also the leg whose forecast chunks are pageable NumPy '
                       'arrays')
  ap.add_argument('--profile', action='store_true',
                  help='cProfile of the host path at the first batch size')
  ap.add_argument('--sections', action='store_true',
                  help='wall time of the host path by section (no profiler)')
  args = ap.parse_args()
  import torch
  dev = torch.device('cuda', 0)
  batches = tuple(None if b == 'default' else int(b)
                  for b in args.batch.split(','))
  if args.sections:
    # wall time of the host path's sections at batch_chunks = first entry,
    # without a profiler's per-call overhead (a few timers per chunk)
    import collections
    from weatherbench2_amd import engine, evaluation, metrics as gm
    from weatherbench2_amd import xarray_lite as xl
    spent = collections.defaultdict(float)
    calls = collections.Counter()

    def timed(owner, attr, label=None):
      fn = getattr(owner, attr)
      label = label or attr

      def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
          return fn(*a, **k)
        finally:
          spent[label] += time.perf_counter() - t0
          calls[label] += 1
      setattr(owner, attr, wrapper)
    for owner, attr in ((evaluation, '_metric_and_region_loop'),
                        (evaluation.RunningMean, 'add'),
                        (evaluation, 'concat_chunks'),
                        (xl, 'merge_metrics'), (gm, '_run_group'),
                        (gm, '_assemble'), (gm, '_det_plan'),
                        (gm, '_wind_plan'), (gm, '_prepare_inputs'),
                        (gm, '_climatology_slabs'),
                        (gm, '_reference_result_dtype'), (gm, '_det_key'),
                        (engine, 'time_accumulate'), (engine, 'upload_table'),
                        (engine, 'stream_reduce_addr')):
      timed(owner, attr)
    chunks, cfg = build(dev, args.chunks, args.pool)
    measure(chunks[:8], cfg, batches[0], timed_events=False)
    spent.clear()
    calls.clear()
    leg = measure(chunks, cfg, batches[0], timed_events=False)
    n = len(chunks)
    print(f"wall {leg['wall_ms_per_chunk']:.3f} ms per chunk")
    for k, v in sorted(spent.items(), key=lambda kv: -kv[1]):
      print(f'{k:28s} {v / n * 1e3:7.3f} ms per chunk  {calls[k] / n:6.1f} calls')
    return
  if args.profile:
    import cProfile
    import pstats
    chunks, cfg = build(dev, args.chunks, args.pool)
    measure(chunks[:8], cfg, batches[0], timed_events=False)
    pr = cProfile.Profile()
    pr.enable()
    measure(chunks, cfg, batches[0], timed_events=False)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(40); pstats.Stats(pr).sort_stats("tottime").print_stats(35)
    return
  print(json.dumps(run(dev, args.chunks, args.pool, batches,
                       headline_batch=batches[-1],
                       host_fed='both' if args.host_fed else False)))


if __name__ == '__main__':
  main()
