"""Chunk programs (weatherbench2_amd/program.py): the loop over one chunk
structure recorded once and replayed with new addresses must give the bits of
the generic path -- for every window size, chunk order, number of leads (the
accumulators grow), host-fed chunks -- and must actually be what runs.
Reference: /root/reference/weatherbench2/evaluation.py:583-599 (the per-chunk
call), 693-705 (chunk source), 735-744 (the temporal mean)."""
import numpy as np
import pytest

from tests import helpers, official_chunks as oc

pytestmark = pytest.mark.gpu


def _setup(**kw):
  from weatherbench2_amd import config, evaluation, metrics as gm
  forecast, truth, clim = oc.make(**kw)
  lat, lon = forecast.coords['latitude'], forecast.coords['longitude']
  lsm = oc.land_sea_mask(len(lat), len(lon))
  oregions = oc.oracle_regions(lat, lon, lsm)
  gregions = {k: helpers.to_gpu_region(v) for k, v in oregions.items()}
  hf, ht, hc = (helpers.to_gpu_dataset(x) for x in (forecast, truth, clim))
  gf, gt, gc = (evaluation.make_resident(x) for x in (hf, ht, hc))
  cfg = config.Eval(metrics=oc.product_metrics(gm, gc), regions=gregions)
  return hf, ht, gf, gt, cfg


def _same(a, b):
  assert sorted(a.data_vars) == sorted(b.data_vars)
  for k in a.coords:
    ca, cb = a.coords[k], b.coords[k]
    np.testing.assert_array_equal(np.asarray(getattr(ca, 'values', ca)),
                                  np.asarray(getattr(cb, 'values', cb)))
  for name in a.data_vars:
    x, y = np.asarray(a[name].values), np.asarray(b[name].values)
    assert a[name].dims == b[name].dims and x.dtype == y.dtype
    assert np.array_equal(x, y, equal_nan=True), name


def _count_runs(monkeypatch):
  from weatherbench2_amd import program
  calls = []
  real = program.ChunkProgram.run

  def run(self, *a, **k):
    calls.append(1)
    return real(self, *a, **k)
  monkeypatch.setattr(program.ChunkProgram, 'run', run)
  return calls


@pytest.mark.parametrize('order', ['init', 'lead'])
@pytest.mark.parametrize('batch', [1, 2, 4, None])
def test_programs_give_the_generic_paths_bits(batch, order, monkeypatch):
  from weatherbench2_amd import evaluation
  _, _, gf, gt, cfg = _setup(n_init=5, n_lead=3, n_lat=31, n_lon=72)
  chunks = oc.chunk_pairs(gf, gt, order=order)
  kwargs = {} if batch is None else {'batch_chunks': batch}
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', '0')
  want = evaluation.evaluate_chunks(chunks, cfg, False, prefetch=0, **kwargs)
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', '1')
  calls = _count_runs(monkeypatch)
  got = evaluation.evaluate_chunks(chunks, cfg, False, prefetch=0, **kwargs)
  _same(got, want)
  if batch == 1:
    # 15 chunks of one structure: the first builds the program, 14 replay it
    assert len(calls) == len(chunks) - 1
  elif batch == 2:
    assert len(calls) >= 1   # windows of 2 (and a ragged last one)
  # and with both paths run on every chunk
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', 'verify')
  again = evaluation.evaluate_chunks(chunks, cfg, False, prefetch=0, **kwargs)
  _same(again, want)


def test_programs_with_skipna_nans_and_growing_accumulators(monkeypatch):
  """11 leads (the accumulators start with 8 rows and grow), NaN patches,
  skipna: chunk by chunk, programs on and off."""
  from weatherbench2_amd import evaluation
  _, _, gf, gt, cfg = _setup(n_init=3, n_lead=11, n_lat=19, n_lon=36,
                             nan_frac=0.01)
  chunks = oc.chunk_pairs(gf, gt)
  for skipna in (False, True):
    monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', '0')
    want = evaluation.evaluate_chunks(chunks, cfg, skipna, prefetch=0,
                                      batch_chunks=1)
    monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', '1')
    calls = _count_runs(monkeypatch)
    got = evaluation.evaluate_chunks(chunks, cfg, skipna, prefetch=0,
                                     batch_chunks=1)
    _same(got, want)
    assert len(calls) == len(chunks) - 1
    assert len(got.coords['lead_time']) == 11


def test_host_fed_chunks_replay_too(monkeypatch):
  """Forecast chunks as NumPy arrays, staged by the fetch thread: the staged
  device tensors have one structure, so the program serves them as well."""
  from weatherbench2_amd import evaluation
  hf, ht, gf, gt, cfg = _setup(n_init=4, n_lead=2, n_lat=31, n_lon=72)
  chunks = oc.chunk_pairs(gf, gt)
  fed = [(h, t) for (h, _), (_, t) in zip(oc.chunk_pairs(hf, ht), chunks)]
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', '0')
  want = evaluation.evaluate_chunks(chunks, cfg, False, prefetch=0,
                                    batch_chunks=1)
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', '1')
  monkeypatch.setattr(evaluation, '_STAGE_MIN_BYTES', 1024)
  calls = _count_runs(monkeypatch)
  got = evaluation.evaluate_chunks(fed, cfg, False, prefetch=2, batch_chunks=1)
  _same(got, want)
  assert len(calls) == len(chunks) - 1


def test_a_change_of_structure_gets_a_program_of_its_own(monkeypatch):
  """Two chunk lists of different grids through one evaluation each, and one
  list whose last chunk drops a variable: the odd chunk takes the generic path
  (or its own program), the result is that of the generic path."""
  from weatherbench2_amd import evaluation
  from weatherbench2_amd import xarray_lite as xl
  _, _, gf, gt, cfg = _setup(n_init=4, n_lead=2, n_lat=31, n_lon=72)
  chunks = oc.chunk_pairs(gf, gt)
  f, t = chunks[-1]
  drop = '2m_temperature'
  chunks[-1] = (
      xl.Dataset({k: v for k, v in f.data_vars.items() if k != drop},
                 dict(f.coords)),
      xl.Dataset({k: v for k, v in t.data_vars.items() if k != drop},
                 dict(t.coords)))
  outs = []
  for how in ('0', '1'):
    monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', how)
    try:
      outs.append(evaluation.evaluate_chunks(chunks, cfg, False, prefetch=0,
                                             batch_chunks=1))
    except Exception as e:   # the generic path's own complaint, both times
      outs.append(type(e))
  if isinstance(outs[0], type):
    assert outs[1] is outs[0]
  else:
    _same(outs[1], outs[0])


@pytest.mark.parametrize('batch', [1, 3, None])
def test_seeps_in_windows_and_programs(batch, monkeypatch):
  """SEEPS (metrics.py:417-524; `--compute_seeps=True` of the official command
  line) beside the deterministic suite: its pass has an auxiliary field (the
  masked dry fraction, resident per climatology) and a table that follows the
  valid time -- windows of chunks and replayed programs give the bits of the
  chunk-by-chunk generic path, which equals the oracle's SEEPS."""
  import torch
  from oracle import metrics_np as om
  from oracle.named import DS, NA
  from weatherbench2_amd import config, evaluation, metrics as gm
  from weatherbench2_amd import xarray_lite as xl
  forecast, truth, clim = oc.make(n_init=4, n_lead=2, n_lat=31, n_lon=72)
  name = '2m_temperature'
  rs = np.random.RandomState(5)
  shape = clim[name].data.shape
  cdims = clim[name].dims
  cvars = dict(clim.items())
  cvars[f'{name}_seeps_threshold'] = NA(
      rs.uniform(0.3, 1.2, size=shape).astype(np.float32), cdims)  # > dry
  frac = rs.uniform(0.0, 1.0, size=shape).astype(np.float32)
  frac[:, :, 3, 5] = np.nan
  cvars[f'{name}_seeps_dry_fraction'] = NA(frac, cdims)
  clim = DS(cvars, clim.coords)
  lat, lon = forecast.coords['latitude'], forecast.coords['longitude']
  oregions = {'global': None, 'tropics': None}
  from oracle import regions_np as oreg
  oregions = {'global': oreg.SliceRegion(),
              'tropics': oreg.SliceRegion(lat_slice=slice(-20, 20))}
  gregions = {k: helpers.to_gpu_region(v) for k, v in oregions.items()}
  gf, gt, gc = (evaluation.make_resident(helpers.to_gpu_dataset(x))
                for x in (forecast, truth, clim))
  metrics = oc.product_metrics(gm, gc)
  metrics['seeps'] = gm.SEEPS(climatology=gc, precip_name=name,
                              dry_threshold_mm=100.0)
  cfg = config.Eval(metrics=metrics, regions=gregions)
  chunks = oc.chunk_pairs(gf, gt)
  kwargs = {} if batch is None else {'batch_chunks': batch}
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', '0')
  want = evaluation.evaluate_chunks(chunks, cfg, False, prefetch=0,
                                    batch_chunks=1)
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', '1')
  calls = _count_runs(monkeypatch)
  got = evaluation.evaluate_chunks(chunks, cfg, False, prefetch=0, **kwargs)
  _same(got, want)
  if batch == 1:
    assert len(calls) == len(chunks) - 1
  # against the oracle: the time mean of the per-chunk SEEPS
  oseeps = om.SEEPS(climatology=clim, precip_name=name, dry_threshold_mm=100.0)
  mi = list(got.coords['metric']).index('seeps')
  for ri, (rname, region) in enumerate(oregions.items()):
    with np.errstate(all='ignore'):
      per = oseeps.compute_chunk(forecast, truth, region=region)[name]
    ax = per.dims.index('init_time')
    want_mean = np.asarray(per.data, dtype=np.float64).mean(axis=ax)
    vals = got[name].values[mi, ri]
    helpers.assert_close(vals, want_mean.reshape(vals.shape), rtol=1e-9,
                         atol=1e-12, err_msg=rname)


@pytest.mark.parametrize('order', ['init', 'lead'])
@pytest.mark.parametrize('batch', [1, 2, None])
def test_temporal_mean_false_keeps_every_chunk(batch, order, monkeypatch):
  """`temporal_mean=False` (config.py:55; the `deterministic_temporal` config
  of scripts/evaluate.py:479-487; evaluation.py:735 skips the mean): the
  per-chunk values filed under (init_time, lead_time) -- the program path, the
  generic path and whole-dataset evaluation give the same bits, and those are
  the oracle's per-chunk values."""
  import dataclasses
  from oracle import evaluation_np as oe
  from weatherbench2_amd import evaluation
  forecast, truth, clim = oc.make(n_init=5, n_lead=3, n_lat=31, n_lon=72)
  hf, ht, gf, gt, cfg = _setup(n_init=5, n_lead=3, n_lat=31, n_lon=72)
  cfg = dataclasses.replace(cfg, temporal_mean=False)
  chunks = oc.chunk_pairs(gf, gt, order=order)
  kwargs = {} if batch is None else {'batch_chunks': batch}
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', '0')
  want = evaluation.evaluate_chunks(chunks, cfg, False, prefetch=0,
                                    batch_chunks=1)
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', '1')
  calls = _count_runs(monkeypatch)
  got = evaluation.evaluate_chunks(chunks, cfg, False, prefetch=0, **kwargs)
  _same(got, want)
  if batch == 1:
    assert len(calls) == len(chunks) - 1
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', 'verify')
  _same(evaluation.evaluate_chunks(chunks, cfg, False, prefetch=0, **kwargs),
        want)
  # the labels in dataset order whichever way the chunks came
  np.testing.assert_array_equal(
      np.sort(np.asarray(got.coords['init_time'])),
      np.asarray(gf.coords['init_time']))
  assert got['geopotential'].dims[:2] == ('metric', 'region')
  assert 'init_time' in got['geopotential'].dims
  # against the oracle: per-chunk values of the whole dataset
  lat, lon = forecast.coords['latitude'], forecast.coords['longitude']
  oregions = oc.oracle_regions(lat, lon, oc.land_sea_mask(len(lat), len(lon)))
  per_chunk = oe.metric_and_region_loop(
      forecast, truth, oc.oracle_metrics(clim), oregions, False,
      compute_chunk=True)
  labels_m, labels_r = list(got.coords['metric']), list(got.coords['region'])
  init_pos = [list(np.asarray(got.coords['init_time'])).index(x)
              for x in np.asarray(forecast.coords['init_time'])]
  lead_pos = [list(np.asarray(got.coords['lead_time'])).index(x)
              for x in np.asarray(forecast.coords['lead_time'])]
  checked = 0
  for (mname, rname), ds in per_chunk.items():
    for var, v in ds.items():
      res = got[var]
      order_ = [d for d in res.dims if d not in ('metric', 'region')]
      vals = res.values[labels_m.index(mname), labels_r.index(rname)]
      vals = np.take(vals, init_pos, axis=order_.index('init_time'))
      vals = np.take(vals, lead_pos, axis=order_.index('lead_time'))
      vals = np.transpose(vals, [order_.index(d) for d in v.dims])
      helpers.assert_close(vals, v.data, rtol=1e-9, atol=1e-12,
                           err_msg=f'{mname}/{rname}/{var}')
      checked += 1
  assert checked > 10


def test_temporal_mean_false_rejects_a_chunk_that_comes_twice(monkeypatch):
  import dataclasses
  from weatherbench2_amd import evaluation
  _, _, gf, gt, cfg = _setup(n_init=3, n_lead=2, n_lat=19, n_lon=36)
  cfg = dataclasses.replace(cfg, temporal_mean=False)
  chunks = oc.chunk_pairs(gf, gt)
  for how in ('0', '1'):
    monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', how)
    with pytest.raises(ValueError, match='came twice'):
      evaluation.evaluate_chunks(chunks + chunks[2:3], cfg, False, prefetch=0,
                                 batch_chunks=1)


def _ensemble_chunks(n_init=4, n_lead=3, n_member=5, n_lat=19, n_lon=36,
                     member_first=True, nan=False, seed=0):
  """(oracle forecast, truth) + device-resident product chunks of an ensemble
  forecast: three variables (two with levels), `number` members."""
  from oracle.named import DS, NA
  from weatherbench2_amd import evaluation
  rs = np.random.RandomState(seed)
  lat = np.linspace(-90, 90, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  level = np.array([500, 850])
  init = (np.datetime64('2020-01-01T00', 'ns') +
          np.arange(n_init) * np.timedelta64(12, 'h'))
  lead = (np.arange(n_lead) * np.timedelta64(6, 'h')).astype('timedelta64[ns]')
  core3 = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
  core2 = ('init_time', 'lead_time', 'latitude', 'longitude')
  put = (lambda dims: ('number',) + dims) if member_first else (
      lambda dims: dims[:2] + ('number',) + dims[2:])
  size = {'number': n_member, 'init_time': n_init, 'lead_time': n_lead,
          'level': 2, 'latitude': n_lat, 'longitude': n_lon}

  def field(dims):
    x = rs.normal(size=[size[d] for d in dims]).astype(np.float32)
    if nan:
      x[rs.rand(*x.shape) < 0.01] = np.nan
    return NA(x, dims)
  coords = {'init_time': init, 'lead_time': lead, 'level': level,
            'latitude': lat, 'longitude': lon, 'number': np.arange(n_member)}
  forecast = DS({'z': field(put(core3)), 'q': field(put(core3)),
                 't2m': field(put(core2))}, coords)
  tcoords = {k: v for k, v in coords.items() if k != 'number'}
  truth = DS({'z': field(core3), 'q': field(core3), 't2m': field(core2)},
             tcoords)
  gf, gt = (evaluation.make_resident(helpers.to_gpu_dataset(x))
            for x in (forecast, truth))
  return forecast, truth, oc.chunk_pairs(gf, gt)


@pytest.mark.parametrize('n_member', [5, 7])
@pytest.mark.parametrize('member_first', [True, False])
@pytest.mark.parametrize('skipna', [False, True])
def test_ensemble_passes_replay_too(skipna, member_first, n_member,
                                    monkeypatch):
  """The `probabilistic` config (scripts/evaluate.py:496-520) at the chunking
  of its command lines (`init_time=1,lead_time=1`,
  docs/source/official-evaluation.md:826-860): the K3 pass of every variable
  + its fold are recorded and replayed with new base pointers -- same bits as
  the generic path, which equals the oracle's time mean."""
  from oracle import evaluation_np as oe, metrics_np as om, regions_np as oreg
  from weatherbench2_amd import config, engine, evaluation, metrics as gm
  from weatherbench2_amd import program
  forecast, truth, chunks = _ensemble_chunks(member_first=member_first,
                                             nan=skipna, n_member=n_member)
  dim = 'number'
  names = {'crps': 'CRPS', 'crps_spread': 'CRPSSpread',
           'crps_skill': 'CRPSSkill', 'ensemble_mean_mse': 'EnsembleMeanMSE',
           'debiased_ensemble_mean_mse': 'DebiasedEnsembleMeanMSE',
           'ensemble_variance': 'EnsembleVariance'}
  oregions = {'global': oreg.SliceRegion(),
              'tropics': oreg.SliceRegion(lat_slice=slice(-20, 20))}
  cfg = config.Eval(
      metrics={k: getattr(gm, v)(ensemble_dim=dim) for k, v in names.items()},
      regions={k: helpers.to_gpu_region(v) for k, v in oregions.items()})
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', '0')
  want = evaluation.evaluate_chunks(chunks, cfg, skipna, prefetch=0,
                                    batch_chunks=1)
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', '1')
  calls = _count_runs(monkeypatch)
  got = evaluation.evaluate_chunks(chunks, cfg, skipna, prefetch=0,
                                   batch_chunks=1)
  _same(got, want)
  assert len(calls) == len(chunks) - 1, program.REASONS
  if not skipna:
    # the passes of z and q (same member stride) are ONE launch through their
    # slabs' addresses -- the kernel of the separate passes, whatever the
    # member count (5 has a sorting program of its own, 7 is hosted by 8's)
    for native in ('0', '1'):
      monkeypatch.setenv('WB2HIP_NATIVE_REPLAY', native)
      seen = []
      old = engine.set_launch_hook(lambda when, kernel: seen.append(kernel)
                                   if when == 'begin' else None)
      try:
        got4 = evaluation.evaluate_chunks(chunks[:4], cfg, skipna, prefetch=0,
                                          batch_chunks=1)
      finally:
        engine.set_launch_hook(old)
      # first chunk: 3 generic passes + the trial of the fused launch; then
      # 3 replays of (z + q, t2m) -- from Python launch by launch, or one
      # wb2_program_replay call per chunk
      if native == '0':
        assert seen.count('ens_partials') == 3 + 1 + 3 * 2, seen
        first4 = got4
      else:
        assert seen.count('ens_partials') == 3 + 1, seen
        assert seen.count('stream_partials') == 3, seen
        _same(got4, first4)
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', 'verify')
  _same(evaluation.evaluate_chunks(chunks, cfg, skipna, prefetch=0,
                                   batch_chunks=1), want)
  # against the oracle
  ometrics = {k: getattr(om, v)(ensemble_dim=dim) for k, v in names.items()}
  per_chunk = oe.metric_and_region_loop(forecast, truth, ometrics, oregions,
                                        skipna, compute_chunk=True)
  labels_m, labels_r = list(got.coords['metric']), list(got.coords['region'])
  for (mname, rname), ds in per_chunk.items():
    for var, v in ds.items():
      ax = v.dims.index('init_time')
      data = np.asarray(v.data, dtype=np.float64)
      with np.errstate(all='ignore'):
        mean = (np.nanmean if skipna else np.mean)(data, axis=ax)
      dims = tuple(d for d in v.dims if d != 'init_time')
      res = got[var]
      rdims = [d for d in res.dims if d not in ('metric', 'region')]
      vals = res.values[labels_m.index(mname), labels_r.index(rname)]
      vals = np.transpose(vals, [rdims.index(d) for d in dims])
      helpers.assert_close(vals, mean, rtol=1e-6, atol=1e-9,
                           err_msg=f'{mname}/{rname}/{var}')


@pytest.mark.parametrize('n_member,double', [(1, False), (3, True)])
def test_ensemble_windows_of_one_member_and_of_float64(n_member, double,
                                                       monkeypatch):
  """The edges of the in-place window read: a single member (member stride 0;
  CRPSSpread's zeros / NaN-over-an-empty-region rule) and float64 chunks."""
  import torch
  from weatherbench2_amd import config, evaluation, metrics as gm
  from weatherbench2_amd import xarray_lite as xl
  _, _, chunks = _ensemble_chunks(n_init=4, n_lead=2, n_member=n_member)

  def own(ds):
    return xl.Dataset({k: xl.DataArray(
        (v.data.double() if double else v.data).contiguous().clone(), v.dims)
                       for k, v in ds.data_vars.items()}, dict(ds.coords))
  chunks = [(own(f), own(t_)) for f, t_ in chunks]
  dim = 'number'
  cfg = config.Eval(
      metrics={'crps': gm.CRPS(ensemble_dim=dim),
               'crps_spread': gm.CRPSSpread(ensemble_dim=dim),
               'crps_skill': gm.CRPSSkill(ensemble_dim=dim),
               'debiased': gm.DebiasedEnsembleMeanMSE(ensemble_dim=dim),
               'ensemble_variance': gm.EnsembleVariance(ensemble_dim=dim)},
      regions={'global': gm_regions().SliceRegion(),
               'nowhere': gm_regions().SliceRegion(lat_slice=slice(91, 95))})
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', '0')
  want = evaluation.evaluate_chunks(chunks, cfg, False, prefetch=0,
                                    batch_chunks=1)
  for how in ('0', '1'):
    monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', how)
    for batch in (2, None):
      kwargs = {} if batch is None else {'batch_chunks': batch}
      _same(evaluation.evaluate_chunks(chunks, cfg, False, prefetch=0,
                                       **kwargs), want)
  del torch


@pytest.mark.parametrize('member_first', [True, False])
@pytest.mark.parametrize('skipna', [False, True])
@pytest.mark.parametrize('batch', [2, 4, None])
def test_ensemble_windows_read_the_chunks_where_they_lie(batch, skipna,
                                                          member_first,
                                                          monkeypatch):
  """The `probabilistic` config over WINDOWS of its (init_time=1, lead_time=1)
  chunks: K3 takes member 0's slab and the truth slab of every (chunk, level)
  by address (wb2_ens_partials_addr) -- nothing is concatenated -- and gives,
  per slab, what it gives chunk by chunk: the same bits for every window size,
  with and without programs."""
  from weatherbench2_amd import config, engine, evaluation, metrics as gm
  from weatherbench2_amd import xarray_lite as xl
  _, _, chunks = _ensemble_chunks(n_init=5, n_lead=3, member_first=member_first,
                                  nan=skipna, n_member=6)

  def own(ds):   # every chunk an allocation of its own, as a reader returns it
    return xl.Dataset({k: xl.DataArray(v.data.contiguous().clone(), v.dims)
                       for k, v in ds.data_vars.items()}, dict(ds.coords))
  chunks = [(own(f), own(t_)) for f, t_ in chunks]
  dim = 'number'
  cfg = config.Eval(
      metrics={'crps': gm.CRPS(ensemble_dim=dim),
               'crps_spread': gm.CRPSSpread(ensemble_dim=dim),
               'ensemble_mean_mse': gm.EnsembleMeanMSE(ensemble_dim=dim),
               'ensemble_variance': gm.EnsembleVariance(ensemble_dim=dim)},
      regions={'global': gm_regions().SliceRegion(),
               'north': gm_regions().SliceRegion(lat_slice=slice(20, 90))})
  kwargs = {} if batch is None else {'batch_chunks': batch}
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', '0')
  want = evaluation.evaluate_chunks(chunks, cfg, skipna, prefetch=0,
                                    batch_chunks=1)
  materialized = []
  real = xl.SlabConcat.materialize
  monkeypatch.setattr(xl.SlabConcat, 'materialize',
                      lambda self, *a, **k: (materialized.append(1),
                                             real(self, *a, **k))[1])
  for how in ('0', '1', 'verify'):
    monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', how)
    seen = []
    old = engine.set_launch_hook(lambda when, kernel: seen.append(kernel)
                                 if when == 'begin' else None)
    try:
      got = evaluation.evaluate_chunks(chunks, cfg, skipna, prefetch=0,
                                       **kwargs)
    finally:
      engine.set_launch_hook(old)
    _same(got, want)
    assert not materialized, 'a window was copied together'
    if how == '0':
      # 3 variables per window, one K3 launch each (the 15 chunks are one
      # window by default; windows of 2 / 4 chunks that are no (init x lead)
      # rectangle are evaluated in smaller pieces)
      n = seen.count('ens_partials')
      assert (n == 3) if batch is None else (n < 3 * len(chunks)), seen


def gm_regions():
  from weatherbench2_amd import regions
  return regions


@pytest.mark.parametrize('batch', [1, 3, None])
def test_several_configs_share_one_pass_over_the_chunks(batch, monkeypatch):
  """`--eval_configs=deterministic,deterministic_temporal` of the documented
  0.25-degree command line (docs/source/official-evaluation.md:537-556;
  evaluation.py:805-828 builds one branch per config, each reading the chunks
  again): evaluate_chunks({name: Eval}) reads every chunk once -- the K1
  launches of the chunk serve both configs -- and returns per config exactly
  what a call of its own returns."""
  import dataclasses
  from weatherbench2_amd import engine, evaluation, metrics as gm
  _, _, gf, gt, cfg = _setup(n_init=5, n_lead=3, n_lat=31, n_lon=72)
  wv = [gm.WindVectorRMSESqrtBeforeTimeAvg(u_name=u, v_name=v, vector_name=n)
        for u, v, n in oc.WIND]
  temporal = dataclasses.replace(
      cfg, temporal_mean=False,
      metrics={**cfg.metrics, 'rmse_sqrt_before_time_avg':
               gm.RMSESqrtBeforeTimeAvg(wind_vector_rmse=wv)})
  both = {'deterministic': cfg, 'deterministic_temporal': temporal}
  chunks = oc.chunk_pairs(gf, gt)
  kwargs = {} if batch is None else {'batch_chunks': batch}
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', '0')
  want = {k: evaluation.evaluate_chunks(chunks, c, False, prefetch=0,
                                        batch_chunks=1)
          for k, c in both.items()}

  def count_launches(fn):
    seen = []
    old = engine.set_launch_hook(lambda when, kernel: seen.append(kernel)
                                 if when == 'begin' else None)
    try:
      out = fn()
    finally:
      engine.set_launch_hook(old)
    return out, seen.count('stream_partials')
  for how in ('0', '1', 'verify'):
    monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', how)
    got, together = count_launches(lambda: evaluation.evaluate_chunks(
        chunks, both, False, prefetch=0, **kwargs))
    assert sorted(got) == sorted(both)
    for k in both:
      _same(got[k], want[k])
    if how == '1' and batch == 1:
      _, alone = count_launches(lambda: evaluation.evaluate_chunks(
          chunks, cfg, False, prefetch=0, batch_chunks=1))
      # the second config costs no launch of its own (the first chunk of the
      # structure is walked three times: record, probe-free verify, build)
      assert together <= alone + 2, (together, alone)


def test_configs_that_replace_the_forecast_differently_are_refused():
  import dataclasses
  from weatherbench2_amd import evaluation
  _, _, gf, gt, cfg = _setup(n_init=2, n_lead=1)
  other = dataclasses.replace(cfg, evaluate_persistence=True)
  with pytest.raises(ValueError, match='baseline switches'):
    evaluation.evaluate_chunks(oc.chunk_pairs(gf, gt), {'a': cfg, 'b': other},
                               False, prefetch=0)
  with pytest.raises(ValueError, match='no eval config'):
    evaluation.evaluate_chunks(oc.chunk_pairs(gf, gt), {}, False, prefetch=0)
