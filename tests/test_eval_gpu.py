"""GPU parity of the metric x region loop and the temporal mean (-m gpu)."""
import numpy as np
import pytest

from oracle import fixtures
from oracle import metrics_np as om
from oracle import regions_np as oreg
from tests import helpers

pytestmark = pytest.mark.gpu


def test_metric_and_region_loop_matches_reference_layout():
  """evaluation.py:388-438: (metric, region, ...) layout, NaN-filled merge."""
  import torch
  assert torch.cuda.is_available()
  from weatherbench2_amd import config, evaluation, metrics as gm
  truth, forecast = fixtures.get_random_truth_and_forecast(
      variables=('u_component_of_wind', 'v_component_of_wind', 'geopotential'),
      spatial_resolution_in_degrees=10)
  clim = fixtures.random_like(fixtures.mock_hourly_climatology_data(
      hour_interval=3, variables_3d=list(truth.keys()), variables_2d=[],
      spatial_resolution_in_degrees=10), seed=5)
  oregions = {'global': oreg.SliceRegion(),
              'tropics': oreg.SliceRegion(lat_slice=slice(-20, 20)),
              'extra-tropics': oreg.ExtraTropicalRegion()}
  g = helpers.to_gpu_dataset
  gregions = {k: helpers.to_gpu_region(v) for k, v in oregions.items()}
  wv = dict(u_name='u_component_of_wind', v_name='v_component_of_wind',
            vector_name='wind_vector')
  ometrics = {'mse': om.MSE(wind_vector_mse=[om.WindVectorMSE(**wv)]),
              'acc': om.ACC(clim), 'bias': om.Bias(), 'mae': om.MAE()}
  gmetrics = {'mse': gm.MSE(wind_vector_mse=[gm.WindVectorMSE(**wv)]),
              'acc': gm.ACC(g(clim)), 'bias': gm.Bias(), 'mae': gm.MAE()}
  for temporal_mean in (True, False):
    cfg = config.Eval(metrics=gmetrics, regions=gregions,
                      temporal_mean=temporal_mean)
    got = evaluation._metric_and_region_loop(g(forecast), g(truth), cfg,
                                             skipna=False)
    # xr.merge joins `metric` with an outer join (a sorted union); regions are
    # concatenated in the order given
    assert list(got.coords['metric']) == sorted(ometrics)
    assert list(got.coords['region']) == list(oregions)
    for mname, metric in ometrics.items():
      mi = list(got.coords['metric']).index(mname)
      for ri, (rname, region) in enumerate(oregions.items()):
        fn = metric.compute if temporal_mean else metric.compute_chunk
        want = fn(forecast, truth, region=region)
        for var in got.keys():
          arr = got[var].values[mi, ri]
          if var in want:
            helpers.assert_close(arr, want[var].data, rtol=1e-9, atol=1e-12,
                                 err_msg=f'{mname}/{rname}/{var}')
          else:  # wind_vector only exists for mse: NaN-filled by the merge
            assert np.isnan(arr).all()
    assert got['geopotential'].dims[:2] == ('metric', 'region')


def test_evaluate_chunks_equals_full_time_mean():
  """Chunked evaluation + RunningMean == Metric.compute on the whole period
  (the reference pins this as in-memory == Beam, evaluation_test.py:30-128)."""
  from weatherbench2_amd import config, evaluation, metrics as gm
  truth, forecast = fixtures.get_random_truth_and_forecast(
      spatial_resolution_in_degrees=10)
  forecast = fixtures.insert_nan(forecast, 0.001, seed=1)
  g = helpers.to_gpu_dataset
  regions = {'global': None, 'tropics': oreg.SliceRegion(
      lat_slice=slice(-20, 20))}
  gregions = {k: helpers.to_gpu_region(v) for k, v in regions.items()}
  n_time = forecast.sizes['time']
  chunks = [(g(forecast.isel(time=slice(i, i + 2))),
             g(truth.isel(time=slice(i, i + 2)))) for i in range(0, n_time, 2)]
  for skipna in (False, True):
    cfg = config.Eval(metrics={'rmse': gm.RMSESqrtBeforeTimeAvg(),
                               'bias': gm.Bias()}, regions=gregions)
    got = evaluation.evaluate_chunks(chunks, cfg, skipna=skipna, device='cuda')
    labels = list(got.coords['metric'])  # merged labels come out sorted
    for name, metric in (('rmse', om.RMSESqrtBeforeTimeAvg()),
                         ('bias', om.Bias())):
      mi = labels.index(name)
      for ri, region in enumerate(regions.values()):
        want = metric.compute(forecast, truth, region=region, skipna=skipna)
        helpers.assert_close(got['geopotential'].values[mi, ri],
                             want['geopotential'].data, rtol=1e-9, atol=1e-12)


def test_threaded_callers_share_the_caches_safely():
  """Beam's DirectRunner calls compute_chunk from worker threads."""
  import threading
  from weatherbench2_amd import metrics as gm
  truth, forecast = fixtures.get_random_truth_and_forecast(
      spatial_resolution_in_degrees=10)
  g = helpers.to_gpu_dataset
  chunks = [(g(forecast.isel(time=slice(i, i + 1))),
             g(truth.isel(time=slice(i, i + 1)))) for i in range(8)]
  want = [om.MSE().compute_chunk(forecast.isel(time=slice(i, i + 1)),
                                 truth.isel(time=slice(i, i + 1)))
          ['geopotential'].data for i in range(8)]
  got, errors = [None] * 8, []

  def work(i):
    try:
      for _ in range(3):
        got[i] = gm.MSE().compute_chunk(*chunks[i])['geopotential'].values
        gm.MAE().compute_chunk(*chunks[i])
    except Exception as e:  # pragma: no cover
      errors.append(e)

  threads = [threading.Thread(target=work, args=(i,)) for i in range(8)]
  for th in threads:
    th.start()
  for th in threads:
    th.join()
  assert not errors, errors
  for a, b in zip(got, want):
    helpers.assert_close(a, b, rtol=1e-9, atol=1e-12)


def test_evaluate_chunks_with_map_valued_metrics_stays_on_device():
  """Spatial* maps and rank histograms flow through the metric
  loop and the running (sum, count) mean without leaving the device."""
  import torch
  from weatherbench2_amd import config, evaluation, metrics as gm
  truth, forecast = fixtures.get_random_truth_and_forecast(
      spatial_resolution_in_degrees=10, ensemble_size=4)
  forecast = fixtures.insert_nan(forecast, 0.002, seed=2)
  g = helpers.to_gpu_dataset
  n_time = forecast.sizes['time']
  chunks = [(g(forecast.isel(time=slice(i, i + 2))),
             g(truth.isel(time=slice(i, i + 2)))) for i in range(0, n_time, 2)]
  for skipna in (False, True):
    cfg = config.Eval(metrics={
        'crps': gm.SpatialCRPS(), 'mse': gm.SpatialEnsembleMeanMSE()})
    one = evaluation._metric_and_region_loop(chunks[0][0], chunks[0][1], cfg,
                                             skipna, compute_chunk=True)
    assert isinstance(one['geopotential'].data, torch.Tensor)
    assert one['geopotential'].data.is_cuda
    got = evaluation.evaluate_chunks(chunks, cfg, skipna=skipna)
    da = got['geopotential']
    assert da.dims[0] == 'metric' and 'time' not in da.dims
    for mi, metric in enumerate((om.SpatialCRPS(),
                                 om.SpatialEnsembleMeanMSE())):
      want = metric.compute(forecast, truth, skipna=skipna)['geopotential']
      have = xl_take(da, mi)
      w = want.transpose(*have[0]).data
      helpers.assert_close(have[1], w, rtol=2e-6, atol=1e-7)


def xl_take(da, metric_index):
  """(dims, values) of one metric of a merged result."""
  ax = da.dims.index('metric')
  dims = tuple(d for d in da.dims if d != 'metric')
  return dims, np.take(np.asarray(da.values), metric_index, axis=ax)


def test_loop_with_ensemble_metrics_and_foreign_metric_objects():
  """Ensemble scalars take the all-regions path; any other object exposing
  compute_chunk / compute (a user-defined metric) is still called per region,
  exactly like evaluation.py:408-435 does."""
  from weatherbench2_amd import config, evaluation, metrics as gm
  from weatherbench2_amd import xarray_lite as xl
  truth, forecast = fixtures.get_random_truth_and_forecast(
      spatial_resolution_in_degrees=10, ensemble_size=5)
  oregions = {'global': oreg.SliceRegion(),
              'tropics': oreg.SliceRegion(lat_slice=slice(-20, 20)),
              'extra-tropics': oreg.ExtraTropicalRegion()}
  g = helpers.to_gpu_dataset
  gregions = {k: helpers.to_gpu_region(v) for k, v in oregions.items()}
  calls = []

  class Foreign:  # not a weatherbench2_amd Metric
    def compute_chunk(self, forecast, truth, region=None, skipna=False):
      calls.append(region)
      return gm.CRPSSkill().compute_chunk(forecast, truth, region=region,
                                          skipna=skipna)

    def compute(self, forecast, truth, region=None, skipna=False):
      calls.append(region)
      return gm.CRPSSkill().compute(forecast, truth, region=region,
                                    skipna=skipna)

  ometrics = {'crps': om.CRPS(), 'spread': om.CRPSSpread(),
              'var': om.EnsembleVariance(), 'skill': om.CRPSSkill()}
  gmetrics = {'crps': gm.CRPS(), 'spread': gm.CRPSSpread(),
              'var': gm.EnsembleVariance(), 'skill': Foreign()}
  for temporal_mean in (True, False):
    calls.clear()
    cfg = config.Eval(metrics=gmetrics, regions=gregions,
                      temporal_mean=temporal_mean)
    got = evaluation._metric_and_region_loop(g(forecast), g(truth), cfg,
                                             skipna=False)
    assert len(calls) == len(gregions)
    assert got['geopotential'].dims[:2] == ('metric', 'region')
    assert list(got.coords['metric']) == sorted(ometrics)
    for mname, metric in ometrics.items():
      mi = list(got.coords['metric']).index(mname)
      for ri, region in enumerate(oregions.values()):
        fn = metric.compute if temporal_mean else metric.compute_chunk
        want = fn(forecast, truth, region=region)['geopotential']
        have = got['geopotential'].values[mi, ri]
        dims = got['geopotential'].dims[2:]
        helpers.assert_close(have, want.transpose(*dims).data, rtol=2e-6,
                             atol=1e-7)


def test_loop_with_several_land_sea_masks():
  """Regions with different 2-D weight fields (land, sea, a thresholded mask)
  in ONE Eval config: one fused pass per distinct field, results as if every
  region had been evaluated on its own (regions.py:112-158)."""
  from oracle.named import NA
  from weatherbench2_amd import config, evaluation, metrics as gm
  truth, forecast = fixtures.get_random_truth_and_forecast(
      spatial_resolution_in_degrees=10)
  forecast = fixtures.insert_nan(forecast, 0.01, seed=6)
  lat, lon = truth.coord('latitude'), truth.coord('longitude')
  rs = np.random.RandomState(1)
  frac = np.clip(rs.rand(len(lat), len(lon)) * 1.5 - 0.25, 0, 1)
  land = NA(frac, ('latitude', 'longitude'))
  sea = NA(1.0 - frac, ('latitude', 'longitude'))
  oregions = {
      'global': oreg.SliceRegion(),
      'land': oreg.LandRegion(land, lat, lon),
      'sea': oreg.LandRegion(sea, lat, lon),
      'mostly_land': oreg.LandRegion(land, lat, lon, threshold=0.5),
      'tropics_land': oreg.CombinedRegion([
          oreg.SliceRegion(lat_slice=slice(-20, 20)),
          oreg.LandRegion(land, lat, lon)]),
      'tropics': oreg.SliceRegion(lat_slice=slice(-20, 20)),
  }
  g = helpers.to_gpu_dataset
  gregions = {k: helpers.to_gpu_region(v) for k, v in oregions.items()}
  ometrics = {'mse': om.MSE(), 'bias': om.Bias()}
  gmetrics = {'mse': gm.MSE(), 'bias': gm.Bias()}
  for skipna in (False, True):
    cfg = config.Eval(metrics=gmetrics, regions=gregions)
    got = evaluation._metric_and_region_loop(g(forecast), g(truth), cfg,
                                             skipna=skipna, compute_chunk=True)
    assert list(got.coords['region']) == list(oregions)
    for mname, metric in ometrics.items():
      mi = list(got.coords['metric']).index(mname)
      for ri, (rname, region) in enumerate(oregions.items()):
        want = metric.compute_chunk(forecast, truth, region=region,
                                    skipna=skipna)
        helpers.assert_close(got['geopotential'].values[mi, ri],
                             want['geopotential'].data, rtol=1e-9, atol=1e-12,
                             err_msg=f'{rname} skipna={skipna}')
    # and one at a time through the per-metric level, inside an announcement
    with gm.fused_regions(gregions):
      for rname, region in gregions.items():
        one = gm.MSE().compute_chunk(g(forecast), g(truth), region=region,
                                     skipna=skipna)
        want = om.MSE().compute_chunk(forecast, truth, region=oregions[rname],
                                      skipna=skipna)
        helpers.assert_close(one['geopotential'].values,
                             want['geopotential'].data, rtol=1e-9, atol=1e-12)


def test_loop_computes_derived_variables_first():
  """evaluation.py:402-405: derived variables are computed on the fly and
  assigned INTO forecast / truth (the reference mutates its arguments too)."""
  from weatherbench2_amd import config, evaluation, metrics as gm
  truth, forecast = fixtures.get_random_truth_and_forecast(
      spatial_resolution_in_degrees=10)
  g = helpers.to_gpu_dataset
  gf, gt = g(forecast), g(truth)

  class Doubled:  # a user-defined DerivedVariable (duck-typed protocol)
    base_variables = ['geopotential']

    def compute(self, dataset):
      return dataset['geopotential'] * 2.0

  cfg = config.Eval(metrics={'mse': gm.MSE()},
                    derived_variables={'doubled': Doubled()})
  got = evaluation._metric_and_region_loop(gf, gt, cfg, False,
                                           compute_chunk=True)
  assert 'doubled' in gf.keys() and 'doubled' in gt.keys()
  want = om.MSE().compute_chunk(forecast, truth)['geopotential'].data
  helpers.assert_close(got['geopotential'].values[0], want, rtol=1e-9)
  helpers.assert_close(got['doubled'].values[0], 4.0 * want, rtol=1e-9)


@pytest.mark.parametrize('inits,zero_copy', [((0, 2, 4), True),
                                              ((0, 2, 3), False)])
def test_by_init_evaluation_with_device_truth_gather(inits, zero_copy):
  """evaluation.py:474-477: truth.sel(time=forecast.valid_time) then the loop,
  with device-resident arrays end to end: regular init/lead steps give an
  overlapping strided VIEW that the passes read through slab tables (no copy),
  irregular ones one index_select on the device."""
  import torch
  from weatherbench2_amd import config, evaluation, metrics as gm
  from weatherbench2_amd import xarray_lite as xl
  dev = torch.device('cuda', 0)
  rs = np.random.RandomState(4)
  t0 = np.datetime64('2020-01-01T00', 'ns')
  time = t0 + np.arange(12) * np.timedelta64(12, 'h')
  init, lead = time[list(inits)], np.arange(4) * np.timedelta64(24, 'h')
  lat, lon = np.linspace(-80, 80, 9), np.arange(16) * 22.5
  t_np = rs.standard_normal((12, 2, 9, 16)).astype(np.float32)
  f_np = rs.standard_normal((3, 4, 2, 9, 16)).astype(np.float32)
  base = {'level': np.array([500, 850]), 'latitude': lat, 'longitude': lon}
  truth = xl.Dataset({'z': xl.DataArray(torch.as_tensor(t_np).to(dev),
                                        ('time', 'level', 'latitude',
                                         'longitude'))}, {**base, 'time': time})
  forecast = xl.Dataset(
      {'z': xl.DataArray(torch.as_tensor(f_np).to(dev),
                         ('init_time', 'prediction_timedelta', 'level',
                          'latitude', 'longitude'))},
      {**base, 'init_time': init, 'prediction_timedelta': lead})
  sel = evaluation.select_truth_at_valid_time(truth, forecast)
  assert sel['z'].data.is_cuda
  shares = (sel['z'].data.untyped_storage().data_ptr()
            == truth['z'].data.untyped_storage().data_ptr())
  assert shares == zero_copy and sel['z'].data.is_contiguous() != zero_copy
  cfg = config.Eval(metrics={'mse': gm.MSE(), 'bias': gm.Bias()})
  got = evaluation._metric_and_region_loop(forecast, sel, cfg, False)
  # oracle on host copies of the same selection
  from oracle.named import DS, NA
  idx = np.array([[int(np.where(time == i + l)[0][0]) for l in lead]
                  for i in init])
  o_truth = DS({'z': NA(t_np[idx], ('init_time', 'prediction_timedelta',
                                    'level', 'latitude', 'longitude'))},
               {**base, 'init_time': init, 'prediction_timedelta': lead})
  o_fc = DS({'z': NA(f_np, ('init_time', 'prediction_timedelta', 'level',
                            'latitude', 'longitude'))},
            {**base, 'init_time': init, 'prediction_timedelta': lead})
  for name, metric in (('mse', om.MSE()), ('bias', om.Bias())):
    mi = list(got.coords['metric']).index(name)  # merged labels are sorted
    want = metric.compute(o_fc, o_truth)['z']
    dims = got['z'].dims[1:]
    helpers.assert_close(got['z'].values[mi], want.transpose(*dims).data,
                         rtol=1e-9, atol=1e-12)


def test_caches_notice_in_place_updates_of_a_reused_buffer():
  """A pipeline that refills ONE device buffer per chunk must not get the
  previous chunk's cached result (the caches key on the tensor's version)."""
  import torch
  from weatherbench2_amd import metrics as gm
  from weatherbench2_amd import xarray_lite as xl
  dev = torch.device('cuda', 0)
  lat, lon = np.linspace(-80, 80, 9), np.arange(16) * 22.5
  coords = {'time': np.arange(2), 'latitude': lat, 'longitude': lon}
  dims = ('time', 'latitude', 'longitude')
  gen = torch.Generator(device=dev).manual_seed(0)
  f = torch.randn((2, 9, 16), device=dev, generator=gen)
  t = torch.randn((2, 9, 16), device=dev, generator=gen)
  forecast = xl.Dataset({'z': xl.DataArray(f, dims)}, coords)
  truth = xl.Dataset({'z': xl.DataArray(t, dims)}, coords)
  first = gm.Bias().compute_chunk(forecast, truth)['z'].values.copy()
  again = gm.Bias().compute_chunk(forecast, truth)['z'].values
  np.testing.assert_array_equal(first, again)          # cache hit
  f.add_(1.0)                                          # refill in place
  shifted = gm.Bias().compute_chunk(forecast, truth)['z'].values
  np.testing.assert_allclose(shifted, first + 1.0, rtol=1e-6)
  # host arrays: a changed buffer is noticed through the sampled fingerprint
  fh, th = f.cpu().numpy().copy(), t.cpu().numpy().copy()
  fo = xl.Dataset({'z': xl.DataArray(fh, dims)}, coords)
  to = xl.Dataset({'z': xl.DataArray(th, dims)}, coords)
  b0 = gm.Bias().compute_chunk(fo, to)['z'].values.copy()
  fh += 2.0
  b1 = gm.Bias().compute_chunk(fo, to)['z'].values
  np.testing.assert_allclose(b1, b0 + 2.0, rtol=1e-6)


def test_chunk_feeder_and_staged_upload_move_the_exact_bytes():
  """feeder.upload (pinned slices on the copy stream, used for NumPy inputs of
  the metrics) and feeder.ChunkFeeder (double-buffered chunk stream from NumPy
  or pinned tensors): what arrives in HBM is byte-identical, chunk after chunk,
  while passes over earlier chunks are still queued."""
  import torch
  from weatherbench2_amd import engine, feeder
  dev = torch.device('cuda')
  rs = np.random.RandomState(0)
  big = rs.standard_normal((3, 721, 1440)).astype(np.float32)   # > one 64 MB... 12 MB
  huge = rs.standard_normal((40, 721, 1440)).astype(np.float32)  # 166 MB: 3 slices
  for arr in (big, huge, rs.standard_normal((5, 7)), np.arange(10, dtype=np.int64)):
    got = feeder.upload(np.ascontiguousarray(arr), dev)
    torch.cuda.synchronize()
    assert got.dtype == torch.from_numpy(arr[:0].copy()).dtype
    np.testing.assert_array_equal(got.cpu().numpy(), arr)
  via_engine = engine.as_device_tensor(big, dev)
  np.testing.assert_array_equal(via_engine.cpu().numpy(), big)
  chunks = [rs.standard_normal((2, 64, 128)).astype(np.float32)
            for _ in range(5)]
  fd = feeder.ChunkFeeder((2, 64, 128), torch.float32, dev, depth=2)
  sums = []
  fd.submit(chunks[0])
  for i in range(len(chunks)):
    if i + 1 < len(chunks):
      src = chunks[i + 1]
      if i % 2:  # alternate NumPy / pinned-tensor sources
        src = torch.from_numpy(src).pin_memory()
      fd.submit(src)
    x = fd.acquire()
    sums.append(x.double().sum())   # queued work on the compute stream
    got = x.clone()
    fd.release()
    np.testing.assert_array_equal(got.cpu().numpy(), chunks[i])
  for s, c in zip(sums, chunks):
    np.testing.assert_allclose(float(s), c.astype(np.float64).sum(), rtol=1e-12)


def test_resident_climatology_and_truth_give_identical_results():
  """evaluation.make_resident: climatology and truth uploaded
  once, gathered on the device; the results are bit-identical to the same
  evaluation from host arrays, the resident variables are device tensors and
  the coordinates stay host labels."""
  import torch
  from weatherbench2_amd import config, evaluation, metrics as gm
  truth, forecast = fixtures.get_random_truth_and_forecast(
      variables=('geopotential', 'temperature'), lead_stop='1 day')
  kw = dict(variables_3d=list(truth.keys()), variables_2d=[],
            spatial_resolution_in_degrees=180 / (
                len(truth.coord('latitude')) - 1),
            levels=tuple(truth.coord('level')))
  clim = fixtures.random_like(
      fixtures.mock_hourly_climatology_data(hour_interval=3, **kw), seed=7)
  cast = lambda ds: ds.copy(data={k: v.data.astype(np.float32)
                                  for k, v in ds.items()})
  truth, forecast, clim = cast(truth), cast(forecast), cast(clim)
  g = helpers.to_gpu_dataset
  host_clim, host_truth = g(clim), g(truth)
  res_clim = evaluation.make_resident(host_clim)
  res_truth = evaluation.make_resident(host_truth, 'cuda')
  for ds, host in ((res_clim, host_clim), (res_truth, host_truth)):
    assert sorted(ds.keys()) == sorted(host.keys())
    for k in ds.keys():
      assert isinstance(ds[k].data, torch.Tensor) and ds[k].data.is_cuda
      assert ds[k].dims == host[k].dims
      np.testing.assert_array_equal(ds[k].values, host[k].values)
    from weatherbench2_amd import xarray_lite as xl
    assert all(not isinstance(c.data if isinstance(c, xl.DataArray) else c,
                              torch.Tensor) for c in ds.coords.values())
  regions = {'global': None,
             'tropics': helpers.to_gpu_region(
                 oreg.SliceRegion(lat_slice=slice(-20, 20)))}

  def run(c, t_):
    cfg = config.Eval(metrics={'acc': gm.ACC(climatology=c), 'mse': gm.MSE()},
                      regions=regions)
    return evaluation._metric_and_region_loop(g(forecast), t_, cfg, False,
                                              compute_chunk=True)
  want = run(host_clim, host_truth)
  got = run(res_clim, res_truth)
  for k in want.keys():
    assert got[k].dims == want[k].dims
    np.testing.assert_array_equal(np.asarray(got[k].values),
                                  np.asarray(want[k].values))
