"""GPU parity of the threshold-metric family (-m gpu): reference known answers
(metrics_test.py:366-535, 985-1030, 1290-1388) and random data vs the oracle."""
import numpy as np
import pytest

from oracle import fixtures
from oracle import metrics_np as om
from oracle import regions_np as oreg
from oracle import thresholds_np as oth
from oracle.named import DS, NA
from tests import helpers
from tests.test_oracle_thresholds import (ensemble_case, gaussian_case,
                                          rps_climatology, _clim_like, _merge,
                                          KW)

pytestmark = pytest.mark.gpu

g = helpers.to_gpu_dataset


def _gth(cls_name, climatology, quantile):
  from weatherbench2_amd import thresholds as gth
  return getattr(gth, cls_name)(climatology=g(climatology), quantile=quantile)


@pytest.mark.parametrize('error,e1,e2', [(0.02, 0.04421, 0.257883),
                                         (1e6, 0.70786, 0.707861)])
def test_gaussian_brier_known_answers(error, e1, e2):
  from weatherbench2_amd import metrics as gm
  forecast, truth, clim, qclim = gaussian_case(error)
  th = _gth('GaussianQuantileThreshold', clim, 0.8)
  res = gm.GaussianBrierScore(thresholds=[th]).compute(g(forecast), g(truth))
  np.testing.assert_allclose(res['2m_temperature'].values, [[e1, e1]],
                             rtol=1e-4)
  assert res.attrs['threshold_method'] == 'GaussianQuantileThreshold'
  th = _gth('QuantileThreshold', qclim, 0.8)
  res = gm.GaussianBrierScore(thresholds=[th]).compute(g(forecast), g(truth))
  np.testing.assert_allclose(res['2m_temperature'].values, [[e2, e2]],
                             rtol=1e-4)


@pytest.mark.parametrize('error,expected', [(0.02, 0.236055), (1e6, 1.841019)])
def test_gaussian_ignorance_known_answers(error, expected):
  from weatherbench2_amd import metrics as gm
  forecast, truth, clim, _ = gaussian_case(error)
  th = _gth('GaussianQuantileThreshold', clim, 0.8)
  res = gm.GaussianIgnoranceScore(thresholds=[th]).compute(g(forecast),
                                                           g(truth))
  np.testing.assert_allclose(res['2m_temperature'].values,
                             [[expected, expected]], rtol=1e-4)


@pytest.mark.parametrize('error,expected', [(0.02, 0.295746), (1e6, 0.758203)])
def test_gaussian_rps_known_answers(error, expected):
  from weatherbench2_amd import metrics as gm
  forecast = fixtures.mock_forecast_data(
      variables_2d=['2m_temperature', '2m_temperature_std'], lead_stop='1 day',
      **KW)
  truth0 = fixtures.mock_truth_data(variables_2d=['2m_temperature'], **KW)
  clim = rps_climatology(truth0)
  truth = truth0 + 1.0
  forecast = forecast + 1.0 + error
  ths = [_gth('QuantileThreshold', clim, q) for q in (0.33, 0.66, 1.0)]
  res = gm.GaussianRPS(thresholds=ths).compute(g(forecast), g(truth))
  np.testing.assert_allclose(res['2m_temperature'].values,
                             [expected, expected], rtol=1e-4)


@pytest.mark.parametrize('error,ens_delta,expected',
                         [(0.0, 0.1, 0.0), (0.0, 1.0, 0.25), (-10.0, 0.1, 1.0)])
def test_ensemble_brier_known_answers(error, ens_delta, expected):
  from weatherbench2_amd import metrics as gm
  forecast, truth, clim = ensemble_case(error, ens_delta)
  th = _gth('GaussianQuantileThreshold', clim, 0.2)
  res = gm.EnsembleBrierScore(thresholds=[th]).compute(g(forecast), g(truth))
  np.testing.assert_allclose(res['2m_temperature'].values,
                             [[expected, expected]], rtol=1e-4, atol=1e-12)
  assert res.attrs['ensemble_size'] == 4


@pytest.mark.parametrize('skipna', [True, False])
def test_brier_nan_propagates_to_output_unless_skipna(skipna):
  # metrics_test.py:1035-1108
  from tests.test_oracle_thresholds import nan_cases
  from weatherbench2_amd import metrics as gm
  forecast, truth, clim, forecast_nan, truth_nan = nan_cases()
  th = _gth('GaussianQuantileThreshold', clim, 0.2)
  expected = [[0.0, 0.0]] if skipna else [[np.nan, np.nan]]
  for f, t in ((forecast_nan, truth), (forecast, truth_nan)):
    res = gm.EnsembleBrierScore(thresholds=[th]).compute(g(f), g(t),
                                                        skipna=skipna)
    np.testing.assert_allclose(res['2m_temperature'].values, expected,
                               atol=1e-12)


@pytest.mark.parametrize('error,expected', [(0.0, 0.0), (-10.0, np.inf)])
def test_ensemble_ignorance_known_answers(error, expected):
  from weatherbench2_amd import metrics as gm
  forecast, truth, clim = ensemble_case(error, 0.0)
  th = _gth('GaussianQuantileThreshold', clim, 0.2)
  res = gm.EnsembleIgnoranceScore(thresholds=[th]).compute(g(forecast),
                                                           g(truth))
  np.testing.assert_allclose(res['2m_temperature'].values,
                             [[expected, expected]], rtol=1e-4)


@pytest.mark.parametrize('error,expected', [(0.02, 0.0), (-2.0, 2.0)])
def test_ensemble_rps_known_answers(error, expected):
  from weatherbench2_amd import metrics as gm
  kw = dict(variables_2d=['2m_temperature'], **KW)
  forecast = fixtures.mock_forecast_data(ensemble_size=4, lead_stop='1 day',
                                         **kw)
  truth0 = fixtures.mock_truth_data(**kw)
  clim = rps_climatology(truth0)
  truth = truth0 + 1.5
  forecast = forecast + 1.0 + error
  ths = [_gth('QuantileThreshold', clim, q) for q in (0.33, 0.66, 1.0)]
  res = gm.EnsembleRPS(thresholds=ths).compute(g(forecast), g(truth))
  np.testing.assert_allclose(res['2m_temperature'].values,
                             [expected, expected], rtol=1e-4, atol=1e-12)


def _random_case(ensemble_size=None, nan_frac=0.0):
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=ensemble_size, spatial_resolution_in_degrees=10,
      lead_stop='2 day')
  doy = {'dayofyear': 1 + np.arange(366)}
  zero = truth.zeros_like()
  clim = _merge(_clim_like(zero + 0.1, doy),
                _clim_like(zero + 0.9, doy,
                           {'geopotential': 'geopotential_std'}))
  if nan_frac:
    forecast = fixtures.insert_nan(forecast, nan_frac, seed=9)
    truth = fixtures.insert_nan(truth, nan_frac / 2, seed=10)
  return truth, forecast, clim


@pytest.mark.parametrize('skipna', [False, True])
def test_gaussian_family_random_vs_oracle(skipna):
  from weatherbench2_amd import metrics as gm
  truth, mean, clim = _random_case()
  _, std = fixtures.get_random_truth_and_forecast(
      spatial_resolution_in_degrees=10, lead_stop='2 day', seed=5)
  dims = mean['geopotential'].dims
  forecast = DS({'geopotential': mean['geopotential'],
                 'geopotential_std': NA(np.abs(std['geopotential'].data) + 0.2,
                                        dims)}, mean.coords)
  if skipna:
    forecast = fixtures.insert_nan(forecast, 0.05, seed=3)
  oths = [oth.GaussianQuantileThreshold(clim, q) for q in (0.3, 0.8)]
  gths = [_gth('GaussianQuantileThreshold', clim, q) for q in (0.3, 0.8)]
  region = oreg.SliceRegion(lat_slice=slice(-40, 50))
  for name in ('GaussianBrierScore', 'GaussianIgnoranceScore', 'GaussianRPS'):
    want = getattr(om, name)(thresholds=oths).compute_chunk(
        forecast, truth, region=region, skipna=skipna)
    got = getattr(gm, name)(thresholds=gths).compute_chunk(
        g(forecast), g(truth), region=helpers.to_gpu_region(region),
        skipna=skipna)
    assert got['geopotential'].dims == want['geopotential'].dims, name
    helpers.assert_close(got['geopotential'].values, want['geopotential'].data,
                         rtol=1e-9, atol=1e-12, err_msg=name)


@pytest.mark.parametrize('skipna', [False, True])
@pytest.mark.parametrize('ensemble_size', [1, 2, 7, 33])
def test_ensemble_family_random_vs_oracle(ensemble_size, skipna):
  from weatherbench2_amd import metrics as gm
  truth, forecast, clim = _random_case(ensemble_size,
                                       nan_frac=0.05 if skipna else 0.0)
  oths = [oth.GaussianQuantileThreshold(clim, q) for q in (0.2, 0.6)]
  gths = [_gth('GaussianQuantileThreshold', clim, q) for q in (0.2, 0.6)]
  for name in ('EnsembleBrierScore', 'DebiasedEnsembleBrierScore',
               'EnsembleIgnoranceScore', 'EnsembleRPS'):
    want = getattr(om, name)(thresholds=oths).compute_chunk(
        forecast, truth, skipna=skipna)
    got = getattr(gm, name)(thresholds=gths).compute_chunk(
        g(forecast), g(truth), skipna=skipna)
    assert got['geopotential'].dims == want['geopotential'].dims, name
    helpers.assert_close(got['geopotential'].values, want['geopotential'].data,
                         rtol=1e-9, atol=1e-12, err_msg=name)


def _to_gpu_with_valid_time(ds):
  from weatherbench2_amd import xarray_lite as xl
  out = g(ds)
  return out


def test_seeps_known_answers_and_random():
  # metrics_test.py:1391-1436 + random precipitation vs the oracle
  from tests.test_oracle_thresholds import seeps_case
  from weatherbench2_amd import metrics as gm
  forecast, truth, climatology = seeps_case()
  seeps = gm.SEEPS(climatology=g(climatology))
  r1 = seeps.compute(g(forecast), g(truth))
  np.testing.assert_allclose(r1['total_precipitation_24hr'].values, 0,
                             atol=1e-4)
  r2 = seeps.compute(g(forecast + 0.5), g(truth))
  np.testing.assert_allclose(r2['total_precipitation_24hr'].values, 1.25,
                             atol=1e-4)
  # random fields, a varying dry fraction (some points masked out), NaNs
  rs = np.random.RandomState(0)
  name = 'total_precipitation_24hr'
  rand = lambda shape: (rs.gamma(0.3, 2.0, size=shape) * 1e-2).astype(
      np.float32)
  f2 = DS({name: NA(rand(forecast[name].shape), forecast[name].dims)},
          forecast.coords)
  t2 = DS({name: NA(rand(truth[name].shape), truth[name].dims)}, truth.coords)
  f2 = fixtures.insert_nan(f2, 0.02, seed=1)
  f2 = f2.copy(data={name: f2[name].data.astype(np.float32)})
  base = climatology[name]
  frac = rs.uniform(0.0, 1.0, size=base.shape[2:]).astype(np.float32)
  thr = (rs.uniform(0.002, 0.02, size=base.shape)).astype(np.float32)
  clim2 = DS({name: base,
              name + '_seeps_dry_fraction': NA(
                  np.broadcast_to(frac, base.shape).copy(), base.dims),
              name + '_seeps_threshold': NA(thr, base.dims)},
             climatology.coords)
  want = om.SEEPS(climatology=clim2).compute_chunk(f2, t2)
  got = gm.SEEPS(climatology=g(clim2)).compute_chunk(g(f2), g(t2))
  assert got[name].dims == want[name].dims
  helpers.assert_close(got[name].values, want[name].data, rtol=1e-9,
                       atol=1e-12)


def _dev_values(da):
  v = da.data
  return v.cpu().numpy() if hasattr(v, 'cpu') else np.asarray(v)


@pytest.mark.parametrize('skipna', [False, True])
@pytest.mark.parametrize('ensemble_size', [1, 5, 33])
def test_spatial_ensemble_threshold_maps_vs_oracle(ensemble_size, skipna):
  # SpatialEnsembleBrierScore & co (metrics.py:1615-1638, 1697-1719,
  # 1780-1802, 1870-1891): the unreduced maps, and their device temporal mean
  from weatherbench2_amd import metrics as gm
  truth, forecast, clim = _random_case(ensemble_size,
                                       nan_frac=0.05 if skipna else 0.0)
  oths = [oth.GaussianQuantileThreshold(clim, q) for q in (0.2, 0.6)]
  gths = [_gth('GaussianQuantileThreshold', clim, q) for q in (0.2, 0.6)]
  for name in ('SpatialEnsembleBrierScore', 'SpatialDebiasedEnsembleBrierScore',
               'SpatialEnsembleIgnoranceScore', 'SpatialEnsembleRPS'):
    want = getattr(om, name)(thresholds=oths).compute_chunk(
        forecast, truth, skipna=skipna)['geopotential']
    metric = getattr(gm, name)(thresholds=gths)
    got = metric.compute_chunk(g(forecast), g(truth), skipna=skipna)
    da = got['geopotential']
    assert set(da.dims) == set(want.dims), name
    assert got.attrs['threshold_method'] == 'GaussianQuantileThreshold'
    w = want.transpose(*da.dims).data
    a = _dev_values(da)
    assert a.dtype == np.float64
    helpers.assert_close(a, w, rtol=1e-12, atol=1e-14, err_msg=name)
    mean = metric.compute(g(forecast), g(truth), skipna=skipna)
    dm = mean['geopotential']
    assert mean.attrs['ensemble_size'] == ensemble_size
    axis = da.dims.index('time')
    with np.errstate(all='ignore'):
      import warnings
      with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        wm = np.nanmean(w, axis) if skipna else np.mean(w, axis)
    assert dm.dims == tuple(d for d in da.dims if d != 'time')
    helpers.assert_close(_dev_values(dm), wm, rtol=1e-12, atol=1e-14,
                         err_msg=name + ' (mean)')


def test_spatial_seeps_vs_oracle():
  from tests.test_oracle_thresholds import seeps_case
  from weatherbench2_amd import metrics as gm
  forecast, truth, climatology = seeps_case()
  name = 'total_precipitation_24hr'
  rs = np.random.RandomState(3)
  rand = lambda shape: (rs.gamma(0.3, 2.0, size=shape) * 1e-2).astype(
      np.float32)
  f2 = DS({name: NA(rand(forecast[name].shape), forecast[name].dims)},
          forecast.coords)
  t2 = DS({name: NA(rand(truth[name].shape), truth[name].dims)}, truth.coords)
  f2 = fixtures.insert_nan(f2, 0.02, seed=1)
  f2 = f2.copy(data={name: f2[name].data.astype(np.float32)})
  base = climatology[name]
  frac = rs.uniform(0.0, 1.0, size=base.shape[2:]).astype(np.float32)
  thr = (rs.uniform(0.002, 0.02, size=base.shape)).astype(np.float32)
  clim2 = DS({name: base,
              name + '_seeps_dry_fraction': NA(
                  np.broadcast_to(frac, base.shape).copy(), base.dims),
              name + '_seeps_threshold': NA(thr, base.dims)},
             climatology.coords)
  want = om.SpatialSEEPS(climatology=clim2).compute_chunk(f2, t2)[name]
  metric = gm.SpatialSEEPS(climatology=g(clim2))
  got = metric.compute_chunk(g(f2), g(t2))[name]
  assert set(got.dims) == set(want.dims)
  w = want.transpose(*got.dims).data
  a = _dev_values(got)
  assert np.isnan(a).any() and np.isfinite(a).any()
  helpers.assert_close(a, w, rtol=1e-6 if w.dtype == np.float32 else 1e-12,
                       atol=1e-14)
  # known answers through the map: perfect forecast -> 0 wherever defined
  m0 = _dev_values(gm.SpatialSEEPS(climatology=g(climatology)).compute_chunk(
      g(forecast), g(truth))[name])
  assert np.nanmax(np.abs(m0)) < 1e-4
  # temporal mean with skipna, accumulated on the device
  mean = metric.compute(g(f2), g(t2), skipna=True)[name]
  axis = got.dims.index('time' if 'time' in got.dims else 'init_time')
  import warnings
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    wm = np.nanmean(w, axis)
  helpers.assert_close(_dev_values(mean), wm, rtol=1e-6, atol=1e-14)


def test_integral_of_debiased_brier_score_is_crps():
  # metrics_test.py:1207-1288 on the GPU path: 200 thresholds x CRPS
  from scipy import stats
  from tests.test_oracle_thresholds import (brier_integral_case,
                                            trapezoid_over_thresholds)
  from weatherbench2_amd import metrics as gm
  truth, forecast, clim, quantiles = brier_integral_case()
  ths = [_gth('GaussianQuantileThreshold', clim, q) for q in quantiles]
  bs = gm.DebiasedEnsembleBrierScore(thresholds=ths).compute(
      g(forecast), g(truth))['geopotential']
  assert bs.dims[0] == 'quantile'
  integral = trapezoid_over_thresholds(np.asarray(bs.values),
                                       stats.norm.ppf(quantiles))
  crps = gm.CRPS().compute(g(forecast), g(truth))['geopotential']
  want = np.asarray(crps.transpose(*bs.dims[1:]).values)
  np.testing.assert_allclose(integral, want, rtol=10 / len(quantiles))
