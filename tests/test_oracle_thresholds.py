"""Pins the oracle's threshold-metric family against the reference's own
known-answer tests (metrics_test.py:366-535, 985-1030, 1290-1388).  CPU only."""
import numpy as np
import pytest

from oracle import fixtures
from oracle import metrics_np as metrics
from oracle import thresholds_np as thresholds
from oracle.named import DS, NA

KW = dict(variables_3d=[], time_start='2022-01-01', time_stop='2022-01-02')


def _clim_like(truth, expand, rename=None, add=0.0):
  """truth.isel(time=0, drop=True).expand_dims(...).rename(...)"""
  ds = (truth + add).isel(time=0)
  for dim, coord in reversed(list(expand.items())):
    ds = ds.expand_dims(dim, coord=coord)
  return ds.rename_vars(rename) if rename else ds


def _merge(*dss):
  out, coords = {}, {}
  for d in dss:
    out.update(d.vars)
    coords.update(d.coords)
  return DS(out, coords)


def gaussian_case(error):
  forecast = fixtures.mock_forecast_data(
      variables_2d=['2m_temperature', '2m_temperature_std'], lead_stop='1 day',
      **KW)
  truth = fixtures.mock_truth_data(variables_2d=['2m_temperature'], **KW) + 1.0
  forecast = forecast + 1.0 + error
  doy = {'dayofyear': 1 + np.arange(366)}
  clim = _merge(_clim_like(truth, doy),
                _clim_like(truth, doy,
                           {'2m_temperature': '2m_temperature_std'}))
  qclim = _clim_like(truth, {'quantile': np.array([0.8]), **doy},
                     {'2m_temperature': '2m_temperature_quantile'})
  return forecast, truth, clim, qclim


@pytest.mark.parametrize('error,e1,e2', [(0.02, 0.04421, 0.257883),
                                         (1e6, 0.70786, 0.707861)])
def test_gaussian_brier_score(error, e1, e2):
  forecast, truth, clim, qclim = gaussian_case(error)
  th = thresholds.GaussianQuantileThreshold(climatology=clim, quantile=0.8)
  res = metrics.GaussianBrierScore(thresholds=[th]).compute(forecast, truth)
  np.testing.assert_allclose(res['2m_temperature'].data, [[e1, e1]], rtol=1e-4)
  th = thresholds.QuantileThreshold(climatology=qclim, quantile=0.8)
  res = metrics.GaussianBrierScore(thresholds=[th]).compute(forecast, truth)
  np.testing.assert_allclose(res['2m_temperature'].data, [[e2, e2]], rtol=1e-4)


@pytest.mark.parametrize('error,expected', [(0.02, 0.236055), (1e6, 1.841019)])
def test_gaussian_ignorance_score(error, expected):
  forecast, truth, clim, _ = gaussian_case(error)
  th = thresholds.GaussianQuantileThreshold(climatology=clim, quantile=0.8)
  res = metrics.GaussianIgnoranceScore(thresholds=[th]).compute(forecast, truth)
  np.testing.assert_allclose(res['2m_temperature'].data,
                             [[expected, expected]], rtol=1e-4)


def rps_climatology(truth0):
  doy = {'dayofyear': 1 + np.arange(366)}
  parts = []
  for q, add in ((0.33, 0.0), (0.66, 1.0), (1.0, 2.0)):
    parts.append(_clim_like(truth0, {'quantile': np.array([q]), **doy},
                            {'2m_temperature': '2m_temperature_quantile'}, add))
  data = np.concatenate([p['2m_temperature_quantile'].data for p in parts], 0)
  coords = dict(parts[0].coords)
  coords['quantile'] = np.array([0.33, 0.66, 1.0])
  return DS({'2m_temperature_quantile':
             NA(data, parts[0]['2m_temperature_quantile'].dims)}, coords)


@pytest.mark.parametrize('error,expected', [(0.02, 0.295746), (1e6, 0.758203)])
def test_gaussian_rps(error, expected):
  forecast = fixtures.mock_forecast_data(
      variables_2d=['2m_temperature', '2m_temperature_std'], lead_stop='1 day',
      **KW)
  truth0 = fixtures.mock_truth_data(variables_2d=['2m_temperature'], **KW)
  clim = rps_climatology(truth0)
  truth = truth0 + 1.0
  forecast = forecast + 1.0 + error
  ths = [thresholds.QuantileThreshold(climatology=clim, quantile=q)
         for q in (0.33, 0.66, 1.0)]
  res = metrics.GaussianRPS(thresholds=ths).compute(forecast, truth)
  np.testing.assert_allclose(res['2m_temperature'].data, [expected, expected],
                             rtol=1e-4)


def ensemble_case(error, ens_delta):
  kw = dict(variables_2d=['2m_temperature'], **KW)
  forecast = fixtures.mock_forecast_data(ensemble_size=4, lead_stop='1 day',
                                         **kw)
  truth = fixtures.mock_truth_data(**kw) + 1.0
  forecast = forecast + 1.0 + error
  delta = NA(ens_delta * np.arange(-2, 2).astype(float), ('realization',))
  forecast = forecast + DS({'2m_temperature': delta})
  doy = {'dayofyear': 1 + np.arange(366)}
  clim = _merge(_clim_like(truth, doy),
                _clim_like(truth, doy,
                           {'2m_temperature': '2m_temperature_std'}))
  return forecast, truth, clim


@pytest.mark.parametrize('error,ens_delta,expected',
                         [(0.0, 0.1, 0.0), (0.0, 1.0, 0.25), (-10.0, 0.1, 1.0)])
def test_ensemble_brier_score(error, ens_delta, expected):
  forecast, truth, clim = ensemble_case(error, ens_delta)
  th = thresholds.GaussianQuantileThreshold(climatology=clim, quantile=0.2)
  res = metrics.EnsembleBrierScore(thresholds=[th]).compute(forecast, truth)
  np.testing.assert_allclose(res['2m_temperature'].data,
                             [[expected, expected]], rtol=1e-4, atol=1e-12)


def nan_cases():
  """metrics_test.py:1035-1075: the score-0 case with NaN along the first
  latitude of the forecast / the first longitude of the truth."""
  forecast, truth, clim = ensemble_case(0.0, 0.1)
  name = '2m_temperature'

  def blank(ds, dim):
    da = ds[name]
    data = da.data.copy()
    sl = [slice(None)] * data.ndim
    sl[da.dims.index(dim)] = 0
    data[tuple(sl)] = np.nan
    return ds.copy(data={name: data})
  return (forecast, truth, clim, blank(forecast, 'latitude'),
          blank(truth, 'longitude'))


@pytest.mark.parametrize('skipna', [True, False])
def test_brier_nan_propagates_to_output_unless_skipna(skipna):
  forecast, truth, clim, forecast_nan, truth_nan = nan_cases()
  th = thresholds.GaussianQuantileThreshold(climatology=clim, quantile=0.2)
  expected = [[0.0, 0.0]] if skipna else [[np.nan, np.nan]]
  for f, t in ((forecast_nan, truth), (forecast, truth_nan)):
    res = metrics.EnsembleBrierScore(thresholds=[th]).compute(
        f, t, skipna=skipna)
    np.testing.assert_allclose(res['2m_temperature'].data, expected,
                               atol=1e-12)


@pytest.mark.parametrize('error,expected', [(0.0, 0.0), (-10.0, np.inf)])
def test_ensemble_ignorance_score(error, expected):
  forecast, truth, clim = ensemble_case(error, 0.0)
  th = thresholds.GaussianQuantileThreshold(climatology=clim, quantile=0.2)
  res = metrics.EnsembleIgnoranceScore(thresholds=[th]).compute(forecast, truth)
  np.testing.assert_allclose(res['2m_temperature'].data,
                             [[expected, expected]], rtol=1e-4)


@pytest.mark.parametrize('error,expected', [(0.02, 0.0), (-2.0, 2.0)])
def test_ensemble_rps(error, expected):
  kw = dict(variables_2d=['2m_temperature'], **KW)
  forecast = fixtures.mock_forecast_data(ensemble_size=4, lead_stop='1 day',
                                         **kw)
  truth0 = fixtures.mock_truth_data(**kw)
  clim = rps_climatology(truth0)
  truth = truth0 + 1.5
  forecast = forecast + 1.0 + error
  ths = [thresholds.QuantileThreshold(climatology=clim, quantile=q)
         for q in (0.33, 0.66, 1.0)]
  res = metrics.EnsembleRPS(thresholds=ths).compute(forecast, truth)
  np.testing.assert_allclose(res['2m_temperature'].data, [expected, expected],
                             rtol=1e-4, atol=1e-12)


def test_debiased_brier_matches_large_ensemble_in_expectation():
  # metrics_test.py:1111-1170 (statistical): debiasing a 2-member Brier score
  # approaches the 500-member score.
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=500, spatial_resolution_in_degrees=20)
  doy = {'dayofyear': 1 + np.arange(366)}
  zero = truth.zeros_like()
  clim = _merge(_clim_like(zero, doy),
                _clim_like(zero + 1.0, doy, {'geopotential': 'geopotential_std'}))
  th = thresholds.GaussianQuantileThreshold(climatology=clim, quantile=0.7)
  small = forecast.isel(realization=slice(2))
  big = metrics.EnsembleBrierScore(thresholds=[th]).compute_chunk(forecast, truth)
  biased = metrics.EnsembleBrierScore(thresholds=[th]).compute_chunk(small, truth)
  debiased = metrics.DebiasedEnsembleBrierScore(thresholds=[th]).compute_chunk(
      small, truth)
  b, s, d = (x['geopotential'].data.mean() for x in (big, biased, debiased))
  assert s - b > 0.05          # the small ensemble is visibly biased
  assert abs(d - b) < 0.02     # and debiasing removes it


def seeps_case():
  """metrics_test.py:1393-1421: by-init forecast with a 2-D valid_time."""
  kw = dict(variables_3d=[], variables_2d=['total_precipitation_24hr'],
            time_start='2022-01-01', time_stop='2022-01-11')
  forecast = fixtures.mock_forecast_data(lead_stop='0 day', **kw)
  fv = forecast['total_precipitation_24hr']
  dims = tuple('init_time' if d == 'time' else d for d in fv.dims)
  coords = dict(forecast.coords)
  coords['init_time'] = coords.pop('time')
  lead = coords['prediction_timedelta']
  coords['valid_time'] = NA(coords['init_time'][:, None] + lead[None, :],
                            ('init_time', 'prediction_timedelta'))
  forecast = DS({'total_precipitation_24hr': NA(fv.data, dims)}, coords)
  truth = fixtures.mock_truth_data(**kw)
  # truth.sel(time=forecast.valid_time): dims (init_time, prediction_timedelta, ...)
  tv = truth['total_precipitation_24hr']
  tl = NA(tv.data[:, None], ('init_time', 'prediction_timedelta') + tv.dims[1:])
  tcoords = {k: v for k, v in coords.items()}
  truth_like_forecast = DS({'total_precipitation_24hr': tl}, tcoords)
  clim0 = truth.isel(time=0)
  base = clim0['total_precipitation_24hr']
  mk = lambda add: NA(np.broadcast_to((base.data + add)[None, None],
                                      (4, 366) + base.shape).copy(),
                      ('hour', 'dayofyear') + base.dims)
  ccoords = {k: v for k, v in clim0.coords.items()}
  ccoords['hour'] = np.arange(0, 24, 6)
  ccoords['dayofyear'] = 1 + np.arange(366)
  climatology = DS({
      'total_precipitation_24hr': mk(0.0),
      'total_precipitation_24hr_seeps_dry_fraction': mk(np.float32(0.4)),
      'total_precipitation_24hr_seeps_threshold': mk(np.float32(1.0))},
      ccoords)
  return forecast, truth_like_forecast, climatology


def test_seeps_expected_values():
  forecast, truth, climatology = seeps_case()
  seeps = metrics.SEEPS(climatology=climatology)
  r1 = seeps.compute(forecast, truth)
  np.testing.assert_allclose(r1['total_precipitation_24hr'].data, 0, atol=1e-4)
  r2 = seeps.compute(forecast + 0.5, truth)
  np.testing.assert_allclose(r2['total_precipitation_24hr'].data, 1.25,
                             atol=1e-4)


def brier_integral_case(n_quantiles=200):
  """metrics_test.py:1207-1260: 2 members, level-dependent bias, thresholds =
  standard-normal quantiles (constant in space/time)."""
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=2, spatial_resolution_in_degrees=60,
      time_start='2019-01-01', time_stop='2019-01-04',
      time_resolution='12 hours', lead_start='0 day', lead_stop='0 day',
      levels=[500, 700, 850])
  fv = forecast['geopotential']
  shift = np.array([-1, 0, 1], dtype=fv.data.dtype).reshape(
      [3 if d == 'level' else 1 for d in fv.dims])
  forecast = forecast.copy(data={
      'geopotential': fv.data + np.abs(fv.data) ** 0.2 + shift})
  doy = {'dayofyear': 1 + np.arange(366)}
  zero = truth.zeros_like()
  clim = _merge(_clim_like(zero, doy),
                _clim_like(zero + 1.0, doy,
                           {'geopotential': 'geopotential_std'}))
  quantiles = np.linspace(0, 1, num=n_quantiles + 2)[1:-1]
  return truth, forecast, clim, quantiles


def trapezoid_over_thresholds(bs, values):
  """bs[quantile, ...] integrated over the threshold VALUES (xarray's
  .integrate('threshold'): trapezoidal rule)."""
  dv = np.diff(values).reshape((-1,) + (1,) * (bs.ndim - 1))
  return ((bs[1:] + bs[:-1]) * 0.5 * dv).sum(0)


def test_integral_of_debiased_brier_score_is_crps():
  # metrics_test.py:1207-1288
  from scipy import stats
  truth, forecast, clim, quantiles = brier_integral_case()
  ths = [thresholds.GaussianQuantileThreshold(climatology=clim, quantile=q)
         for q in quantiles]
  bs = metrics.DebiasedEnsembleBrierScore(thresholds=ths).compute(
      forecast, truth)['geopotential']
  assert bs.dims[0] == 'quantile'
  integral = trapezoid_over_thresholds(bs.data, stats.norm.ppf(quantiles))
  crps = metrics.CRPS().compute(forecast, truth)['geopotential']
  want = crps.transpose(*bs.dims[1:]).data
  np.testing.assert_allclose(integral, want, rtol=10 / len(quantiles))
