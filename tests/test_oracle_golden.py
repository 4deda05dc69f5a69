"""Pins the NumPy oracle against the reference's own known-answer tests.

Every test here is a port of a test in /root/reference/weatherbench2/
(metrics_test.py, regions_test.py, derived_variables_test.py); the expected
numbers are the reference's.  CPU only.
"""
import numpy as np
import pytest
from scipy import stats

from oracle import fixtures
from oracle import metrics_np as metrics
from oracle import regions_np as regions
from oracle import spectrum_np
from oracle.named import DS, NA


def test_get_lat_weights():
  # metrics_test.py:63-82
  w = metrics.get_lat_weights(np.array([-75, -45, -15, 15, 45, 75]))
  assert float(w.data.mean()) == pytest.approx(1.0)
  expected = 3 * np.array([1 - np.sqrt(3) / 2, (np.sqrt(3) - 1) / 2, 1 / 2,
                           1 / 2, (np.sqrt(3) - 1) / 2, 1 - np.sqrt(3) / 2])
  np.testing.assert_allclose(w.data, expected, rtol=1e-12)


def test_wind_vector_rmse():
  # metrics_test.py:84-131 -> [0, 10, nan]
  wv = metrics.WindVectorRMSESqrtBeforeTimeAvg(
      u_name='u_component_of_wind', v_name='v_component_of_wind',
      vector_name='wind_vector')
  kw = dict(variables_3d=['u_component_of_wind', 'v_component_of_wind'],
            variables_2d=[], time_start='2022-01-01', time_stop='2022-01-02')
  forecast = fixtures.mock_forecast_data(lead_stop='0 day', **kw)
  truth = fixtures.mock_truth_data(**kw)
  fmod = DS({'u_component_of_wind': NA([0, 3, np.nan], ('level',)),
             'v_component_of_wind': NA([0, -4, 1], ('level',))})
  tmod = DS({'u_component_of_wind': NA([0, -3, np.nan], ('level',)),
             'v_component_of_wind': NA([0, 4, 1], ('level',))})
  forecast = forecast + fmod
  truth = truth + tmod
  result = wv.compute(forecast, truth)['wind_vector'].data.squeeze()
  np.testing.assert_allclose(result, np.array([0, 10, np.nan]))


@pytest.mark.parametrize('invalid_value', [np.inf, np.nan])
def test_rmse_over_invalid_region(invalid_value):
  # metrics_test.py:133-152
  rmse = metrics.RMSESqrtBeforeTimeAvg()
  truth = DS({'wind_speed': NA(np.array([0.0, invalid_value, 0.0]
                                        ).reshape(1, 1, 3),
                               ('time', 'longitude', 'latitude'))},
             coords={'latitude': np.array([-45, 0, 45]),
                     'longitude': np.array([0]), 'time': np.array([0])})
  forecast = truth + 1
  actual = rmse.compute(forecast, truth)
  assert np.isnan(actual['wind_speed'].data)
  actual = rmse.compute(forecast, truth, region=regions.ExtraTropicalRegion())
  np.testing.assert_allclose(actual['wind_speed'].data, 1.0)


def test_daily_avg_acc_naming():
  # metrics_test.py:154-170
  truth, forecast = fixtures.get_random_truth_and_forecast(
      time_resolution='1 day')
  clim0 = truth.isel(time=0).expand_dims('dayofyear',
                                         coord=1 + np.arange(366))
  clim_mean = clim0.rename_vars({'geopotential': 'geopotential_mean'})
  acc1 = metrics.ACC(clim0).compute_chunk(forecast, truth)
  acc2 = metrics.ACC(clim_mean).compute_chunk(forecast, truth)
  np.testing.assert_allclose(acc1['geopotential'].data,
                             acc2['geopotential'].data)


@pytest.mark.parametrize('shape,axis', [((4, 5, 6), 0), ((4, 8, 6), 1),
                                        ((4, 2, 6), 2), ((4, 5, 7), -1),
                                        ((1, 5), 0), ((1, 5), 1)])
def test_rankdata_vs_scipy(shape, axis):
  # metrics_test.py:173-187
  x = np.random.RandomState(1729 + axis + np.prod(shape)).rand(*shape)
  np.testing.assert_array_equal(
      metrics.rankdata(x, axis), stats.rankdata(x, method='ordinal', axis=axis))


@pytest.mark.parametrize('ensemble_size', [2, 3, 5])
def test_crps_vs_brute_force(ensemble_size):
  # metrics_test.py:192-206
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=ensemble_size)
  expected = metrics.crps_brute_force(forecast, truth, skipna=False)
  got = metrics.CRPS().compute_chunk(forecast, truth)
  np.testing.assert_allclose(got['geopotential'].data,
                             expected['score']['geopotential'].data,
                             rtol=1e-5)


def test_crps_ensemble_size_1_gives_mae():
  # metrics_test.py:208-228
  truth, forecast = fixtures.get_random_truth_and_forecast(ensemble_size=1)
  expected_skill = metrics.spatial_average(
      abs(truth - forecast.isel(realization=0)), region=None, skipna=False)
  skill = metrics.CRPSSkill().compute_chunk(forecast, truth)
  spread = metrics.CRPSSpread().compute_chunk(forecast, truth)
  crps = metrics.CRPS().compute_chunk(forecast, truth)
  np.testing.assert_allclose(skill['geopotential'].data,
                             expected_skill['geopotential'].data)
  np.testing.assert_array_equal(spread['geopotential'].data, 0)
  np.testing.assert_allclose(crps['geopotential'].data,
                             expected_skill['geopotential'].data)


@pytest.mark.parametrize('skipna', [True, False])
def test_nan_forecasts_result_in_nan_crps(skipna):
  # metrics_test.py:230-267
  truth, forecast = fixtures.get_random_truth_and_forecast(
      variables=['geopotential', 'temperature'], ensemble_size=7)
  new_values = forecast['geopotential'].data.copy()
  new_values[(0,) * new_values.ndim] = np.nan
  forecast = forecast.copy(data={'geopotential': new_values,
                                 'temperature': forecast['temperature'].data})
  crps = metrics.CRPS().compute_chunk(forecast, truth, skipna=skipna)
  score_values = crps['geopotential'].data.copy()
  if skipna:
    assert not np.isnan(score_values[0, 0, 0])
  else:
    assert np.isnan(score_values[0, 0, 0])
  score_values[0, 0, 0] = 0
  assert np.all(np.isfinite(score_values))
  assert np.all(np.isfinite(crps['temperature'].data))
  expected = metrics.crps_brute_force(forecast, truth, skipna=skipna)['score']
  for k in ('geopotential', 'temperature'):
    np.testing.assert_allclose(crps[k].data, expected[k].data, rtol=1e-4,
                               atol=1e-4)


def test_crps_repeated_forecasts_are_okay():
  # metrics_test.py:269-281
  truth, forecast = fixtures.get_random_truth_and_forecast(ensemble_size=7)
  new_values = forecast['geopotential'].data.copy()
  new_values[0] = new_values[1]
  forecast = forecast.copy(data={'geopotential': new_values})
  crps = metrics.CRPS().compute_chunk(forecast, truth)
  expected = metrics.crps_brute_force(forecast, truth, skipna=False)['score']
  np.testing.assert_allclose(crps['geopotential'].data,
                             expected['geopotential'].data, rtol=1e-5)


@pytest.mark.parametrize('ensemble_size', [1, 2, 3, 100])
def test_ensemble_mean_rmse_and_stddev(ensemble_size):
  # metrics_test.py:784-831
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=ensemble_size)
  rmse = metrics.EnsembleMeanRMSESqrtBeforeTimeAvg().compute_chunk(
      forecast, truth)
  stddev = metrics.EnsembleStddevSqrtBeforeTimeAvg().compute_chunk(
      forecast, truth)
  for ds in (rmse, stddev):
    assert dict(ds['geopotential'].sizes) == {
        k: v for k, v in forecast.sizes.items()
        if k not in ('realization', 'latitude', 'longitude')}
  if ensemble_size == 1:
    np.testing.assert_array_equal(stddev['geopotential'].data, 0)
    return
  n = np.prod(rmse['geopotential'].shape)
  atol = 4 * (1 / np.sqrt(n) + 1 / ensemble_size)
  np.testing.assert_allclose(rmse['geopotential'].data.mean(),
                             stddev['geopotential'].data.mean(), atol=atol)


def test_effect_of_large_bias_on_rmse():
  # metrics_test.py:833-844
  truth, forecast = fixtures.get_random_truth_and_forecast(ensemble_size=10)
  truth = truth + 1000
  mean_rmse = metrics.EnsembleMeanRMSESqrtBeforeTimeAvg().compute_chunk(
      forecast, truth)['geopotential'].data.mean()
  np.testing.assert_allclose(1000, mean_rmse, rtol=1e-3)


def test_debiased_mse_versus_large_ensemble():
  # metrics_test.py:856-893
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=1000, spatial_resolution_in_degrees=20)
  small = forecast.isel(realization=slice(2))
  mse_large = metrics.EnsembleMeanMSE().compute_chunk(forecast, truth)
  mse_small = metrics.EnsembleMeanMSE().compute_chunk(small, truth)
  mse_debiased_small = metrics.DebiasedEnsembleMeanMSE().compute_chunk(
      small, truth)
  var_large = metrics.EnsembleVariance().compute_chunk(forecast, truth)
  anticipated_bias = var_large['geopotential'].data.max() / 2
  observed_bias = (mse_small['geopotential'].data
                   - mse_large['geopotential'].data).mean()
  np.testing.assert_allclose(observed_bias, anticipated_bias, rtol=0.05)
  total_points = np.prod(list(truth.sizes.values()))
  stderr = np.sqrt(var_large['geopotential'].data.max() / total_points)
  np.testing.assert_allclose(mse_large['geopotential'].data.mean(),
                             mse_debiased_small['geopotential'].data.mean(),
                             atol=4 * stderr)


def _gaussian_fixture():
  kw = dict(variables_3d=[], time_start='2022-01-01')
  forecast = fixtures.mock_forecast_data(
      variables_2d=['2m_temperature', '2m_temperature_std'],
      time_stop='2022-01-02', lead_stop='1 day', **kw)
  truth = fixtures.mock_truth_data(variables_2d=['2m_temperature'],
                                   time_stop='2022-01-20', **kw)
  return forecast, truth


def test_gaussian_crps_known_answer():
  # metrics_test.py:286-304 (forecast and truth only share ONE time label:
  # xarray's inner join is part of the contract)
  forecast, truth = _gaussian_fixture()
  forecast = forecast + 1.0
  truth = truth + 1.02
  result = metrics.GaussianCRPS().compute(forecast, truth)
  np.testing.assert_allclose(result['2m_temperature'].data,
                             np.array([0.23385455, 0.23385455]), rtol=1e-6)


def test_gaussian_variance_known_answer():
  # metrics_test.py:340-362
  forecast, truth = _gaussian_fixture()
  forecast['2m_temperature_std'] = forecast['2m_temperature_std'] + 1.0
  result = metrics.GaussianVariance().compute(forecast, truth)
  np.testing.assert_allclose(result['2m_temperature'].data, [1.0, 1.0])


@pytest.mark.parametrize('ensemble_size', [1, 2, 3])
def test_energy_score_on_random_dataset(ensemble_size):
  # metrics_test.py:924-965
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=ensemble_size)
  score = metrics.EnergyScore().compute_chunk(forecast, truth)
  spread = metrics.EnergyScoreSpread().compute_chunk(forecast, truth)
  skill = metrics.EnergyScoreSkill().compute_chunk(forecast, truth)
  want_sizes = {k: v for k, v in forecast.sizes.items()
                if k not in ('realization', 'latitude', 'longitude')}
  for ds in (score, spread, skill):
    assert dict(ds['geopotential'].sizes) == want_sizes
  if ensemble_size == 1:
    np.testing.assert_array_equal(spread['geopotential'].data, 0)
    np.testing.assert_allclose(score['geopotential'].data,
                               skill['geopotential'].data)
    return
  n = np.prod(score['geopotential'].shape)
  np.testing.assert_allclose(
      spread['geopotential'].data.mean(), skill['geopotential'].data.mean(),
      atol=4 * score['geopotential'].data.std() / np.sqrt(n))
  np.testing.assert_allclose(
      score['geopotential'].data,
      skill['geopotential'].data - 0.5 * spread['geopotential'].data)


def test_energy_score_effect_of_bias():
  # metrics_test.py:967-981
  truth, forecast = fixtures.get_random_truth_and_forecast(ensemble_size=10)
  truth = truth + 1000
  score = metrics.EnergyScore().compute_chunk(forecast, truth)
  spread = metrics.EnergyScoreSpread().compute_chunk(forecast, truth)
  np.testing.assert_allclose(1000, score['geopotential'].data.mean(), rtol=1e-3)
  np.testing.assert_allclose(spread['geopotential'].data.mean(), np.sqrt(2),
                             rtol=0.05)


def test_land_region():
  # regions_test.py:25-49
  truth, forecast = fixtures.get_random_truth_and_forecast(
      spatial_resolution_in_degrees=30)
  lat = truth.coord('latitude')
  lon = truth.coord('longitude')
  lsm = NA(np.broadcast_to((lat >= 1).astype(float)[:, None],
                           (len(lat), len(lon))).copy(),
           ('latitude', 'longitude'))
  # forecast == truth wherever lsm == 1; error only where lat <= 0.
  truth_b = DS({'geopotential': NA(np.broadcast_to(
      truth['geopotential'].data, forecast['geopotential'].shape).copy(),
      forecast['geopotential'].dims)}, forecast.coords)
  err = NA((lat <= 0).astype(float), ('latitude',))
  forecast = truth_b + DS({'geopotential': err})
  region = regions.LandRegion(land_sea_mask=lsm, latitude=lat, longitude=lon)
  rmse = metrics.RMSESqrtBeforeTimeAvg().compute_chunk(
      forecast, truth, region=region)
  np.testing.assert_allclose(rmse['geopotential'].data, 0, atol=1e-12)
  rmse_global = metrics.RMSESqrtBeforeTimeAvg().compute_chunk(forecast, truth)
  assert np.all(rmse_global['geopotential'].data > 0.1)


# ---------------------------------------------------------------------------
# Zonal energy spectrum (derived_variables_test.py)
# ---------------------------------------------------------------------------
def _multispectral(res=5, lat=None, min_wl=50, max_wl=100):
  nlat = round(180 / res) + 1
  nlon = round(360 / res)
  latitude = np.linspace(-90, 90, nlat)
  if lat is not None:
    latitude = np.asarray(lat, dtype=float)
  longitude = np.linspace(0, 360, nlon, endpoint=False)
  level = np.array([500, 700, 850])
  x = np.zeros((len(level), len(longitude), len(latitude)))
  for wl in np.linspace(min_wl, max_wl, num=100):
    x += (np.cos(2 * np.pi * longitude / wl)[None, :, None]
          * np.exp(-wl / max_wl)
          * np.sin(level / 500)[:, None, None]
          * np.cos(latitude / 100)[None, None, :]) / 100
  return x, latitude, longitude


def test_spectrum_parseval():
  # derived_variables_test.py:415-435: sum_k S[k] == spacing * sum_l f^2
  lat = np.arange(-30, 31, 5)
  x, latitude, longitude = _multispectral(res=5, lat=lat)
  spec, _, _ = spectrum_np.zonal_energy_spectrum(
      x, latitude, longitude, lat_axis=2, lon_axis=1)
  spacing = spectrum_np.lon_spacing_m(latitude, longitude)
  lhs = spec.sum(axis=-1)                      # (level, lat)
  rhs = (x ** 2).sum(axis=1) * spacing[None]   # (level, lat)
  np.testing.assert_allclose(lhs, rhs, rtol=2e-3)


@pytest.mark.parametrize('lat0', [0.0, 30.0, 60.0])
def test_spectrum_peak(lat0):
  # derived_variables_test.py:290-321: cosine with 100 degree wavelength
  res = 10
  longitude = np.linspace(0, 360, 36, endpoint=False)
  latitude = np.array([lat0])
  x = 10 * np.cos(2 * np.pi * longitude / 100)[None, :]  # (lat, lon)
  spec, freq, wavelength = spectrum_np.zonal_energy_spectrum(
      x, latitude, longitude, lat_axis=0, lon_axis=1)
  k = int(np.argmax(spec[0]))
  circ = spectrum_np.circumference(latitude)[0]
  expected_wavelength = circ * 100 / 360
  nearest = int(np.argmin(np.abs(wavelength[1:, 0] - expected_wavelength))) + 1
  assert k == nearest


def test_spectrum_last_bin_doubled_even_n():
  # derived_variables.py:600 quirk
  x = np.array([[1.0, -1.0, 1.0, -1.0]])  # pure Nyquist, N = 4
  p = spectrum_np.simple_power(x)
  np.testing.assert_allclose(p[0], [0, 0, 2.0])


@pytest.mark.parametrize('ensemble_size,num_bins', [(1, None), (10, None),
                                                    (2, None), (9, 5)])
def test_rank_histogram_well_and_mis_calibrated(ensemble_size, num_bins):
  # metrics_test.py:540-600
  num_bins = ensemble_size + 1 if num_bins is None else num_bins
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=ensemble_size, time_start='2019-12-01',
      time_stop='2019-12-10', levels=(0, 1, 2, 3, 4))
  data = forecast['geopotential'].data.copy()
  lev = forecast['geopotential'].dims.index('level')
  sl = lambda i: tuple(i if a == lev else slice(None) for a in range(data.ndim))
  data[sl(1)] *= 0.1
  data[sl(2)] *= 10
  data[sl(3)] -= 1
  data[sl(4)] += 1
  forecast = forecast.copy(data={'geopotential': data})
  one_hot = metrics.RankHistogram(num_bins=num_bins).compute_chunk(forecast,
                                                                   truth)
  v = one_hot['geopotential']
  assert v.sizes == {**{d: s for d, s in forecast.sizes.items()
                        if d != 'realization'}, 'bins': num_bins}
  avg = tuple(d for d in v.dims if d not in ('bins', 'level'))
  sample_size = np.prod([v.sizes[d] for d in avg])
  rtol = 5 * np.sqrt((num_bins - 1) / sample_size)
  hist = v.mean(avg).transpose('level', 'bins').data
  np.testing.assert_allclose(1 / num_bins, hist[0], rtol=rtol)
  if num_bins > 2:
    convex, concave = hist[1], hist[2]
    assert (np.diff(convex[:len(convex) // 2 + 1]) < 0).all()
    assert (np.diff(convex[len(convex) // 2:]) > 0).all()
    assert (np.diff(concave[:len(concave) // 2 + 1]) > 0).all()
    assert (np.diff(concave[len(concave) // 2:]) < 0).all()
  assert (np.diff(hist[3]) > 0).all()
  assert (np.diff(hist[4]) < 0).all()


@pytest.mark.parametrize('n_bins', [3, 4, 10, 11])
def test_central_reliability_perfectly_calibrated(n_bins):
  # metrics_test.py:666-700
  hist = np.ones((n_bins,)) / n_bins
  probs, desired = metrics.central_reliability(hist)
  expected = np.ones((n_bins // 2,))
  if n_bins % 2:
    expected = np.concatenate(([0.5], expected))
  expected = np.cumsum(expected) / np.sum(expected)
  assert len(desired) == n_bins // 2 + n_bins % 2
  np.testing.assert_allclose(probs, expected)
  np.testing.assert_allclose(desired, expected)
  with pytest.raises(ValueError):
    metrics.central_reliability(np.ones(2) / 2)


_RELIABILITY_VECTORS = [
    # metrics_test.py:700-779 (hist, expected_prob, desired_prob)
    ([0.2, 0.1, 0.7], [0.1, 1.0], [1 / 3, 1.0]),
    ([0.2, 0.0, 0.1, 0.1, 0.6], [0.1, 0.2, 1.0], [1 / 5, 3 / 5, 1.0]),
    ([0.1, 0.1, 0.5, 0.3], [0.6, 1.0], [1 / 2, 1.0]),
    ([0.1, 0.1, 0.3, 0.2, 0.0, 0.3], [0.5, 0.6, 1.0], [1 / 3, 2 / 3, 1.0]),
]


@pytest.mark.parametrize('hist,expected,desired', _RELIABILITY_VECTORS)
def test_central_reliability_particular_histograms(hist, expected, desired):
  probs, want = metrics.central_reliability(np.array(hist))
  np.testing.assert_allclose(probs, expected, rtol=1e-12)
  np.testing.assert_allclose(want, desired, rtol=1e-12)


def _censored_case(ensemble_size, cutoff_below):
  """metrics_test.py:612-630: truth/forecast with a point mass at 0."""
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=ensemble_size, time_start='2019-12-01',
      time_stop='2019-12-20')
  comp = np.less_equal if cutoff_below else np.greater_equal
  cens = lambda ds: ds.copy(data={
      k: np.where(comp(v.data, 0), 0, v.data) for k, v in ds.items()})
  return cens(truth), cens(forecast)


@pytest.mark.parametrize('cutoff_below', [True, False])
@pytest.mark.parametrize('ensemble_size', [1, 2, 3, 10])
def test_rank_histogram_repeated_entries_get_random_bin(ensemble_size,
                                                        cutoff_below):
  # metrics_test.py:603-650
  num_bins = ensemble_size + 1
  truth, forecast = _censored_case(ensemble_size, cutoff_below)
  v = metrics.RankHistogram(num_bins=num_bins, seed=802701).compute_chunk(
      forecast, truth)['geopotential']
  sample_size = v.data.size / num_bins
  rtol = 5 * (num_bins - 1) / np.sqrt(sample_size)
  hist = v.data.reshape(-1, num_bins).mean(0)
  np.testing.assert_allclose(hist, 1 / num_bins, rtol=rtol)


def test_perfect_prediction_zero_ensemble_mean_rmse():
  # metrics_test.py:842-851
  truth, _ = fixtures.get_random_truth_and_forecast(ensemble_size=10)
  forecast = truth.expand_dims('realization', size=1)
  rmse = metrics.EnsembleMeanRMSESqrtBeforeTimeAvg().compute_chunk(forecast,
                                                                   truth)
  np.testing.assert_allclose(rmse['geopotential'].data, 0.0, atol=1e-12)


def test_gaussian_crps_is_the_limit_of_ensemble_crps():
  # metrics_test.py:306-343 (2000 members instead of 5000: rtol 2e-2 holds)
  kw = dict(variables_3d=[], time_start='2022-01-01')
  forecast = fixtures.mock_forecast_data(
      variables_2d=['2m_temperature', '2m_temperature_std'],
      time_stop='2022-01-02', lead_stop='1 day', **kw)
  ens = fixtures.mock_forecast_data(
      variables_2d=['2m_temperature'], time_stop='2022-01-02',
      lead_stop='1 day', ensemble_size=2000, **kw)
  truth = fixtures.mock_truth_data(variables_2d=['2m_temperature'],
                                   time_stop='2022-01-20', **kw)
  from oracle.named import DS, NA
  forecast = DS({'2m_temperature': forecast['2m_temperature'] + 0.1,
                 '2m_temperature_std': forecast['2m_temperature_std'] + 1.0},
                forecast.coords)
  e = ens['2m_temperature']
  noise = np.random.RandomState(0).randn(*e.shape).astype(np.float32)
  ens = ens.copy(data={'2m_temperature': e.data + noise + np.float32(0.1)})
  gaussian = metrics.GaussianCRPS().compute(forecast, truth)
  ensemble = metrics.CRPS().compute(ens, truth)
  np.testing.assert_allclose(gaussian['2m_temperature'].data,
                             ensemble['2m_temperature'].data, rtol=2e-2)
