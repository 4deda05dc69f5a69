"""Seeded cases shared by the reference-vector generator and the tests that
consume its output (tests/golden/reference_vectors_v1.npz).

Everything here is plain NumPy: a case is a dict of named arrays
(`{'data': ndarray, 'dims': tuple}`) plus 1-D coordinates, and a list of
(label, metric factory, region factory) entries.  The factories receive the
*implementation's* modules -- the reference's `weatherbench2.metrics/regions/
thresholds` (generator), the NumPy oracle's, or the product's -- which all use
the reference's class and argument names, plus an adapter that wraps the case's
arrays in that implementation's dataset type.  So one description drives

  make_reference_vectors.py   reference code on the mini-xarray  -> .npz
  test_reference_vectors.py   oracle (CPU) and HIP path (GPU)    == .npz
"""
from __future__ import annotations

import numpy as np

LAT10 = np.linspace(-90.0, 90.0, 19)
LON10 = np.linspace(0.0, 360.0, 36, endpoint=False)


def _arr(data, *dims):
  return {'data': np.ascontiguousarray(data), 'dims': tuple(dims)}


def _times(n, start='2020-02-27T00', step_h=12):
  return (np.datetime64(start, 'ns') +
          np.arange(n) * np.timedelta64(step_h, 'h').astype('timedelta64[ns]'))


def _leads(n, step_h=6):
  return np.arange(n) * np.timedelta64(step_h, 'h').astype('timedelta64[ns]')


# --------------------------------------------------------------------------
# regions (label -> factory(regions_module, ctx))
# --------------------------------------------------------------------------
def region_factories():
  return {
      'global': lambda r, c: None,
      'tropics': lambda r, c: r.SliceRegion(lat_slice=slice(-20, 20)),
      'extra_tropics_slices': lambda r, c: r.SliceRegion(
          lat_slice=[slice(None, -20), slice(20, None)]),
      'europe': lambda r, c: r.SliceRegion(
          lat_slice=slice(35, 75),
          lon_slice=[slice(360 - 12.5, None), slice(0, 42.5)]),
      'north_pacific': lambda r, c: r.SliceRegion(
          lat_slice=slice(25, 60), lon_slice=slice(145, 360 - 130)),
      'antarctic': lambda r, c: r.SliceRegion(lat_slice=slice(-90, -60)),
      'extra_tropical': lambda r, c: r.ExtraTropicalRegion(),
      'land': lambda r, c: r.LandRegion(land_sea_mask=c['lsm']),
      'land_thr': lambda r, c: r.LandRegion(land_sea_mask=c['lsm'],
                                            threshold=0.5),
      'land_europe': lambda r, c: r.CombinedRegion(regions=[
          r.SliceRegion(lat_slice=slice(35, 75),
                        lon_slice=[slice(360 - 12.5, None), slice(0, 42.5)]),
          r.LandRegion(land_sea_mask=c['lsm'], threshold=0.3)]),
  }


DET_METRICS = {
    'mse': lambda m, c: m.MSE(),
    'rmse': lambda m, c: m.RMSESqrtBeforeTimeAvg(),
    'mae': lambda m, c: m.MAE(),
    'bias': lambda m, c: m.Bias(),
    'acc': lambda m, c: m.ACC(climatology=c['climatology']),
}

WIND_METRICS = {
    'mse_wind': lambda m, c: m.MSE(wind_vector_mse=[m.WindVectorMSE(
        u_name='u_component_of_wind', v_name='v_component_of_wind',
        vector_name='wind_vector')]),
    'rmse_wind': lambda m, c: m.RMSESqrtBeforeTimeAvg(wind_vector_rmse=[
        m.WindVectorRMSESqrtBeforeTimeAvg(
            u_name='u_component_of_wind', v_name='v_component_of_wind',
            vector_name='wind_vector')]),
}

ENS_METRICS = {
    'crps': lambda m, c: m.CRPS(),
    'crps_spread': lambda m, c: m.CRPSSpread(),
    'crps_skill': lambda m, c: m.CRPSSkill(),
    'ensemble_mean_mse': lambda m, c: m.EnsembleMeanMSE(),
    'ensemble_mean_rmse': lambda m, c: m.EnsembleMeanRMSESqrtBeforeTimeAvg(),
    'ensemble_variance': lambda m, c: m.EnsembleVariance(),
    'ensemble_stddev': lambda m, c: m.EnsembleStddevSqrtBeforeTimeAvg(),
    'debiased_ensemble_mean_mse': lambda m, c: m.DebiasedEnsembleMeanMSE(),
}

SPATIAL_METRICS = {
    'spatial_mse': lambda m, c: m.SpatialMSE(),
    'spatial_mae': lambda m, c: m.SpatialMAE(),
    'spatial_bias': lambda m, c: m.SpatialBias(),
}

SPATIAL_ENS_METRICS = {
    'spatial_crps': lambda m, c: m.SpatialCRPS(),
    'spatial_crps_spread': lambda m, c: m.SpatialCRPSSpread(),
    'spatial_crps_skill': lambda m, c: m.SpatialCRPSSkill(),
    'spatial_ensemble_variance': lambda m, c: m.SpatialEnsembleVariance(),
    'spatial_ensemble_mean_mse': lambda m, c: m.SpatialEnsembleMeanMSE(),
    'debiased_spatial_ensemble_mean_mse':
        lambda m, c: m.DebiasedSpatialEnsembleMeanMSE(),
}


# --------------------------------------------------------------------------
# cases
# --------------------------------------------------------------------------
def _insert_nan(x, rs, frac):
  out = x.copy()
  out[rs.rand(*x.shape) < frac] = np.nan
  return out


def det_case(dtype, layout='lonlat', nan_frac=0.0, seed=11):
  """Deterministic suite: truth(time, level, lon, lat), forecast with a lead
  dimension, an hourly climatology restricted to the days the case touches
  (label-based selection, metrics.py:398-404) -- 2020 is a leap year, the valid
  times cross Feb 29 -- and a land-sea mask."""
  rs = np.random.RandomState(seed)
  n_time, n_lead, levels = 6, 2, np.array([500, 850])
  spatial = (('longitude', 'latitude') if layout == 'lonlat'
             else ('latitude', 'longitude'))
  sshape = tuple(len(LON10) if d == 'longitude' else len(LAT10)
                 for d in spatial)
  times = _times(n_time)
  truth = rs.standard_normal((n_time, len(levels)) + sshape)
  forecast = truth[None] + 0.5 * rs.standard_normal(
      (n_lead, n_time, len(levels)) + sshape)
  days = np.arange(55, 65)           # dayofyear labels present
  hours = np.array([0, 6, 12, 18])
  clim = 0.3 * rs.standard_normal((len(hours), len(days), len(levels)) + sshape)
  lsm = rs.rand(len(LAT10), len(LON10))
  lsm[lsm < 0.25] = 0.0
  if nan_frac:
    truth = _insert_nan(truth, rs, nan_frac)
    forecast = _insert_nan(forecast, rs, nan_frac)
  c = lambda a: a.astype(dtype)
  return {
      'coords': {'time': times, 'prediction_timedelta': _leads(n_lead),
                 'level': levels, 'latitude': LAT10, 'longitude': LON10,
                 'hour': hours, 'dayofyear': days},
      'truth': {'geopotential': _arr(c(truth), 'time', 'level', *spatial)},
      'forecast': {'geopotential': _arr(c(forecast), 'prediction_timedelta',
                                        'time', 'level', *spatial)},
      'climatology': {'geopotential': _arr(c(clim), 'hour', 'dayofyear',
                                           'level', *spatial)},
      'lsm': _arr(lsm, 'latitude', 'longitude'),
  }


def wind_case(dtype, seed=12):
  rs = np.random.RandomState(seed)
  n_time, levels = 3, np.array([500, 700, 850])
  shape = (n_time, len(levels), len(LON10), len(LAT10))
  dims = ('time', 'level', 'longitude', 'latitude')
  names = ('u_component_of_wind', 'v_component_of_wind', 'temperature')
  truth = {k: _arr(rs.standard_normal(shape).astype(dtype), *dims)
           for k in names}
  forecast = {k: _arr((truth[k]['data'] + 0.7 * rs.standard_normal(shape)
                       ).astype(dtype), *dims) for k in names}
  return {'coords': {'time': _times(n_time), 'level': levels,
                     'latitude': LAT10, 'longitude': LON10},
          'truth': truth, 'forecast': forecast,
          'lsm': _arr(rs.rand(len(LAT10), len(LON10)), 'latitude',
                      'longitude')}


def ens_case(dtype, n_member=5, nan_frac=0.0, seed=13):
  rs = np.random.RandomState(seed + n_member)
  n_time, n_lead, levels = 3, 2, np.array([500, 850])
  sshape = (len(LON10), len(LAT10))
  truth = rs.standard_normal((n_time, len(levels)) + sshape)
  forecast = truth[None, None] * 0.6 + rs.standard_normal(
      (n_member, n_lead, n_time, len(levels)) + sshape)
  lsm = rs.rand(len(LAT10), len(LON10))
  if nan_frac:
    truth = _insert_nan(truth, rs, nan_frac)
    forecast = _insert_nan(forecast, rs, nan_frac)
  return {
      'coords': {'time': _times(n_time), 'prediction_timedelta': _leads(n_lead),
                 'level': levels, 'latitude': LAT10, 'longitude': LON10,
                 'realization': np.arange(n_member)},
      'truth': {'geopotential': _arr(truth.astype(dtype), 'time', 'level',
                                     'longitude', 'latitude')},
      'forecast': {'geopotential': _arr(
          forecast.astype(dtype), 'realization', 'prediction_timedelta', 'time',
          'level', 'longitude', 'latitude')},
      'lsm': _arr(lsm, 'latitude', 'longitude'),
  }


def spectrum_case(dtype, n_lon=72, seed=14):
  rs = np.random.RandomState(seed + n_lon)
  lat = np.arange(-80.0, 81.0, 10.0)
  lon = np.linspace(0.0, 360.0, n_lon, endpoint=False)
  x = rs.standard_normal((3, 2, len(lat), n_lon)) * 5.0 + 1.0
  return {'coords': {'time': _times(3), 'level': np.array([500, 850]),
                     'latitude': lat, 'longitude': lon},
          'dataset': {'geopotential': _arr(x.astype(dtype), 'time', 'level',
                                           'latitude', 'longitude')}}


def case_table():
  """name -> (case builder, metric table, region labels, skipna, mode)
  mode 'chunk' = compute_chunk, 'compute' = compute (time mean included)."""
  R = list(region_factories())
  small = ['global', 'europe', 'land_thr']
  return {
      'det_f64': (lambda: det_case(np.float64), DET_METRICS, R, False, 'chunk'),
      'det_f32': (lambda: det_case(np.float32), DET_METRICS, R, False, 'chunk'),
      'det_f32_latlon': (lambda: det_case(np.float32, layout='latlon', seed=21),
                         DET_METRICS, small + ['extra_tropical'], False,
                         'chunk'),
      'det_f32_nan_skipna': (lambda: det_case(np.float32, nan_frac=0.05,
                                              seed=22), DET_METRICS, R, True,
                             'chunk'),
      'det_f64_nan_noskip': (lambda: det_case(np.float64, nan_frac=0.002,
                                              seed=23), DET_METRICS, small,
                             False, 'chunk'),
      'det_f32_compute': (lambda: det_case(np.float32, nan_frac=0.05, seed=24),
                          DET_METRICS, small, True, 'compute'),
      'wind_f32': (lambda: wind_case(np.float32), WIND_METRICS, small, False,
                   'chunk'),
      'ens_f32': (lambda: ens_case(np.float32), ENS_METRICS, R, False, 'chunk'),
      'ens_f64': (lambda: ens_case(np.float64), ENS_METRICS, small, False,
                  'chunk'),
      'ens_f32_m2': (lambda: ens_case(np.float32, n_member=2), ENS_METRICS,
                     small, False, 'chunk'),
      'ens_f32_m1': (lambda: ens_case(np.float32, n_member=1), ENS_METRICS,
                     ['global'], False, 'chunk'),
      'ens_f64_nan_skipna': (lambda: ens_case(np.float64, nan_frac=0.03,
                                              seed=31), ENS_METRICS, small,
                             True, 'chunk'),
      'ens_f32_compute': (lambda: ens_case(np.float32, seed=32), ENS_METRICS,
                          ['global', 'europe'], False, 'compute'),
      'spatial_f32': (lambda: det_case(np.float32, seed=41), SPATIAL_METRICS,
                      ['global'], False, 'chunk'),
      'spatial_ens_f32': (lambda: ens_case(np.float32, seed=42),
                          SPATIAL_ENS_METRICS, ['global'], False, 'chunk'),
  }


SPECTRUM_CASES = {
    'spectrum_f32_72': lambda: spectrum_case(np.float32, 72),
    'spectrum_f64_72': lambda: spectrum_case(np.float64, 72),
    'spectrum_f32_64': lambda: spectrum_case(np.float32, 64),
    'spectrum_f32_45': lambda: spectrum_case(np.float32, 45),  # odd length
}


# --------------------------------------------------------------------------
# tier 2: Gaussian family, threshold metrics, energy score, SEEPS, rank
# histogram.  Metric factories find the implementation's thresholds module in
# ctx['th'].
# --------------------------------------------------------------------------
T2M = '2m_temperature'
QUANTILES = (0.25, 0.5, 0.9)


def _quantile_thresholds(c, qs):
  return [c['th'].QuantileThreshold(climatology=c['clim_q'], quantile=q)
          for q in qs]


def _gaussian_thresholds(c, qs):
  return [c['th'].GaussianQuantileThreshold(climatology=c['clim_g'],
                                            quantile=q) for q in qs]


GAUSS_METRICS = {
    'gaussian_crps': lambda m, c: m.GaussianCRPS(),
    'gaussian_variance': lambda m, c: m.GaussianVariance(),
    'gaussian_brier_q': lambda m, c: m.GaussianBrierScore(
        thresholds=_quantile_thresholds(c, (0.25, 0.9))),
    'gaussian_brier_g': lambda m, c: m.GaussianBrierScore(
        thresholds=_gaussian_thresholds(c, (0.5, 0.9))),
    'gaussian_ignorance_q': lambda m, c: m.GaussianIgnoranceScore(
        thresholds=_quantile_thresholds(c, (0.5,))),
    'gaussian_rps_q': lambda m, c: m.GaussianRPS(
        thresholds=_quantile_thresholds(c, QUANTILES)),
}

ENS_THRESHOLD_METRICS = {
    'ensemble_brier': lambda m, c: m.EnsembleBrierScore(
        thresholds=_quantile_thresholds(c, (0.25, 0.9))),
    'debiased_ensemble_brier': lambda m, c: m.DebiasedEnsembleBrierScore(
        thresholds=_gaussian_thresholds(c, (0.5, 0.9))),
    'ensemble_ignorance': lambda m, c: m.EnsembleIgnoranceScore(
        thresholds=_quantile_thresholds(c, (0.5,))),
    'ensemble_rps': lambda m, c: m.EnsembleRPS(
        thresholds=_quantile_thresholds(c, QUANTILES)),
}

ENERGY_METRICS = {
    'energy_score': lambda m, c: m.EnergyScore(),
    'energy_score_spread': lambda m, c: m.EnergyScoreSpread(),
    'energy_score_skill': lambda m, c: m.EnergyScoreSkill(),
}

SEEPS_METRICS = {
    'seeps': lambda m, c: m.SEEPS(climatology=c['climatology']),
    'spatial_seeps': lambda m, c: m.SpatialSEEPS(climatology=c['climatology']),
}

RANK_METRICS = {
    'rank_histogram': lambda m, c: m.RankHistogram(seed=802701),
    'rank_histogram_3bins': lambda m, c: m.RankHistogram(num_bins=3, seed=7),
}


def _threshold_climatologies(rs, dtype, base, sshape, hours, days):
  from scipy import stats
  mean = base + rs.standard_normal((len(hours), len(days)) + sshape)
  std = 0.5 + rs.rand(len(hours), len(days), *sshape)
  q = np.stack([mean + stats.norm.ppf(p) * std +
                0.05 * rs.standard_normal(mean.shape) for p in QUANTILES])
  tail = ('hour', 'dayofyear', 'longitude', 'latitude')
  return ({T2M + '_quantile': _arr(q.astype(dtype), 'quantile', *tail)},
          {T2M: _arr(mean.astype(dtype), *tail),
           T2M + '_std': _arr(std.astype(dtype), *tail)})


def gauss_case(dtype, seed=51):
  """Gaussian forecasts (mean + `_std` variables) of a surface variable, with
  quantile / Gaussian-quantile climatologies for the thresholds."""
  rs = np.random.RandomState(seed)
  n_time, n_lead = 5, 2
  sshape = (len(LON10), len(LAT10))
  hours, days = np.array([0, 12]), np.arange(57, 63)
  truth = 280.0 + 2.0 * rs.standard_normal((n_time,) + sshape)
  fmean = truth[None] + rs.standard_normal((n_lead, n_time) + sshape)
  fstd = 0.5 + rs.rand(n_lead, n_time, *sshape)
  clim_q, clim_g = _threshold_climatologies(rs, dtype, 280.0, sshape, hours,
                                            days)
  fd = ('prediction_timedelta', 'time', 'longitude', 'latitude')
  return {
      'coords': {'time': _times(n_time), 'prediction_timedelta': _leads(n_lead),
                 'latitude': LAT10, 'longitude': LON10, 'hour': hours,
                 'dayofyear': days, 'quantile': np.array(QUANTILES)},
      'truth': {T2M: _arr(truth.astype(dtype), 'time', 'longitude',
                          'latitude')},
      'forecast': {T2M: _arr(fmean.astype(dtype), *fd),
                   T2M + '_std': _arr(fstd.astype(dtype), *fd)},
      'clim_q': clim_q, 'clim_g': clim_g,
      'lsm': _arr(rs.rand(len(LAT10), len(LON10)), 'latitude', 'longitude'),
  }


def ens_threshold_case(dtype, n_member=6, nan_frac=0.0, seed=52):
  rs = np.random.RandomState(seed)
  n_time = 5
  sshape = (len(LON10), len(LAT10))
  hours, days = np.array([0, 12]), np.arange(57, 63)
  truth = 280.0 + 2.0 * rs.standard_normal((n_time,) + sshape)
  forecast = truth[None] + 1.5 * rs.standard_normal((n_member, n_time) + sshape)
  clim_q, clim_g = _threshold_climatologies(rs, dtype, 280.0, sshape, hours,
                                            days)
  if nan_frac:
    truth = _insert_nan(truth, rs, nan_frac)
    forecast = _insert_nan(forecast, rs, nan_frac)
  return {
      'coords': {'time': _times(n_time), 'latitude': LAT10, 'longitude': LON10,
                 'hour': hours, 'dayofyear': days,
                 'quantile': np.array(QUANTILES),
                 'realization': np.arange(n_member)},
      'truth': {T2M: _arr(truth.astype(dtype), 'time', 'longitude',
                          'latitude')},
      'forecast': {T2M: _arr(forecast.astype(dtype), 'realization', 'time',
                             'longitude', 'latitude')},
      'clim_q': clim_q, 'clim_g': clim_g,
      'lsm': _arr(rs.rand(len(LAT10), len(LON10)), 'latitude', 'longitude'),
  }


def seeps_case(dtype, seed=53):
  """24 h precipitation in metres with a `valid_time` coordinate on forecast
  and truth (metrics.py:449-451 reads `da.valid_time`)."""
  rs = np.random.RandomState(seed)
  n_time = 6
  name = 'total_precipitation_24hr'
  sshape = (len(LON10), len(LAT10))
  hours, days = np.array([0, 12]), np.arange(57, 63)
  rain = lambda: np.where(rs.rand(n_time, *sshape) < 0.45, 0.0,
                          rs.gamma(0.6, 0.004, (n_time,) + sshape))
  truth, forecast = rain(), rain()
  forecast[0] = truth[0]  # one perfect time step
  wet = 0.002 + 0.004 * rs.rand(len(hours), len(days), *sshape)
  dry_fraction = rs.rand(len(hours), len(days), *sshape)
  dry_fraction[..., :6, :] = 0.95  # p1 > max_p1: masked out
  dry_fraction[..., 6:9, :] = 0.02  # p1 < min_p1: masked out
  dry_fraction[0, :2, 12:15, :] = np.nan  # skipped by the mean (skipna=None)
  dry_fraction[:, :, 15, 3] = np.nan      # NaN everywhere: p1 NaN, masked out
  tail = ('hour', 'dayofyear', 'longitude', 'latitude')
  times = _times(n_time)
  return {
      'coords': {'time': times, 'latitude': LAT10, 'longitude': LON10,
                 'hour': hours, 'dayofyear': days},
      'extra_coords': {'valid_time': _arr(times, 'time')},
      'truth': {name: _arr(truth.astype(dtype), 'time', 'longitude',
                           'latitude')},
      'forecast': {name: _arr(forecast.astype(dtype), 'time', 'longitude',
                              'latitude')},
      'climatology': {
          name + '_seeps_threshold': _arr(wet.astype(dtype), *tail),
          name + '_seeps_dry_fraction': _arr(dry_fraction.astype(dtype),
                                             *tail)},
      'lsm': _arr(rs.rand(len(LAT10), len(LON10)), 'latitude', 'longitude'),
  }


def rank_case(dtype, n_member=5, ties=True, seed=54):
  """Members and truth rounded to one decimal: many exact ties, so the seeded
  tie-breaking stream (metrics.py:1955-1985) decides the bins."""
  rs = np.random.RandomState(seed)
  n_time, levels = 4, np.array([500, 850])
  sshape = (len(LON10), len(LAT10))
  truth = rs.standard_normal((n_time, len(levels)) + sshape)
  forecast = rs.standard_normal((n_member, n_time, len(levels)) + sshape)
  if ties:
    truth, forecast = np.round(truth, 1), np.round(forecast, 1)
    forecast[1] = forecast[0]  # two identical members everywhere
  return {
      'coords': {'time': _times(n_time), 'level': levels, 'latitude': LAT10,
                 'longitude': LON10, 'realization': np.arange(n_member)},
      'truth': {'geopotential': _arr(truth.astype(dtype), 'time', 'level',
                                     'longitude', 'latitude')},
      'forecast': {'geopotential': _arr(
          forecast.astype(dtype), 'realization', 'time', 'level', 'longitude',
          'latitude')},
  }


def tier2_table():
  small = ['global', 'europe', 'land_thr']
  return {
      'gauss_f64': (lambda: gauss_case(np.float64), GAUSS_METRICS, small,
                    False, 'chunk'),
      'gauss_f32': (lambda: gauss_case(np.float32, seed=61), GAUSS_METRICS,
                    ['global', 'europe'], False, 'chunk'),
      'ensthr_f64': (lambda: ens_threshold_case(np.float64),
                     ENS_THRESHOLD_METRICS, small, False, 'chunk'),
      'ensthr_f32_nan_skipna': (
          lambda: ens_threshold_case(np.float32, nan_frac=0.03, seed=62),
          ENS_THRESHOLD_METRICS, ['global', 'europe'], True, 'chunk'),
      'energy_f32': (lambda: ens_case(np.float32, n_member=4, seed=63),
                     ENERGY_METRICS, ['global', 'europe'], False, 'chunk'),
      'energy_f64': (lambda: ens_case(np.float64, n_member=3, seed=64),
                     ENERGY_METRICS, ['global'], False, 'chunk'),
      'seeps_f32': (lambda: seeps_case(np.float32), SEEPS_METRICS,
                    ['global', 'europe'], False, 'chunk'),
      'seeps_f64': (lambda: seeps_case(np.float64, seed=65), SEEPS_METRICS,
                    ['global'], False, 'chunk'),
      'rank_f32_ties': (lambda: rank_case(np.float32), RANK_METRICS,
                        ['global'], False, 'chunk'),
      'rank_f64_ties': (lambda: rank_case(np.float64, seed=66), RANK_METRICS,
                        ['global'], False, 'chunk'),
      'rank_f32_compute': (lambda: rank_case(np.float32, ties=False, seed=67),
                           RANK_METRICS, ['global'], False, 'compute'),
  }


# --------------------------------------------------------------------------
# the metric x region loop (evaluation.py:388-438)
# --------------------------------------------------------------------------
def loop_case(dtype, seed=71):
  rs = np.random.RandomState(seed)
  n_time, levels = 4, np.array([500, 850])
  names = ('u_component_of_wind', 'v_component_of_wind', 'geopotential')
  dims = ('time', 'level', 'longitude', 'latitude')
  shape = (n_time, len(levels), len(LON10), len(LAT10))
  hours, days = np.array([0, 12]), np.arange(57, 63)
  truth = {k: _arr(rs.standard_normal(shape).astype(dtype), *dims)
           for k in names}
  forecast = {k: _arr((truth[k]['data'] + 0.6 * rs.standard_normal(shape)
                       ).astype(dtype), *dims) for k in names}
  clim = {k: _arr((0.3 * rs.standard_normal(
      (len(hours), len(days)) + shape[1:])).astype(dtype), 'hour', 'dayofyear',
                  *dims[1:]) for k in names}
  return {'coords': {'time': _times(n_time), 'level': levels,
                     'latitude': LAT10, 'longitude': LON10, 'hour': hours,
                     'dayofyear': days},
          'truth': truth, 'forecast': forecast, 'climatology': clim,
          'lsm': _arr(rs.rand(len(LAT10), len(LON10)), 'latitude',
                      'longitude')}


# insertion order is deliberately NOT alphabetical: xr.merge joins the `metric`
# index with an outer join, whose union comes out sorted
LOOP_METRICS = {
    'mse': WIND_METRICS['mse_wind'],
    'acc': DET_METRICS['acc'],
    'mae': DET_METRICS['mae'],
    'bias': DET_METRICS['bias'],
}
LOOP_REGIONS = ('global', 'tropics', 'extra_tropical', 'europe')


# --------------------------------------------------------------------------
# by-init layout (evaluation.py:474-477): forecast (init_time, lead_time, ...)
# with a 2-D valid_time coordinate, truth selected at the valid times
# --------------------------------------------------------------------------
def by_init_case(dtype, seed=81):
  rs = np.random.RandomState(seed)
  n_init, n_lead, levels = 3, 3, np.array([500, 850])
  sshape = (len(LON10), len(LAT10))
  step = np.timedelta64(12, 'h').astype('timedelta64[ns]')
  init = _times(n_init)
  lead = np.arange(n_lead) * step
  valid = init[:, None] + lead[None, :]
  times = _times(n_init + n_lead - 1)           # every valid time, once
  truth_full = rs.standard_normal((len(times), len(levels)) + sshape)
  idx = np.arange(n_init)[:, None] + np.arange(n_lead)[None, :]
  forecast = truth_full[idx] + 0.5 * rs.standard_normal(
      (n_init, n_lead, len(levels)) + sshape)
  hours, days = np.array([0, 12]), np.arange(57, 63)
  clim = 0.3 * rs.standard_normal((len(hours), len(days), len(levels)) + sshape)
  dims = ('init_time', 'lead_time', 'level', 'longitude', 'latitude')
  return {
      'coords': {'init_time': init, 'lead_time': lead, 'time': times,
                 'level': levels, 'latitude': LAT10, 'longitude': LON10,
                 'hour': hours, 'dayofyear': days},
      'extra_coords': {'valid_time': _arr(valid, 'init_time', 'lead_time')},
      'truth_full': {'geopotential': _arr(truth_full.astype(dtype), 'time',
                                          'level', 'longitude', 'latitude')},
      'truth': {'geopotential': _arr(truth_full[idx].astype(dtype), *dims)},
      'forecast': {'geopotential': _arr(forecast.astype(dtype), *dims)},
      'climatology': {'geopotential': _arr(clim.astype(dtype), 'hour',
                                           'dayofyear', 'level', 'longitude',
                                           'latitude')},
      'lsm': _arr(rs.rand(len(LAT10), len(LON10)), 'latitude', 'longitude'),
  }


def f32_coordinate_case(seed=82):
  """float32 latitude / longitude coordinates, as 0.25-degree ERA5 ships them:
  the latitude weights inherit float32 (metrics.py:41, 57) and xarray's dot of
  float32 data with float32 weights accumulates and returns float32."""
  case = det_case(np.float32, seed=seed)
  case['coords'] = dict(case['coords'],
                        latitude=LAT10.astype(np.float32),
                        longitude=LON10.astype(np.float32))
  return case


LAT025 = np.linspace(-90.0, 90.0, 721)
LON025 = np.linspace(0.0, 360.0, 1440, endpoint=False)


def f32_coordinate_case_era5(seed=84):
  """The same at the real size: one 0.25-degree grid (721 x 1440) with float32
  coordinates, (time, level, latitude, longitude) float32 data -- a million
  points per spatial sum, where the reference's float32 einsum shows its
  summation noise (profiles/NOTES.md 4 records the measured deviation)."""
  rs = np.random.RandomState(seed)
  n_time, levels = 2, np.array([500])
  sshape = (len(LAT025), len(LON025))
  spatial = ('latitude', 'longitude')
  truth = rs.standard_normal((n_time, 1) + sshape).astype(np.float32)
  forecast = (truth + 0.5 * rs.standard_normal(truth.shape)).astype(np.float32)
  days, hours = np.arange(57, 61), np.array([0, 12])
  clim = (0.3 * rs.standard_normal((len(hours), len(days), 1) + sshape)
          ).astype(np.float32)
  lsm = rs.rand(*sshape)
  lsm[lsm < 0.25] = 0.0
  return {
      'coords': {'time': _times(n_time), 'level': levels,
                 'latitude': LAT025.astype(np.float32),
                 'longitude': LON025.astype(np.float32),
                 'hour': hours, 'dayofyear': days},
      'truth': {'geopotential': _arr(truth, 'time', 'level', *spatial)},
      'forecast': {'geopotential': _arr(forecast, 'time', 'level', *spatial)},
      'climatology': {'geopotential': _arr(clim, 'hour', 'dayofyear', 'level',
                                           *spatial)},
      'lsm': _arr(lsm, 'latitude', 'longitude'),
  }


def layout_table():
  small = ['global', 'europe', 'land_thr']
  return {
      'by_init_f32_chunk': (lambda: by_init_case(np.float32), DET_METRICS,
                            small, False, 'chunk'),
      'by_init_f64_compute': (lambda: by_init_case(np.float64, seed=83),
                              DET_METRICS, ['global', 'tropics'], False,
                              'compute'),
      'det_f32_coords32': (f32_coordinate_case, DET_METRICS,
                           small + ['extra_tropical'], False, 'chunk'),
      'det_f32_era5_coords32': (f32_coordinate_case_era5, DET_METRICS,
                                small + ['extra_tropical'], False, 'chunk'),
  }


def ragged_case(seed=91):
  """Forecast and truth that only partly overlap: xarray arithmetic keeps the
  variables present in both and inner-joins the `time` index (forecast's order);
  truth has times and a variable the forecast lacks, and vice versa."""
  case = det_case(np.float32, seed=seed)
  rs = np.random.RandomState(seed + 1)
  keep = np.array([1, 3, 4, 5])
  f = case['forecast']['geopotential']
  t = case['truth']['geopotential']
  case['forecast'] = {
      'geopotential': _arr(f['data'][:, keep], *f['dims']),
      'temperature': _arr(rs.standard_normal(f['data'][:, keep].shape).astype(
          np.float32), *f['dims'])}
  case['truth'] = {
      'geopotential': t,
      'specific_humidity': _arr(rs.standard_normal(t['data'].shape).astype(
          np.float32), *t['dims'])}
  c = case['climatology']['geopotential']
  case['climatology'] = dict(case['climatology'], specific_humidity=_arr(
      rs.standard_normal(c['data'].shape).astype(np.float32), *c['dims']))
  case['coords_forecast'] = dict(case['coords'],
                                 time=case['coords']['time'][keep])
  return case


def ragged_table():
  return {'ragged_f32': (ragged_case, DET_METRICS, ['global', 'europe'], False,
                         'chunk')}


# --------------------------------------------------------------------------
# _evaluate_all_metrics with its baseline substitutions (evaluation.py:441-483):
# forecast := climatology / probabilistic climatology / persistence.
# A case carries the OPENED datasets the reference's driver would have read
# (`forecast`, `truth`, `climatology`, each with its own coordinates) and the
# Eval / Data switches.
# --------------------------------------------------------------------------
LAT30 = np.linspace(-75.0, 75.0, 6)
LON45 = np.linspace(0.0, 360.0, 8, endpoint=False)
EVALALL_REGIONS = ('global', 'tropics', 'extra_tropical')
EVALALL_DET = {k: DET_METRICS[k] for k in ('mse', 'rmse', 'mae', 'bias')}


def evalall_case(kind: str, dtype=np.float32, seed=71):
  """kind: 'clim_byinit' | 'clim_byvalid' | 'persist_byvalid' |
  'persist_byinit_chunk' | 'probclim_byinit' | 'clim_chunk_mean_names'."""
  rs = np.random.RandomState(seed + sum(map(ord, kind)))
  levels = np.array([500, 850])
  sshape = (len(LAT30), len(LON45))
  spatial = ('latitude', 'longitude')
  h = np.timedelta64(1, 'h').astype('timedelta64[ns]')
  by_init = 'byinit' in kind or kind == 'clim_chunk_mean_names'
  if kind == 'probclim_byinit':
    # two years and a bit of 12-hourly truth: 2019 has no day 366, 2020 has
    t0, t1 = np.datetime64('2018-12-20T00', 'ns'), np.datetime64(
        '2021-01-12T00', 'ns')
  else:
    # crosses the end of a leap year: dayofyear 364, 365, 366, 1, 2, ...
    t0, t1 = np.datetime64('2020-12-26T00', 'ns'), np.datetime64(
        '2021-01-08T00', 'ns')
  times = np.arange(t0, t1, 12 * h)
  truth = rs.standard_normal((len(times), len(levels)) + sshape).astype(dtype)
  lead = np.arange(3) * 12 * h
  n_init = 6
  first = int(np.nonzero(times == np.datetime64('2020-12-28T00', 'ns'))[0][0])
  init = times[first:first + n_init]
  case = {
      'kind': kind, 'by_init': by_init,
      'coords_truth': {'time': times, 'level': levels, 'latitude': LAT30,
                       'longitude': LON45},
      'truth': {'geopotential': _arr(truth, 'time', 'level', *spatial)},
  }
  if by_init:
    f = rs.standard_normal((n_init, len(lead), len(levels)) + sshape)
    case['coords_forecast'] = {'init_time': init, 'lead_time': lead,
                               'level': levels, 'latitude': LAT30,
                               'longitude': LON45}
    case['forecast'] = {'geopotential': _arr(
        f.astype(dtype), 'init_time', 'lead_time', 'level', *spatial)}
    case['forecast_extra_coords'] = {
        'valid_time': _arr(init[:, None] + lead[None, :], 'init_time',
                           'lead_time')}
  else:
    ftime = times[first:first + n_init + 2]
    f = rs.standard_normal((len(ftime), len(lead), len(levels)) + sshape)
    case['coords_forecast'] = {'time': ftime, 'lead_time': lead,
                               'level': levels, 'latitude': LAT30,
                               'longitude': LON45}
    case['forecast'] = {'geopotential': _arr(
        f.astype(dtype), 'time', 'lead_time', 'level', *spatial)}
    case['forecast_extra_coords'] = {
        'init_time': _arr(ftime[:, None] - lead[None, :], 'time', 'lead_time')}
  hours = np.array([0, 12])
  days = np.arange(1, 367)
  cname = 'geopotential_mean' if kind == 'clim_chunk_mean_names' else (
      'geopotential')
  clim = 0.4 * rs.standard_normal((len(hours), len(days), len(levels)) + sshape)
  case['coords_climatology'] = {'hour': hours, 'dayofyear': days,
                                'level': levels, 'latitude': LAT30,
                                'longitude': LON45}
  case['climatology'] = {cname: _arr(clim.astype(dtype), 'hour', 'dayofyear',
                                     'level', *spatial)}
  # the ACC of the case reads a climatology under the plain name
  case['coords_acc_climatology'] = case['coords_climatology']
  case['acc_climatology'] = {'geopotential': _arr(
      clim.astype(dtype), 'hour', 'dayofyear', 'level', *spatial)}
  return case


def evalall_table():
  """name -> (case builder, eval switches, metric table, region labels,
  skipna).  The `_chunk` kinds go through the Beam driver's per-chunk
  functions (evaluation.py:618-675) followed by _evaluate_chunk."""
  det_acc = dict(EVALALL_DET,
                 acc=lambda m, c: m.ACC(climatology=c['acc_climatology']))
  ens = {k: ENS_METRICS[k] for k in ('crps', 'crps_spread', 'crps_skill',
                                     'ensemble_mean_mse', 'ensemble_variance')}
  number = lambda table: {
      k: (lambda m, c, f=f: _with_ensemble_dim(f(m, c), 'number'))
      for k, f in table.items()}
  return {
      'evalall_clim_byinit': (
          lambda: evalall_case('clim_byinit'),
          {'evaluate_climatology': True}, det_acc, EVALALL_REGIONS, False),
      'evalall_clim_byvalid': (
          lambda: evalall_case('clim_byvalid'),
          {'evaluate_climatology': True}, EVALALL_DET, EVALALL_REGIONS, False),
      'evalall_persist_byvalid': (
          lambda: evalall_case('persist_byvalid', np.float64),
          {'evaluate_persistence': True}, det_acc, EVALALL_REGIONS, False),
      'evalall_probclim_byinit': (
          lambda: evalall_case('probclim_byinit'),
          {'evaluate_probabilistic_climatology': True,
           'probabilistic_climatology_start_year': 2019,
           'probabilistic_climatology_end_year': 2020,
           'probabilistic_climatology_hour_interval': 12},
          number(ens), ('global', 'tropics'), False),
      'evalall_probclim_byinit_skipna': (
          lambda: evalall_case('probclim_byinit', seed=72),
          {'evaluate_probabilistic_climatology': True,
           'probabilistic_climatology_start_year': 2019,
           'probabilistic_climatology_end_year': 2020,
           'probabilistic_climatology_hour_interval': 12},
          number(ens), ('global',), True),
      'evalall_persist_byinit_chunk': (
          lambda: evalall_case('persist_byinit_chunk'),
          {'evaluate_persistence': True}, det_acc, EVALALL_REGIONS, False),
      'evalall_clim_chunk_mean_names': (
          lambda: evalall_case('clim_chunk_mean_names'),
          {'evaluate_climatology': True}, EVALALL_DET, EVALALL_REGIONS, False),
  }


def _with_ensemble_dim(metric, dim):
  metric.ensemble_dim = dim
  return metric
