"""NumPy restatement of weatherbench2/metrics.py hot path (TEST INFRASTRUCTURE).

Each function cites the reference lines (/root/reference/weatherbench2/) it
follows.  Third-party behaviour restated:

* ``Dataset.weighted(w).mean(dims, skipna)`` (xarray core/weighted.py) =
  ``dot(x.fillna(0) if skipna else x, w) / dot(notnull(x), w)`` with a zero
  sum-of-weights mapped to NaN; ``dot`` is ``np.einsum`` in the promoted dtype;
* ``mean/var/std`` over members stay in the input dtype;
* ``int64 * float32 -> float64`` (the CRPS rank product).
"""
from __future__ import annotations

import dataclasses
import typing as t

import numpy as np

from oracle.named import DS, NA
from oracle.regions_np import Region

REALIZATION = 'realization'


# ---------------------------------------------------------------------------
# metrics.py:35-60
# ---------------------------------------------------------------------------
def _assert_increasing(x: np.ndarray):
  if not (np.diff(x) > 0).all():
    raise ValueError(f'array is not increasing: {x}')


def _latitude_cell_bounds(x: np.ndarray) -> np.ndarray:
  pi_over_2 = np.array([np.pi / 2], dtype=x.dtype)
  return np.concatenate([-pi_over_2, (x[:-1] + x[1:]) / 2, pi_over_2])


def _cell_area_from_latitude(points: np.ndarray) -> np.ndarray:
  bounds = _latitude_cell_bounds(points)
  _assert_increasing(bounds)
  upper = bounds[1:]
  lower = bounds[:-1]
  return np.sin(upper) - np.sin(lower)


def get_lat_weights(latitude: np.ndarray) -> NA:
  """metrics.py:55-60. Weights inherit the latitude coord dtype."""
  weights = _cell_area_from_latitude(np.deg2rad(np.asarray(latitude)))
  weights = weights / np.mean(weights)
  return NA(weights, ('latitude',))


# ---------------------------------------------------------------------------
# xarray DatasetWeighted.mean, restated
# ---------------------------------------------------------------------------
def _weighted_mean(x: NA, w: NA, skipna: bool) -> NA:
  dims = ('latitude', 'longitude')
  rest = tuple(d for d in x.dims if d not in dims)
  xd = x.transpose(*rest, *dims).data
  # weights broadcast over (latitude, longitude)
  ones = NA(np.ones((x.sizes['latitude'], x.sizes['longitude']),
                    dtype=w.dtype), dims)
  wd = (ones * w).transpose(*dims).data
  mask = ~np.isnan(xd)
  if skipna:
    xd = np.where(mask, xd, np.zeros((), xd.dtype))
  with np.errstate(all='ignore'):
    num = np.einsum('...ij,ij->...', xd, wd)
    den = np.einsum('...ij,ij->...', mask.astype(wd.dtype), wd)
    den = np.where(den != 0.0, den, np.nan)
    out = num / den
  return NA(out, rest)


def spatial_average(dataset: DS, region: t.Optional[Region],
                    skipna: bool) -> DS:
  """metrics.py:141-163."""
  weights = get_lat_weights(dataset.coord('latitude'))
  if region is not None:
    dataset, weights = region.apply(dataset, weights)
    # metrics.py:159-160  ignore NaN/Inf values in regions with zero weight
    dataset = dataset.map(lambda v: v.where(weights > 0, 0))
  coords = {k: c for k, c in dataset.coords.items()
            if k not in ('latitude', 'longitude')}
  return DS({k: _weighted_mean(v, weights, skipna)
             for k, v in dataset.items()}, coords)


def spatial_average_l2_norm(dataset, region, skipna):
  """metrics.py:166-172."""
  return spatial_average(dataset ** 2, region=region, skipna=skipna).sqrt()


# ---------------------------------------------------------------------------
# Metric base, metrics.py:84-138
# ---------------------------------------------------------------------------
@dataclasses.dataclass
class Metric:

  def compute_chunk(self, forecast, truth, region=None, skipna=False) -> DS:
    raise NotImplementedError

  def compute(self, forecast, truth, region=None, skipna=False) -> DS:
    if 'time' in forecast.dims:
      avg_dim = 'time'
    elif 'init_time' in forecast.dims:
      avg_dim = 'init_time'
    else:
      raise ValueError(
          f'Forecast has neither valid_time or init_time dimension {forecast}')
    return self.compute_chunk(
        forecast, truth, region=region, skipna=skipna).mean(
            avg_dim, skipna=skipna)


# ---------------------------------------------------------------------------
# Deterministic metrics, metrics.py:175-414
# ---------------------------------------------------------------------------
@dataclasses.dataclass
class WindVectorMSE(Metric):
  u_name: str
  v_name: str
  vector_name: str

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    diff = forecast - truth
    ds = DS({self.vector_name: diff[self.u_name] ** 2 + diff[self.v_name] ** 2},
            diff.coords)
    return spatial_average(ds, region=region, skipna=skipna)


@dataclasses.dataclass
class WindVectorRMSESqrtBeforeTimeAvg(Metric):
  u_name: str
  v_name: str
  vector_name: str

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    return WindVectorMSE(self.u_name, self.v_name, self.vector_name
                         ).compute_chunk(forecast, truth, region, skipna).sqrt()


@dataclasses.dataclass
class RMSESqrtBeforeTimeAvg(Metric):
  wind_vector_rmse: t.Optional[list] = None

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    results = spatial_average_l2_norm(forecast - truth, region, skipna)
    if self.wind_vector_rmse is not None:
      for wv in self.wind_vector_rmse:
        results[wv.vector_name] = wv.compute_chunk(
            forecast, truth, region=region, skipna=skipna)[wv.vector_name]
    return results


@dataclasses.dataclass
class MSE(Metric):
  wind_vector_mse: t.Optional[list] = None

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    results = spatial_average((forecast - truth) ** 2, region, skipna)
    if self.wind_vector_mse is not None:
      for wv in self.wind_vector_mse:
        results[wv.vector_name] = wv.compute_chunk(
            forecast, truth, region=region, skipna=skipna)[wv.vector_name]
    return results


@dataclasses.dataclass
class SpatialMSE(Metric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    return (forecast - truth) ** 2


@dataclasses.dataclass
class MAE(Metric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    return spatial_average(abs(forecast - truth), region, skipna)


@dataclasses.dataclass
class SpatialMAE(Metric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    return abs(forecast - truth)


@dataclasses.dataclass
class Bias(Metric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    return spatial_average(forecast - truth, region, skipna)


@dataclasses.dataclass
class SpatialBias(Metric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    return forecast - truth


def _dayofyear_hour(times: np.ndarray):
  import pandas as pd
  times = times.data if isinstance(times, NA) else times  # 2-D valid_time
  idx = pd.DatetimeIndex(np.asarray(times).ravel())
  shape = np.shape(times)
  return (np.asarray(idx.dayofyear).reshape(shape),
          np.asarray(idx.hour).reshape(shape))


def get_climatology_chunk(climatology: DS, truth: DS) -> DS:
  """metrics.py:63-81 (falls back to '<var>_mean' names)."""
  names = list(truth.keys())
  if all(k in climatology for k in names):
    return climatology.select_vars(names)
  clim_var_dict = {str(k) + '_mean': k for k in names}
  not_found = set(names).difference(climatology.keys())
  not_found_means = set(clim_var_dict).difference(climatology.keys())
  if not_found and not_found_means:
    raise KeyError(
        f"Did not find {not_found} keys in climatology. Appending 'mean' did"
        ' not help.')
  return climatology.select_vars(list(clim_var_dict)).rename_vars(clim_var_dict)


def select_climatology(climatology_chunk: DS, forecast: DS) -> DS:
  """The gather of metrics.py:394-404: sel(level), sel(dayofyear[, hour])."""
  if 'init_time' in forecast.dims:
    valid = forecast.coords['valid_time']  # 2-D (init_time, lead_time)
    tdims = ('init_time', 'lead_time')
  else:
    valid = forecast.coords['time']
    tdims = ('time',)
  doy, hour = _dayofyear_hour(valid)
  doy_pos = {v: i for i, v in enumerate(
      climatology_chunk.coord('dayofyear').tolist())}
  doy_idx = np.vectorize(doy_pos.__getitem__, otypes=[np.int64])(doy)
  has_hour = 'hour' in climatology_chunk.coords
  if has_hour:
    hour_pos = {v: i for i, v in enumerate(
        climatology_chunk.coord('hour').tolist())}
    hour_idx = np.vectorize(hour_pos.__getitem__, otypes=[np.int64])(hour)
  level_idx = None
  if 'level' in forecast.coords and 'level' in climatology_chunk.coords:
    pos = {v: i for i, v in enumerate(climatology_chunk.coord('level').tolist())}
    level_idx = np.array([pos[v] for v in forecast.coord('level').tolist()])

  def gather(v: NA) -> NA:
    if level_idx is not None and 'level' in v.dims:
      v = v.isel(level=level_idx)
    rest = tuple(d for d in v.dims if d not in ('dayofyear', 'hour'))
    if has_hour and 'hour' in v.dims:
      data = v.transpose('dayofyear', 'hour', *rest).data[doy_idx, hour_idx]
    else:
      data = v.transpose('dayofyear', *rest).data[doy_idx]
    return NA(data, tdims + rest)

  coords = {k: c for k, c in forecast.coords.items()}
  return DS({k: gather(v) for k, v in climatology_chunk.items()}, coords)


@dataclasses.dataclass
class ACC(Metric):
  """metrics.py:377-414."""
  climatology: DS

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    climatology_chunk = get_climatology_chunk(self.climatology, truth)
    climatology_chunk = select_climatology(climatology_chunk, forecast)
    forecast_anom = forecast - climatology_chunk
    truth_anom = truth - climatology_chunk
    num = spatial_average(forecast_anom * truth_anom, region, skipna)
    den = (spatial_average(forecast_anom ** 2, region, skipna)
           * spatial_average(truth_anom ** 2, region, skipna)).sqrt()
    return num / den


# ---------------------------------------------------------------------------
# Ensemble metrics, metrics.py:532-846, 1161-1399
# ---------------------------------------------------------------------------
def _get_n_ensemble(ds: DS, ensemble_dim: str,
                    expect_n_ensemble_at_least: int = 1) -> int:
  if ensemble_dim not in ds.dims:
    raise ValueError(f'{ensemble_dim=} not found in {ds.dims=}')
  n_ensemble = ds.sizes[ensemble_dim]
  if n_ensemble < expect_n_ensemble_at_least:
    raise ValueError(f'{n_ensemble=} is less than expected size of '
                     f'{expect_n_ensemble_at_least}')
  return n_ensemble


def rankdata(x: np.ndarray, axis: int) -> np.ndarray:
  """metrics.py:837-846."""
  x = np.asarray(x)
  x = np.swapaxes(x, axis, -1)
  j = np.argsort(x, axis=-1)
  ordinal_ranks = np.broadcast_to(
      np.arange(1, x.shape[-1] + 1, dtype=int), x.shape)
  ordered_ranks = np.empty(j.shape, dtype=ordinal_ranks.dtype)
  np.put_along_axis(ordered_ranks, j, ordinal_ranks, axis=-1)
  return np.swapaxes(ordered_ranks, axis, -1)


def _rank_ds(ds: DS, dim: str) -> DS:
  return DS({k: NA(rankdata(v.data, v.dims.index(dim)), v.dims)
             for k, v in ds.items()}, ds.coords)


def pointwise_crps_spread(forecast: DS, ensemble_dim: str, skipna: bool) -> DS:
  """metrics.py:781-813."""
  n_ensemble = _get_n_ensemble(forecast, ensemble_dim)
  if n_ensemble < 2:
    return forecast.isel(**{ensemble_dim: 0}).zeros_like()
  rank = _rank_ds(forecast, ensemble_dim)
  return (2 * ((2 * rank - n_ensemble - 1) * forecast).mean(
      ensemble_dim, skipna=skipna)) / (n_ensemble - 1)


def pointwise_crps_skill(forecast: DS, truth: DS, ensemble_dim: str,
                         skipna: bool) -> DS:
  """metrics.py:816-824."""
  _get_n_ensemble(forecast, ensemble_dim)
  return abs(truth - forecast).mean(ensemble_dim, skipna=skipna)


def debiased_ensemble_mean_mse(forecast, truth, ensemble_dim, skipna):
  """metrics.py:532-565."""
  forecast_mean = forecast.mean(ensemble_dim, skipna=skipna)
  forecast_var = forecast.var(ensemble_dim, skipna=skipna, ddof=1)
  biased_mse = (truth - forecast_mean) ** 2
  return biased_mse - forecast_var / _get_n_ensemble(forecast, ensemble_dim)


@dataclasses.dataclass
class EnsembleMetric(Metric):
  ensemble_dim: str = REALIZATION


@dataclasses.dataclass
class CRPSSpread(EnsembleMetric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    return spatial_average(
        pointwise_crps_spread(forecast, self.ensemble_dim, skipna),
        region, skipna)


@dataclasses.dataclass
class CRPSSkill(EnsembleMetric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    return spatial_average(
        pointwise_crps_skill(forecast, truth, self.ensemble_dim, skipna),
        region, skipna)


@dataclasses.dataclass
class CRPS(EnsembleMetric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    return CRPSSkill(self.ensemble_dim).compute_chunk(
        forecast, truth, region, skipna
    ) - 0.5 * CRPSSpread(self.ensemble_dim).compute_chunk(
        forecast, truth, region, skipna)


@dataclasses.dataclass
class SpatialCRPSSpread(EnsembleMetric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    return pointwise_crps_spread(forecast, self.ensemble_dim, skipna)


@dataclasses.dataclass
class SpatialCRPSSkill(EnsembleMetric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    return pointwise_crps_skill(forecast, truth, self.ensemble_dim, skipna)


@dataclasses.dataclass
class SpatialCRPS(EnsembleMetric):
  """metrics.py:718-739."""
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    return SpatialCRPSSkill(self.ensemble_dim).compute_chunk(
        forecast, truth, region, skipna
    ) - 0.5 * SpatialCRPSSpread(self.ensemble_dim).compute_chunk(
        forecast, truth, region, skipna)


@dataclasses.dataclass
class SpatialEnsembleVariance(EnsembleMetric):
  """metrics.py:1244-1266."""
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    n_ensemble = _get_n_ensemble(forecast, self.ensemble_dim)
    if n_ensemble == 1:
      return forecast.zeros_like().mean(self.ensemble_dim, skipna=skipna)
    return forecast.var(self.ensemble_dim, ddof=1, skipna=skipna)


@dataclasses.dataclass
class SpatialEnsembleMeanMSE(EnsembleMetric):
  """metrics.py:1366-1381."""
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    _get_n_ensemble(forecast, self.ensemble_dim)
    return (truth - forecast.mean(self.ensemble_dim, skipna=skipna)) ** 2


@dataclasses.dataclass
class DebiasedSpatialEnsembleMeanMSE(EnsembleMetric):
  """metrics.py:1384-1399."""
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    _get_n_ensemble(forecast, self.ensemble_dim)
    return debiased_ensemble_mean_mse(forecast, truth, self.ensemble_dim,
                                      skipna)


def _zeros_like_spatial_mean(forecast, ensemble_dim, region, skipna):
  """metrics.py:1196-1204 / 1228-1235 (the n_ensemble == 1 branch)."""
  return spatial_average(forecast, region, skipna).mean(
      ensemble_dim, skipna=skipna).zeros_like()


@dataclasses.dataclass
class EnsembleStddevSqrtBeforeTimeAvg(EnsembleMetric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    n_ensemble = _get_n_ensemble(forecast, self.ensemble_dim)
    if n_ensemble == 1:
      return _zeros_like_spatial_mean(forecast, self.ensemble_dim, region,
                                      skipna)
    return spatial_average_l2_norm(
        forecast.std(self.ensemble_dim, ddof=1, skipna=skipna), region, skipna)


@dataclasses.dataclass
class EnsembleVariance(EnsembleMetric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    n_ensemble = _get_n_ensemble(forecast, self.ensemble_dim)
    if n_ensemble == 1:
      return _zeros_like_spatial_mean(forecast, self.ensemble_dim, region,
                                      skipna)
    return spatial_average(
        forecast.var(self.ensemble_dim, ddof=1, skipna=skipna), region, skipna)


@dataclasses.dataclass
class EnsembleMeanRMSESqrtBeforeTimeAvg(EnsembleMetric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    _get_n_ensemble(forecast, self.ensemble_dim)
    return spatial_average_l2_norm(
        truth - forecast.mean(self.ensemble_dim, skipna=skipna), region,
        skipna)


@dataclasses.dataclass
class EnsembleMeanMSE(EnsembleMetric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    _get_n_ensemble(forecast, self.ensemble_dim)
    return spatial_average(
        (truth - forecast.mean(self.ensemble_dim, skipna=skipna)) ** 2,
        region, skipna)


@dataclasses.dataclass
class DebiasedEnsembleMeanMSE(EnsembleMetric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    _get_n_ensemble(forecast, self.ensemble_dim)
    return spatial_average(
        debiased_ensemble_mean_mse(forecast, truth, self.ensemble_dim, skipna),
        region, skipna)


# ---------------------------------------------------------------------------
# Tier 2: Gaussian CRPS / variance (metrics.py:849-937), EnergyScore (:1402-1517)
# ---------------------------------------------------------------------------
def pointwise_gaussian_crps(forecast: DS, truth: DS) -> DS:
  """metrics.py:869-905.  scipy's norm.cdf/pdf promote to float64."""
  from scipy import stats
  dataset = {}
  for var_name in [v for v in forecast.keys() if f'{v}_std' in forecast.keys()]:
    fm = DS({var_name: forecast[var_name]}, forecast.coords)
    diff = (fm - truth)[var_name]
    std = forecast[f'{var_name}_std']
    if diff.dims != std.dims or diff.shape != std.shape:
      # std follows the forecast; restrict it like the aligned difference
      a, _ = __import__('oracle.named', fromlist=['align_inner']).align_inner(
          DS({var_name: std}, forecast.coords), truth)
      std = a[var_name]
    norm_diff = diff / std
    with np.errstate(all='ignore'):
      inner = (norm_diff * NA(2 * stats.norm.cdf(norm_diff.data) - 1,
                              norm_diff.dims)
               + NA(2 * stats.norm.pdf(norm_diff.data), norm_diff.dims)
               - 1 / np.sqrt(np.pi))
    dataset[var_name] = std * inner
  return DS(dataset, (fm - truth).coords)


@dataclasses.dataclass
class GaussianCRPS(Metric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    return spatial_average(pointwise_gaussian_crps(forecast, truth), region,
                           skipna)


@dataclasses.dataclass
class GaussianVariance(Metric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    dataset = {}
    for var_name in [v for v in forecast.keys()
                     if f'{v}_std' in forecast.keys()]:
      dataset[var_name] = forecast[f'{var_name}_std'] * forecast[
          f'{var_name}_std']
    return spatial_average(DS(dataset, forecast.coords), region, skipna)


def _ensemble_slice(ds: DS, dim: str, sl: slice) -> DS:
  """metrics.py:591-596: isel + relabel 0..n-1."""
  out = ds.isel(**{dim: sl})
  out.coords[dim] = np.arange(out.sizes[dim])
  return out


@dataclasses.dataclass
class EnergyScoreSpread(EnsembleMetric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    n_ensemble = _get_n_ensemble(forecast, self.ensemble_dim)
    if n_ensemble == 1:
      return _zeros_like_spatial_mean(forecast, self.ensemble_dim, region,
                                      skipna)
    return spatial_average_l2_norm(
        _ensemble_slice(forecast, self.ensemble_dim, slice(None, -1))
        - _ensemble_slice(forecast, self.ensemble_dim, slice(1, None)),
        region, skipna).mean(self.ensemble_dim, skipna=skipna)


@dataclasses.dataclass
class EnergyScoreSkill(EnsembleMetric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    _get_n_ensemble(forecast, self.ensemble_dim)
    return spatial_average_l2_norm(forecast - truth, region, skipna).mean(
        self.ensemble_dim, skipna=skipna)


@dataclasses.dataclass
class EnergyScore(EnsembleMetric):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    return EnergyScoreSkill(self.ensemble_dim).compute_chunk(
        forecast, truth, region, skipna
    ) - 0.5 * EnergyScoreSpread(self.ensemble_dim).compute_chunk(
        forecast, truth, region, skipna)


# ---------------------------------------------------------------------------
# Tier 2: threshold metrics (metrics.py:940-1158 Gaussian, :1524-1891 ensemble)
# ---------------------------------------------------------------------------
def _where01(cond: NA) -> NA:
  """xr.where(cond, 1.0, 0.0) -> float64."""
  return NA(np.where(cond.data, 1.0, 0.0), cond.dims)


def _ds_binary(a: DS, b: DS, fn) -> DS:
  from oracle.named import align_inner
  a, b = align_inner(a, b)
  names = [k for k in a.keys() if k in b.keys()]
  return DS({k: fn(a[k], b[k]) for k in names}, {**b.coords, **a.coords})


def _gauss_vars(forecast: DS):
  return [v for v in forecast.keys() if f'{v}_std' in forecast.keys()]


def _norm_threshold(forecast: DS, threshold: DS, var: str) -> NA:
  # threshold - forecast: dims of the threshold come first
  return (threshold[var] - forecast[var]) / forecast[f'{var}_std']


def _cdf(x: NA) -> NA:
  from scipy import stats
  return NA(stats.norm.cdf(x.data), x.dims)


def compute_gaussian_brier_score(forecast, truth, threshold):
  """metrics.py:975-1000."""
  tp = _ds_binary(truth, threshold, lambda a, b: _where01(a > b))
  fp = DS({v: 1 - _cdf(_norm_threshold(forecast, threshold, v))
           for v in _gauss_vars(forecast)}, forecast.coords)
  return (fp - tp) ** 2


def compute_gaussian_ignorance_score(forecast, truth, threshold):
  """metrics.py:1043-1066."""
  tp = _ds_binary(truth, threshold, lambda a, b: _where01(a > b))
  out = {}
  for v in _gauss_vars(forecast):
    cdf = _cdf(_norm_threshold(forecast, threshold, v))
    with np.errstate(all='ignore'):
      a, b, dims = NA._align(tp[v], cdf)
      out[v] = NA(-np.where(a, np.log(1 - b), np.log(b)), dims)
  return DS(out, forecast.coords)


def compute_gaussian_rps_part(forecast, truth, threshold):
  """metrics.py:1104-1121."""
  te = _ds_binary(truth, threshold, lambda a, b: _where01(a < b))
  fc = DS({v: _cdf(_norm_threshold(forecast, threshold, v))
           for v in _gauss_vars(forecast)}, forecast.coords)
  return (fc - te) ** 2


def _nan_where_null(x: NA, values: NA) -> NA:
  a, b, dims = NA._align(x, values)
  return NA(np.where(np.isnan(a), np.nan, b), dims)


def compute_brier_score(forecast, truth, threshold, ensemble_dim, debias,
                        skipna):
  """metrics.py:1524-1560."""
  tp = _ds_binary(truth, threshold,
                  lambda a, b: _nan_where_null(a, _where01(a > b)))
  fp = _ds_binary(forecast, threshold,
                  lambda a, b: _nan_where_null(a, _where01(a > b)))
  if debias:
    return debiased_ensemble_mean_mse(fp, tp, ensemble_dim, skipna)
  return (fp.mean(ensemble_dim, skipna=skipna) - tp) ** 2


def compute_ignorance_score(forecast, truth, threshold, ensemble_dim, skipna):
  """metrics.py:1720-1738."""
  tp = _ds_binary(truth, threshold, lambda a, b: _where01(a > b))
  fp = _ds_binary(forecast, threshold, lambda a, b: _where01(a > b)).mean(
      ensemble_dim, skipna=skipna)
  out = {}
  for v in fp.keys():
    with np.errstate(all='ignore'):
      a, b, dims = NA._align(tp[v], fp[v])
      out[v] = NA(-np.where(a, np.log(b), np.log(1 - b)), dims)
  return DS(out, fp.coords)


def compute_rps_part(forecast, truth, threshold, ensemble_dim, skipna):
  """metrics.py:1791-1802."""
  te = _ds_binary(truth, threshold, lambda a, b: _where01(a < b))
  fe = _ds_binary(forecast, threshold, lambda a, b: _where01(a < b)).mean(
      ensemble_dim, skipna=skipna)
  return (fe - te) ** 2


@dataclasses.dataclass
class ThresholdMetric(Metric):
  thresholds: t.Sequence = ()
  _score = None
  _sum_over_quantile = False
  _spatial_agg = True

  def _score_fn(self, skipna):
    return self._score

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    """metrics.py:947-972 (_map_over_thresholds)."""
    scores = []
    for threshold in self.thresholds:
      threshold_ds = threshold.compute(truth)
      score = self._score_fn(skipna)(forecast, truth, threshold_ds)
      if self._spatial_agg:
        score = spatial_average(score, region=region, skipna=skipna)
      scores.append(score)
    out = DS({k: NA(np.stack([s[k].data for s in scores]),
                    ('quantile',) + scores[0][k].dims)
              for k in scores[0].keys()},
             {**scores[0].coords,
              'quantile': np.array([th.quantile for th in self.thresholds])})
    if self._sum_over_quantile:
      coords = {k: c for k, c in out.coords.items() if k != 'quantile'}
      out = DS({k: v.sum('quantile') for k, v in out.items()}, coords)
    return out


@dataclasses.dataclass
class GaussianBrierScore(ThresholdMetric):
  _score = staticmethod(compute_gaussian_brier_score)


@dataclasses.dataclass
class GaussianIgnoranceScore(ThresholdMetric):
  _score = staticmethod(compute_gaussian_ignorance_score)


@dataclasses.dataclass
class GaussianRPS(ThresholdMetric):
  _score = staticmethod(compute_gaussian_rps_part)
  _sum_over_quantile = True


@dataclasses.dataclass
class _EnsembleThresholdMetric(ThresholdMetric):
  ensemble_dim: str = REALIZATION


@dataclasses.dataclass
class EnsembleBrierScore(_EnsembleThresholdMetric):
  def _score_fn(self, skipna):
    return lambda f, t_, th: compute_brier_score(
        f, t_, th, self.ensemble_dim, False, skipna)


@dataclasses.dataclass
class DebiasedEnsembleBrierScore(_EnsembleThresholdMetric):
  def _score_fn(self, skipna):
    return lambda f, t_, th: compute_brier_score(
        f, t_, th, self.ensemble_dim, True, skipna)


@dataclasses.dataclass
class EnsembleIgnoranceScore(_EnsembleThresholdMetric):
  def _score_fn(self, skipna):
    return lambda f, t_, th: compute_ignorance_score(
        f, t_, th, self.ensemble_dim, skipna)


@dataclasses.dataclass
class EnsembleRPS(_EnsembleThresholdMetric):
  _sum_over_quantile = True

  def _score_fn(self, skipna):
    return lambda f, t_, th: compute_rps_part(
        f, t_, th, self.ensemble_dim, skipna)


@dataclasses.dataclass
class SpatialEnsembleBrierScore(EnsembleBrierScore):
  """metrics.py:1615-1638."""
  _spatial_agg = False


@dataclasses.dataclass
class SpatialDebiasedEnsembleBrierScore(DebiasedEnsembleBrierScore):
  """metrics.py:1697-1719."""
  _spatial_agg = False


@dataclasses.dataclass
class SpatialEnsembleIgnoranceScore(EnsembleIgnoranceScore):
  """metrics.py:1780-1802."""
  _spatial_agg = False


@dataclasses.dataclass
class SpatialEnsembleRPS(EnsembleRPS):
  """metrics.py:1870-1891."""
  _spatial_agg = False


# ---------------------------------------------------------------------------
# Tier 2: SEEPS (metrics.py:417-524)
# ---------------------------------------------------------------------------
@dataclasses.dataclass
class SpatialSEEPS(Metric):
  climatology: DS = None
  dry_threshold_mm: float = 0.25
  precip_name: str = 'total_precipitation_24hr'
  min_p1: float = 0.1
  max_p1: float = 0.85

  @property
  def p1(self) -> NA:
    # metrics.py:443-444: `.mean(("hour", "dayofyear"))` with xarray's default
    # skipna=None, i.e. NaN dry fractions are skipped
    return self.climatology[f'{self.precip_name}_seeps_dry_fraction'].mean(
        ('hour', 'dayofyear'), skipna=True)

  def _valid_time(self, ds: DS) -> NA:
    vt = ds.coords['valid_time']
    return vt if isinstance(vt, NA) else NA(vt, ('time',))

  @staticmethod
  def _lift(x: NA, dims) -> np.ndarray:
    """x's data broadcastable against an array laid out as `dims`."""
    perm = [d for d in dims if d in x.dims]
    data = np.transpose(x.data, [x.dims.index(d) for d in perm])
    return data.reshape([x.sizes[d] if d in x.dims else 1 for d in dims])

  def _convert_precip_to_seeps_cat(self, ds: DS):
    """metrics.py:444-468 -> [dry, light, heavy] as 0/1 floats, NaN kept."""
    wet_threshold = self.climatology[f'{self.precip_name}_seeps_threshold']
    dry_threshold = self.dry_threshold_mm / 1000.0
    da = ds[self.precip_name]
    vt = self._valid_time(ds)
    doy, hour = _dayofyear_hour(vt.data)
    doy_pos = {v: i for i, v in enumerate(
        self.climatology.coord('dayofyear').tolist())}
    hour_pos = {v: i for i, v in enumerate(
        self.climatology.coord('hour').tolist())}
    di = np.vectorize(doy_pos.__getitem__, otypes=[np.int64])(doy)
    hi = np.vectorize(hour_pos.__getitem__, otypes=[np.int64])(hour)
    rest = tuple(d for d in wet_threshold.dims if d not in ('dayofyear',
                                                            'hour'))
    wet = NA(wet_threshold.transpose('dayofyear', 'hour', *rest).data[di, hi],
             vt.dims + rest)
    x = da.data
    w = self._lift(wet, da.dims)
    with np.errstate(invalid='ignore'):
      conds = [x < dry_threshold, np.logical_and(x > dry_threshold, x < w),
               x >= w]
    return [NA(np.where(np.isnan(x), np.nan, c.astype('int').astype(float)),
               da.dims) for c in conds]

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    fc = self._convert_precip_to_seeps_cat(forecast)
    tc = self._convert_precip_to_seeps_cat(truth)
    p1 = self.p1
    matrix = [[0 * p1, 1 / (1 - p1), 4 / (1 - p1)],
              [1 / p1, 0 * p1, 3 / (1 - p1)],
              [1 / p1 + 3 / (2 + p1), 3 / (2 + p1), 0 * p1]]
    result = None
    for i in range(3):
      for j in range(3):
        term = (fc[i] * tc[j]) * (0.5 * matrix[i][j])
        result = term if result is None else result + term
    p1l = self._lift(p1, result.dims)
    keep = np.logical_and(p1l < self.max_p1, p1l > self.min_p1)
    result = NA(np.where(keep, result.data, np.nan), result.dims)
    coords = {k: c for k, c in forecast.coords.items()}
    return DS({self.precip_name: result}, coords)


@dataclasses.dataclass
class SEEPS(SpatialSEEPS):
  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    result = super().compute_chunk(forecast, truth, region)
    return spatial_average(result, region=region, skipna=True)


# ---------------------------------------------------------------------------
# Tier 2: RankHistogram (metrics.py:1894-2042), central_reliability (:2045-2126)
# ---------------------------------------------------------------------------
class RankHistogram(EnsembleMetric):
  """One-hot encoding of truth's rank among the members (bins dim last)."""

  def __init__(self, ensemble_dim=REALIZATION, num_bins=None,
               break_ties_randomly=True, seed=None):
    super().__init__(ensemble_dim=ensemble_dim)
    self.num_bins = num_bins
    self._break_ties_randomly = break_ties_randomly
    self._seed = seed

  def _num_bins_actual(self, ensemble_size):
    default_n_bins = ensemble_size + 1
    if self.num_bins is None:
      return default_n_bins
    if default_n_bins % self.num_bins:
      raise ValueError(
          f'Cannot bin data with {ensemble_size=} into {self.num_bins} bins')
    return self.num_bins

  def _perturb(self, data: np.ndarray, idx: int) -> np.ndarray:
    """metrics.py:1955-1980.  `data` must be laid out like the reference's
    concatenated array: `np.random.default_rng(seed).uniform(size=da.shape)`
    consumes the stream in C order of THAT shape."""
    if data.shape[idx] < 2:
      return data
    with np.errstate(all='ignore'):
      diffs = np.diff(np.sort(data, axis=idx), axis=idx)
      diffs = np.where(diffs == 0, np.inf, diffs)
      min_diff = diffs.min(axis=idx, keepdims=True)
      size = np.where(min_diff < np.inf, min_diff / 2, 1)
    perturbation = np.random.default_rng(self._seed).uniform(
        size=data.shape, low=-size / 2, high=size / 2)
    return data + perturbation

  @staticmethod
  def _concat_dims(fdims, tdims, ensemble_dim):
    """Dim order of `xr.concat([truth, forecast], dim=ensemble_dim)`
    (metrics.py:2014) -- third-party behaviour, restated from xarray
    (core/concat.py, `_dataset_concat.ensure_common_dims`):
        common_dims = tuple(OrderedSet(d for v in vars for d in v.dims))
        if dim not in common_dims: common_dims = (dim,) + common_dims
    with vars = [truth variable, forecast variable]: the truth's dims come
    first, then the forecast's dims not seen yet IN THE FORECAST'S ORDER -- the
    ensemble dim (truth only has it as a scalar coordinate) and any dim only the
    forecast has (e.g. a lead time the truth was not expanded over).  The
    perturbation stream is consumed in C order of this layout."""
    out = list(tdims)
    for d in fdims:
      if d not in out:
        out.append(d)
    if ensemble_dim not in out:
      out.insert(0, ensemble_dim)
    return tuple(out)

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    out = {}
    for name in [k for k in forecast.keys() if k in truth.keys()]:
      f = forecast[name]
      t = truth[name]
      cdims = self._concat_dims(f.dims, t.dims, self.ensemble_dim)
      ax = cdims.index(self.ensemble_dim)
      rest = tuple(d for d in cdims if d != self.ensemble_dim)
      fd = f.transpose(*cdims).data                        # forecast in concat order
      td = NA._align(NA(np.zeros([n for i, n in enumerate(fd.shape) if i != ax]),
                        rest), t)[1]
      td = np.broadcast_to(td, [n for i, n in enumerate(fd.shape) if i != ax])
      td = np.expand_dims(td, ax).astype(
          np.result_type(t.data.dtype, fd.dtype), copy=False)
      combined = np.concatenate([td, fd], axis=ax)  # truth prepended
      if self._break_ties_randomly:
        combined = self._perturb(combined, ax)
      combined = np.moveaxis(combined, ax, 0)
      ensemble_size = fd.shape[ax]
      num_bins = self._num_bins_actual(ensemble_size)
      order = np.argsort(combined, axis=0)
      ranks = np.argmin(order, axis=0)  # where the truth (index 0) ended up
      factor = (ensemble_size + 1) // num_bins
      if factor != 1:
        ranks = ranks // factor
      out[name] = NA(np.eye(num_bins)[ranks], rest + ('bins',))
    coords = {k: c for k, c in forecast.coords.items() if k != self.ensemble_dim}
    coords['bins'] = np.arange(num_bins)
    return DS(out, coords)


def central_reliability(hist: np.ndarray) -> tuple:
  """metrics.py:2045-2126 on an array whose LAST axis is `bins`.
  Returns (probs[..., prob_index], desired_prob[prob_index])."""
  n_bins = hist.shape[-1]
  if n_bins < 3:
    raise ValueError(f'Too few bins. {n_bins=} but should be >= 3')
  left = hist[..., :n_bins // 2]
  right = hist[..., n_bins // 2 + n_bins % 2:]
  probs = np.cumsum(left[..., ::-1] + right, axis=-1)
  desired = np.ones(probs.shape[-1])
  if n_bins % 2:
    center = hist[..., n_bins // 2]
    probs = np.concatenate([center[..., None], center[..., None] + probs], -1)
    desired = np.concatenate(([0.5], desired))
  desired = np.cumsum(desired)
  return probs, desired / desired[-1]


def crps_brute_force(forecast: DS, truth: DS, skipna: bool) -> dict:
  """The reference TEST's O(M^2) eFAIR CRPS (metrics_test.py:896-920)."""

  def _l1_norm(x):
    return spatial_average(abs(x), region=None, skipna=skipna)

  n_ensemble = forecast.sizes[REALIZATION]
  skill = _l1_norm(truth - forecast).mean(REALIZATION, skipna=skipna)
  if n_ensemble == 1:
    spread = skill.zeros_like()
  else:
    dummy = DS({k: NA(v.data, tuple('dummy' if d == REALIZATION else d
                                    for d in v.dims))
                for k, v in forecast.items()}, forecast.coords)
    spread = _l1_norm(forecast - dummy).mean(
        (REALIZATION, 'dummy'), skipna=skipna) * (n_ensemble / (n_ensemble - 1))
  return {'score': skill - 0.5 * spread, 'spread': spread, 'skill': skill}
