"""Thresholds for the discrete probabilistic metrics (host side).

Same dataclasses/fields as weatherbench2/thresholds.py:90-197.  A threshold is
label work (climatology gather by dayofyear / hour / level, nearest-quantile
lookup) plus, for the Gaussian variant, `mean + norm.ppf(q) * std`; it is
evaluated on the host with the same NumPy/SciPy operations as the reference and
handed to the GPU kernels as one more input array.

Kept quirk: the day of year always comes from
truth['time'], the hour from truth[time_dim] (thresholds.py:140, 176).
"""
from __future__ import annotations

import dataclasses
import typing as t

import numpy as np
from scipy import stats

from weatherbench2_amd import xarray_lite as xl


def _coord(ds: xl.Dataset, name: str) -> np.ndarray:
  c = ds.coords[name]
  return np.asarray(c.values if isinstance(c, xl.DataArray) else c)


def _positions(have: np.ndarray, want: np.ndarray, what: str) -> np.ndarray:
  pos = {v: i for i, v in enumerate(np.asarray(have).tolist())}
  try:
    return np.array([pos[v] for v in np.asarray(want).tolist()], dtype=np.int64)
  except KeyError as e:
    raise KeyError(f'{what} label {e} not found in climatology') from e


def _time_gather(climatology: xl.Dataset, truth: xl.Dataset,
                 variables: dict) -> xl.Dataset:
  """climatology.sel(level=..., dayofyear=..., [hour=...]) at the truth's times.

  thresholds.py:131-142 / 166-178: the day of year comes from truth['time'],
  the hour from truth[time_dim] with time_dim = 'time' if it is a dimension of
  truth, else 'valid_time'.  For a by-init forecast the truth
  (`truth.sel(time=forecast.valid_time)`, evaluation.py:474) has dims
  (init_time, lead) and BOTH are 2-D coordinates over them; the gathered
  threshold then gets those dims, exactly like xarray's vectorised `.sel`."""
  import pandas as pd
  time_dim = 'time' if 'time' in truth.dims else 'valid_time'

  def labels(name):
    if name not in truth.coords:
      raise KeyError(name)  # what the reference's truth[name] raises
    c = truth.coords[name]
    if isinstance(c, xl.DataArray):
      return np.asarray(c.values), tuple(c.dims)
    return np.asarray(c), (name,)
  try:
    day_labels, tdims = labels('time')
  except KeyError:
    if time_dim == 'time':
      raise
    day_labels, tdims = labels('valid_time')  # truth carries only valid_time
  hour_labels, hdims = labels(time_dim)
  if hdims != tdims:
    hour_labels = np.transpose(hour_labels, [hdims.index(d) for d in tdims])
  shape = day_labels.shape
  doy = np.asarray(pd.DatetimeIndex(day_labels.ravel()).dayofyear)
  doy_idx = _positions(_coord(climatology, 'dayofyear'), doy,
                       'dayofyear').reshape(shape)
  has_hour = 'hour' in climatology.coords
  if has_hour:
    hour = np.asarray(pd.DatetimeIndex(hour_labels.ravel()).hour)
    hour_idx = _positions(_coord(climatology, 'hour'), hour,
                          'hour').reshape(shape)
  level_idx = None
  if 'level' in truth.dims and 'level' in climatology.coords:
    level_idx = _positions(_coord(climatology, 'level'), _coord(truth, 'level'),
                           'level')
  coords = {k: c for k, c in truth.coords.items()}
  out = xl.Dataset(coords=coords)
  for src, dst in variables.items():
    v = climatology[src]
    if level_idx is not None and 'level' in v.dims:
      v = v.isel(level=level_idx)
    rest = tuple(d for d in v.dims if d not in ('dayofyear', 'hour'))
    if has_hour and 'hour' in v.dims:
      data = v.transpose('dayofyear', 'hour', *rest).values[doy_idx, hour_idx]
    else:
      data = v.transpose('dayofyear', *rest).values[doy_idx]
    out.data_vars[dst] = xl.DataArray(data, tdims + rest, coords, dst)
  return out


@dataclasses.dataclass
class Threshold:
  """Threshold for discrete probabilistic metric evaluation
  (thresholds.py:90-113)."""

  climatology: t.Any
  quantile: float

  def compute(self, truth) -> xl.Dataset:
    raise NotImplementedError


@dataclasses.dataclass
class QuantileThreshold(Threshold):
  """Climatological quantile `<var>_quantile` nearest to `quantile` within
  0.01 (thresholds.py:116-148, 62-87)."""

  def compute(self, truth) -> xl.Dataset:
    given = truth  # xarray in, xarray out (the metrics call this with lite data)
    truth = xl.as_dataset(truth)
    climatology = xl.as_dataset(self.climatology)
    names = {str(k) + '_quantile': str(k) for k in truth.keys()}
    missing = set(names).difference(climatology.keys())
    if missing:
      raise KeyError(f'Did not find {missing} keys in climatology.')
    q = np.asarray(_coord(climatology, 'quantile'), dtype=float)
    i = int(np.argmin(np.abs(q - self.quantile)))
    if abs(q[i] - self.quantile) > 0.01:
      raise KeyError(f'Did not find quantiles {self.quantile}+-0.01 in '
                     'climatology. Consider increasing the tolerance or '
                     'recomputing the climatology.')
    return xl.like_input(_time_gather(climatology.isel(quantile=i), truth,
                                      names), given)


@dataclasses.dataclass
class GaussianQuantileThreshold(Threshold):
  """mean + norm.ppf(quantile) * std of the climatology
  (thresholds.py:151-187)."""

  def compute(self, truth) -> xl.Dataset:
    given = truth
    truth = xl.as_dataset(truth)
    climatology = xl.as_dataset(self.climatology)
    variables = [str(k) for k in truth.keys()]
    if all(v in climatology for v in variables):
      mean_names = {v: v for v in variables}
    else:
      mean_names = {v + '_mean': v for v in variables}
      missing = set(mean_names).difference(climatology.keys())
      if missing:
        raise KeyError(f'Did not find {set(variables)} keys in climatology. '
                       "Appending 'mean' did not help.")
    std_names = {v + '_std': v for v in variables}
    missing = set(std_names).difference(climatology.keys())
    if missing:
      raise KeyError(f'Did not find {missing} keys in climatology.')
    mean = _time_gather(climatology, truth, mean_names)
    std = _time_gather(climatology, truth, std_names)
    z = stats.norm.ppf(self.quantile)  # np.float64: promotes float32 data
    out = xl.Dataset(coords=mean.coords)
    for v in variables:
      out.data_vars[v] = xl.DataArray(mean[v].values + z * std[v].values,
                                      mean[v].dims, mean.coords, v)
    return xl.like_input(out, given)


def get_threshold_cls(threshold_method: str):
  """thresholds.py:190-197."""
  if threshold_method == 'quantile':
    return QuantileThreshold
  if threshold_method == 'gaussian_quantile':
    return GaussianQuantileThreshold
  raise NotImplementedError(f'Unknown threshold method: {threshold_method}')
