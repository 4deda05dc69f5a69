"""NumPy restatement of weatherbench2/thresholds.py (TEST INFRASTRUCTURE).

  _get_climatology_mean/std/quantile   thresholds.py:25-87
  QuantileThreshold                    thresholds.py:116-148
  GaussianQuantileThreshold            thresholds.py:151-187
Quirk kept: dayofyear always comes from truth['time'], the hour from
truth[time_dim] (thresholds.py:140, 176).
"""
from __future__ import annotations

import dataclasses

import numpy as np
from scipy import stats

from oracle.named import DS, NA


def _time_gather(climatology: DS, truth: DS, variables: dict) -> DS:
  """climatology.sel(level=..., dayofyear=..., [hour=...]) for the truth's times.

  `variables`: {climatology variable name: output name}.
  """
  import pandas as pd
  times = pd.DatetimeIndex(np.asarray(truth.coords['time']).ravel())
  doy_pos = {v: i for i, v in enumerate(
      climatology.coord('dayofyear').tolist())}
  doy_idx = np.array([doy_pos[v] for v in np.asarray(times.dayofyear).tolist()])
  has_hour = 'hour' in climatology.coords
  if has_hour:
    hour_pos = {v: i for i, v in enumerate(climatology.coord('hour').tolist())}
    hour_idx = np.array([hour_pos[v] for v in np.asarray(times.hour).tolist()])
  level_idx = None
  if 'level' in truth.dims and 'level' in climatology.coords:
    pos = {v: i for i, v in enumerate(climatology.coord('level').tolist())}
    level_idx = np.array([pos[v] for v in truth.coord('level').tolist()])
  out = {}
  for src, dst in variables.items():
    v = climatology[src]
    if level_idx is not None and 'level' in v.dims:
      v = v.isel(level=level_idx)
    rest = tuple(d for d in v.dims if d not in ('dayofyear', 'hour'))
    if has_hour and 'hour' in v.dims:
      data = v.transpose('dayofyear', 'hour', *rest).data[doy_idx, hour_idx]
    else:
      data = v.transpose('dayofyear', *rest).data[doy_idx]
    out[dst] = NA(data, ('time',) + rest)
  coords = {k: c for k, c in truth.coords.items()}
  return DS(out, coords)


@dataclasses.dataclass
class Threshold:
  climatology: DS
  quantile: float

  def compute(self, truth: DS) -> DS:
    raise NotImplementedError


@dataclasses.dataclass
class QuantileThreshold(Threshold):

  def compute(self, truth: DS) -> DS:
    variables = [str(k) for k in truth.keys()]
    names = {k + '_quantile': k for k in variables}
    missing = set(names).difference(self.climatology.keys())
    if missing:
      raise KeyError(f'Did not find {missing} keys in climatology.')
    # .sel(quantile=q, tolerance=0.01, method='nearest')  (thresholds.py:76-78)
    q = np.asarray(self.climatology.coord('quantile'), dtype=float)
    i = int(np.argmin(np.abs(q - self.quantile)))
    if abs(q[i] - self.quantile) > 0.01:
      raise KeyError(f'Did not find quantiles {self.quantile}+-0.01 in '
                     'climatology.')
    clim = self.climatology.isel(quantile=i)
    return _time_gather(clim, truth, names)


@dataclasses.dataclass
class GaussianQuantileThreshold(Threshold):

  def compute(self, truth: DS) -> DS:
    variables = [str(k) for k in truth.keys()]
    if all(v in self.climatology for v in variables):
      mean_names = {v: v for v in variables}
    else:
      mean_names = {v + '_mean': v for v in variables}
      missing = set(mean_names).difference(self.climatology.keys())
      if missing:
        raise KeyError(f'Did not find {missing} keys in climatology.')
    std_names = {v + '_std': v for v in variables}
    missing = set(std_names).difference(self.climatology.keys())
    if missing:
      raise KeyError(f'Did not find {missing} keys in climatology.')
    mean = _time_gather(self.climatology, truth, mean_names)
    std = _time_gather(self.climatology, truth, std_names)
    # np.float64 scalar * array: float64 under NumPy >= 2 promotion rules
    z = stats.norm.ppf(self.quantile)
    return DS({k: NA(mean[k].data + z * std[k].data, mean[k].dims)
               for k in variables}, mean.coords)
