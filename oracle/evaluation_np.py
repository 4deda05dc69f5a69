"""NumPy restatement of the baseline substitutions of the reference's
evaluation drivers (TEST INFRASTRUCTURE, see oracle/__init__.py).

  reference (/root/reference/weatherbench2/)           here
  evaluation.py:452-460   forecast := climatology      climatology_forecast
  evaluation.py:461-471 + utils.py:47-70               make_probabilistic_climatology
                          probabilistic climatology    + climatology_forecast
  evaluation.py:165-193   persistence, by-valid        create_persistence_forecast
  evaluation.py:651-675   persistence, by-init chunk   persistence_like_forecast_chunk
  evaluation.py:618-649   climatology chunk            climatology_like_forecast_chunk
  evaluation.py:474-475   truth.sel(time=valid_time)   truth_at_valid_time
  evaluation.py:388-438   metric x region loop         metric_and_region_loop
  evaluation.py:441-483   _evaluate_all_metrics        evaluate_all_metrics

Everything is EAGER here (np.take copies), unlike the product's slab tables.
xarray's vectorised `.sel` puts the indexer's dims where the first indexed dim
was, walking the variable's dims in order (Variable._broadcast_indexes_
vectorized) -- `_vectorised_take` restates that rule.  Pinned by the outputs of
the reference's own `_evaluate_all_metrics` on the stand-in xarray
(tests/golden/make_reference_vectors.py, `evalall_*` vectors).
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from oracle.named import DS, NA


def _labels(ds: DS, name: str) -> np.ndarray:
  c = ds.coords[name]
  return np.asarray(c.data if isinstance(c, NA) else c)


def _lookup(have, want, what):
  index = pd.Index(np.asarray(have))
  pos = index.get_indexer(np.asarray(want).ravel())
  if (pos < 0).any():
    raise KeyError(f'not all values found in index {what!r}')
  return pos.reshape(np.shape(want))


def _vectorised_take(var: NA, positions: dict, new_dims: tuple, fill=False):
  """var.isel({dim: positions[dim]}) with every indexer over `new_dims`;
  positions < 0 (only with `fill`) give NaN."""
  out_dims = []
  for d in var.dims:
    for nd in (new_dims if d in positions else (d,)):
      if nd not in out_dims:
        out_dims.append(nd)
  key, hole = [], None
  shape_new = np.shape(next(iter(positions.values())))
  for d in var.dims:
    if d in positions:
      p = np.asarray(positions[d])
      if fill:
        hole = (p < 0) if hole is None else (hole | (p < 0))
        p = np.maximum(p, 0)
      key.append(p)
    else:
      key.append(slice(None))
  data = var.data[tuple(key)]
  # numpy: adjacent advanced indices stay in place, separated ones go first
  adv = [i for i, d in enumerate(var.dims) if d in positions]
  adjacent = adv == list(range(adv[0], adv[0] + len(adv)))
  rest = [d for d in var.dims if d not in positions]
  if adjacent:
    np_dims = list(var.dims[:adv[0]]) + list(new_dims) + [
        d for d in var.dims[adv[-1] + 1:]]
  else:
    np_dims = list(new_dims) + rest
  data = np.transpose(data, [np_dims.index(d) for d in out_dims])
  if fill and hole is not None and hole.any():
    if data.dtype.kind != 'f':
      data = data.astype(np.float64)
    else:
      data = data.copy()
    mask = np.zeros([1] * 0, dtype=bool)
    shape = [1] * len(out_dims)
    for nd, n in zip(new_dims, shape_new):
      shape[out_dims.index(nd)] = n
    order = [nd for nd in out_dims if nd in new_dims]
    mask = np.transpose(hole, [new_dims.index(nd) for nd in order]
                        ).reshape(shape)
    data[np.broadcast_to(mask, data.shape)] = np.nan
  return NA(data, out_dims)


def _time_indexer(forecast: DS, time_dim: str):
  c = forecast.coords[time_dim]
  if isinstance(c, NA):
    values, dims = np.asarray(c.data), tuple(c.dims)
  else:
    values, dims = np.asarray(c), (time_dim,)
  carried = {}
  for k, v in forecast.coords.items():
    vdims = tuple(v.dims) if isinstance(v, NA) else (k,)
    if all(d in dims for d in vdims) and (isinstance(v, NA) or k in dims):
      carried[k] = v
  return values, dims, carried


def climatology_forecast(forecast: DS, climatology: DS, time_dim: str,
                         variables=None, hour_if_present=False) -> DS:
  """evaluation.py:452-460 (and :629-646 with `hour_if_present`)."""
  values, dims, carried = _time_indexer(forecast, time_dim)
  idx = pd.DatetimeIndex(values.ravel())
  doy = np.asarray(idx.dayofyear).reshape(values.shape)
  hour = np.asarray(idx.hour).reshape(values.shape)
  positions = {'dayofyear': _lookup(_labels(climatology, 'dayofyear'), doy,
                                    'dayofyear')}
  coords = {k: v for k, v in climatology.coords.items()
            if k not in ('dayofyear', 'hour')}
  coords.update(carried)
  coords['dayofyear'] = NA(doy, dims)
  if not hour_if_present or 'hour' in climatology.coords:
    positions['hour'] = _lookup(_labels(climatology, 'hour'), hour, 'hour')
    coords['hour'] = NA(hour, dims)
  variables = list(forecast.keys()) if variables is None else list(variables)
  names = {v: v for v in variables}
  if hour_if_present and not all(v in climatology for v in variables):
    names = {v: f'{v}_mean' for v in variables}  # evaluation.py:635-639
  out = {}
  for v, cname in names.items():
    var = climatology[cname]  # KeyError like the reference
    pos = {d: p for d, p in positions.items() if d in var.dims}
    out[v] = _vectorised_take(var, pos, dims, fill=True)
  return DS(out, coords)


def make_probabilistic_climatology(ds: DS, start_year, end_year, hour_interval,
                                   variables=None) -> DS:
  """utils.py:47-70: for every hour, the years stacked along `number`, indexed
  by dayofyear (concat's outer join: sorted union, NaN where a year lacks the
  day), then the hours stacked along `hour`."""
  hours = np.arange(0, 24, hour_interval)
  years = np.arange(start_year, end_year + 1)
  times = pd.DatetimeIndex(_labels(ds, 'time'))
  names = list(ds.keys()) if variables is None else list(variables)
  per_hour, doys_all = [], set()
  for hour in hours:
    per_year = []
    for year in years:
      pick = np.nonzero((times.hour == hour) & (times.year == year))[0]
      if not (times.year == year).any():
        raise KeyError(str(year))
      per_year.append((pick, np.asarray(times.dayofyear)[pick]))
      doys_all |= set(per_year[-1][1].tolist())
    per_hour.append(per_year)
  doys = np.array(sorted(doys_all), dtype=np.int64)
  out = {}
  for name in names:
    var = ds[name]
    ax = var.dims.index('time')
    rest_shape = var.data.shape[:ax] + var.data.shape[ax + 1:]
    dtype = var.data.dtype if var.data.dtype.kind == 'f' else np.float64
    stack = np.full((len(hours), len(years), len(doys)) + rest_shape, np.nan,
                    dtype=dtype)
    moved = np.moveaxis(var.data, ax, 0)
    for hi, per_year in enumerate(per_hour):
      for yi, (pick, d) in enumerate(per_year):
        stack[hi, yi, np.searchsorted(doys, d)] = moved[pick]
    rest_dims = tuple(x for x in var.dims if x != 'time')
    # swap_dims puts dayofyear where time was; concat adds number, then hour,
    # as new leading dims
    data = np.moveaxis(stack, 2, 2 + ax)
    dims = ('hour', 'number') + rest_dims[:ax] + ('dayofyear',) + rest_dims[ax:]
    out[name] = NA(data, dims)
  coords = {k: v for k, v in ds.coords.items()
            if k != 'time' and not (isinstance(v, NA) and 'time' in v.dims)}
  coords.update(hour=hours, number=np.arange(len(years)), dayofyear=doys)
  return DS(out, coords)


def create_persistence_forecast(forecast: DS, obs: DS) -> DS:
  """evaluation.py:165-193 (by-valid: init_time is a (time, lead_time) coord)."""
  init = forecast.coords['init_time']
  if not isinstance(init, NA) or 'time' not in init.dims:
    raise AttributeError("init_time has no 'time' dim")
  lead_dims = tuple(d for d in init.dims if d != 'time')
  init = init.transpose('time', *lead_dims)
  time = _labels(forecast, 'time')
  lead_max = max(np.max(_labels(forecast, d)) for d in lead_dims)
  keep = time >= time[0] + lead_max
  init_values = init.data[keep]
  pos = _lookup(_labels(obs, 'time'), init_values, 'time')
  coords = {k: v for k, v in obs.coords.items()
            if k != 'time' and not (isinstance(v, NA) and 'time' in v.dims)}
  coords['time'] = time[keep]
  for d in lead_dims:
    coords[d] = _labels(forecast, d)
  coords['init_time'] = NA(init_values, init.dims)
  out = {}
  for name, var in obs.items():
    out[name] = (_vectorised_take(var, {'time': pos}, tuple(init.dims))
                 if 'time' in var.dims else var)
  return DS(out, coords)


def persistence_like_forecast_chunk(forecast_chunk: DS, truth: DS,
                                    variables=None, lead_dim='lead_time') -> DS:
  """evaluation.py:651-675 (by-init)."""
  init = _labels(forecast_chunk, 'init_time')
  lead = _labels(forecast_chunk, lead_dim)
  pos = _lookup(_labels(truth, 'time'), init, 'time')
  coords = {k: v for k, v in truth.coords.items()
            if k != 'time' and not (isinstance(v, NA) and 'time' in v.dims)}
  coords.update({lead_dim: lead, 'init_time': init})
  if 'valid_time' in forecast_chunk.coords:
    coords['valid_time'] = forecast_chunk.coords['valid_time']
  out = {}
  for name in (variables or truth.keys()):
    var = truth[name]
    picked = _vectorised_take(var, {'time': pos}, ('init_time',))
    out[name] = picked.expand_dims(lead_dim, len(lead))  # new dim first
  return DS(out, coords)


def climatology_like_forecast_chunk(forecast_chunk: DS, climatology: DS,
                                    variables, by_init=True) -> DS:
  """evaluation.py:618-649."""
  return climatology_forecast(forecast_chunk, climatology,
                              'valid_time' if by_init else 'time', variables,
                              hour_if_present=True)


def truth_at_valid_time(truth: DS, forecast: DS) -> DS:
  """evaluation.py:474-475: truth.sel(time=forecast.valid_time)."""
  vt = forecast.coords['valid_time']
  pos = _lookup(_labels(truth, 'time'), vt.data, 'time')
  coords = {k: v for k, v in truth.coords.items()
            if k != 'time' and not (isinstance(v, NA) and 'time' in v.dims)}
  for d in vt.dims:
    coords[d] = forecast.coords[d]
  coords['valid_time'] = vt
  coords['time'] = NA(vt.data, vt.dims)
  out = {}
  for name, var in truth.items():
    out[name] = (_vectorised_take(var, {'time': pos}, tuple(vt.dims))
                 if 'time' in var.dims else var)
  return DS(out, coords)


def metric_and_region_loop(forecast: DS, truth: DS, metrics: dict, regions,
                           skipna: bool, temporal_mean: bool = True,
                           compute_chunk: bool = False) -> dict:
  """evaluation.py:388-438 without the final concat / merge: {(metric name,
  region name or None): DS}."""
  results = {}
  for name, metric in metrics.items():
    fn = (metric.compute_chunk if compute_chunk or not temporal_mean
          else metric.compute)
    if regions is not None:
      for region_name, region in regions.items():
        results[(name, region_name)] = fn(forecast, truth, region=region,
                                          skipna=skipna)
    else:
      results[(name, None)] = fn(forecast, truth, skipna=skipna)
  return results


def evaluate_all_metrics(forecast: DS, truth: DS, climatology, metrics: dict,
                         regions, skipna: bool, by_init: bool,
                         evaluate_climatology=False, evaluate_persistence=False,
                         evaluate_probabilistic_climatology=False,
                         start_year=None, end_year=None, hour_interval=None,
                         temporal_mean=True) -> dict:
  """evaluation.py:441-483 between opening the data and writing the result."""
  time_dim = 'valid_time' if by_init else 'time'
  if evaluate_climatology:
    forecast = climatology_forecast(forecast, climatology, time_dim)
  if evaluate_probabilistic_climatology:
    prob = make_probabilistic_climatology(truth, start_year, end_year,
                                          hour_interval,
                                          variables=list(forecast.keys()))
    forecast = climatology_forecast(forecast, prob, time_dim)
  if evaluate_persistence:
    forecast = create_persistence_forecast(forecast, truth)
  if by_init:
    truth = truth_at_valid_time(truth, forecast)
  return metric_and_region_loop(forecast, truth, metrics, regions, skipna,
                                temporal_mean)
