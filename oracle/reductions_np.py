"""NumPy restatement of the reference's averaging pipelines (test/measurement
infrastructure only -- see oracle/__init__.py):

  scripts/compute_ensemble_mean.py:111-141   xbeam.Mean(realization, skipna)
  scripts/compute_averages.py:125-167        v * lat_weights.reindex_like(v),
                                             then xbeam.Mean(averaging_dims)
  scripts/compute_statistical_moments.py:52-80  zeroth / first / second raw
                                             moments over (latitude, longitude)

xbeam.Mean combines per-chunk (sum, count) pairs, i.e. it IS `mean(dims, skipna)`
of the whole array; restated with np.mean / np.nanmean on named arrays.
"""
import warnings

import numpy as np

from oracle import metrics_np
from oracle.named import DS, NA


def _mean(a: NA, dims, skipna: bool) -> NA:
  axes = tuple(a.dims.index(d) for d in dims if d in a.dims)
  if not axes:
    return a
  keep = tuple(d for d in a.dims if d not in dims)
  with warnings.catch_warnings(), np.errstate(all='ignore'):
    warnings.simplefilter('ignore')
    fn = np.nanmean if skipna else np.mean
    return NA(fn(a.data, axis=axes), keep)


def _drop(ds: DS, dims) -> dict:
  return {k: c for k, c in ds.coords.items()
          if k not in dims and not (isinstance(c, NA)
                                    and any(d in dims for d in c.dims))}


def ensemble_mean(ds: DS, realization_name='realization',
                  skipna=False) -> DS:
  """compute_ensemble_mean.py:134."""
  return DS({k: _mean(v, (realization_name,), skipna) for k, v in ds.items()},
            _drop(ds, (realization_name,)))


def averages(ds: DS, averaging_dims, skipna=False) -> DS:
  """compute_averages.py:139-160: latitude is weighted by multiplying with the
  (mean-one) latitude weights BEFORE the plain mean."""
  averaging_dims = tuple(averaging_dims)
  out = {}
  for k, v in ds.items():
    if 'latitude' in averaging_dims and 'latitude' in v.dims:
      w = metrics_np.get_lat_weights(ds.coord('latitude'))
      v = v * w
    out[k] = _mean(v, averaging_dims, skipna)
  return DS(out, _drop(ds, averaging_dims))


def statistical_moments(ds: DS, reduce_dims=('latitude', 'longitude')) -> DS:
  """compute_statistical_moments.py:52-80 for the three orders; `.mean()`
  without arguments skips NaNs (xarray's default for floats)."""
  out = {}
  for k, v in ds.items():
    out[f'{k}_zeroth'] = _mean(NA(~np.isnan(v.data), v.dims), reduce_dims,
                               False)
    out[f'{k}_first'] = _mean(v, reduce_dims, True)
    out[f'{k}_second'] = _mean(NA(np.square(v.data), v.dims), reduce_dims,
                               True)
  return DS(out, _drop(ds, reduce_dims))
