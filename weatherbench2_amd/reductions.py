"""Averaging pipelines on the MI355X: the compute cores of

  scripts/compute_ensemble_mean.py:111-141    ensemble_mean
  scripts/compute_averages.py:125-167         averages
  scripts/compute_statistical_moments.py:52-80  statistical_moments

(reference = /root/reference).  All three are `sum_r w_r f(x) / count` along a
merged axis: one read of the data by wb2_axis_moments (fp64 sums, deterministic
slice order), results cast back to what xarray would return (the input dtype
for plain means, float64 where the float64 latitude weights take part).
Inputs may be numpy- or device-backed Datasets (xarray Datasets are converted
at the boundary); results stay on the device until `.values` is asked for.
Dataset IO, Beam and the CLI flags around these cores are out of scope.
"""
from __future__ import annotations

import functools
import typing as t

import numpy as np
import torch

from weatherbench2_amd import engine
from weatherbench2_amd import plan as plan_lib
from weatherbench2_amd import xarray_lite as xl


def _like_input(fn):
  """xarray in -> xarray out (xarray_lite.like_input)."""
  @functools.wraps(fn)
  def wrapper(dataset, *args, **kwargs):
    return xl.like_input(fn(dataset, *args, **kwargs), dataset)
  return wrapper


def _moments(da: xl.DataArray, dims: t.Sequence[str], skipna: bool,
             weights: t.Optional[dict] = None, want_sq: bool = False):
  """(sum, sumsq|None, count, kept dims) over `dims` of one variable."""
  red = [d for d in da.dims if d in dims]
  keep = tuple(d for d in da.dims if d not in dims)
  device = engine.require_gpu()
  x = engine.as_device_tensor(da.data, device)
  if x.dtype not in (torch.float32, torch.float64):
    x = x.to(torch.float64)
  axes = [da.dims.index(d) for d in red]
  # [lead][red...][tail]: the reduced dims must be adjacent; if they are not,
  # move them behind the leading kept dims (one transposed copy).
  lo, hi = min(axes), max(axes)
  if axes != list(range(lo, hi + 1)):
    order = [i for i in range(x.ndim) if i not in axes]
    order = order[:lo] + axes + order[lo:]
    x = x.permute(*order)
    keep = tuple(da.dims[i] for i in order if i not in axes)
    dims_now = [da.dims[i] for i in order]
    axes = [dims_now.index(d) for d in red]
    lo, hi = min(axes), max(axes)
  x = x.contiguous()
  shape = tuple(x.shape)
  n_lead = int(np.prod(shape[:lo], dtype=np.int64))
  n_red = int(np.prod(shape[lo:hi + 1], dtype=np.int64))
  n_tail = int(np.prod(shape[hi + 1:], dtype=np.int64))
  w_red, w_repeat = None, 1
  if weights:
    # weights vary along (some of) the merged reduced dims: expand them over
    # the dims up to the last weighted one; the dims behind it share a weight
    last = max(j for j, d in enumerate(red) if d in weights)
    w = np.ones(shape[lo:lo + last + 1], dtype=np.float64)
    for j, d in enumerate(red[:last + 1]):
      if d in weights:
        sh = [1] * (last + 1)
        sh[j] = shape[lo + j]
        w = w * np.asarray(weights[d], dtype=np.float64).reshape(sh)
    w_red = torch.from_numpy(np.ascontiguousarray(w).ravel()).to(device)
    w_repeat = int(np.prod(shape[lo + last + 1:hi + 1], dtype=np.int64))
  out_shape = shape[:lo] + shape[hi + 1:]
  total, sq, count = engine.axis_moments(x, n_lead, n_red, n_tail, w_red,
                                         skipna, want_sq, w_repeat)
  rs = lambda v: None if v is None else v.reshape(out_shape)
  return rs(total), rs(sq), rs(count), keep, x.dtype


def _coords_without(ds: xl.Dataset, dims) -> dict:
  out = {}
  for k, c in ds.coords.items():
    if k in dims:
      continue
    if isinstance(c, xl.DataArray) and any(d in dims for d in c.dims):
      continue
    out[k] = c
  return out


@_like_input
def mean(dataset, dims: t.Sequence[str], skipna: bool = False) -> xl.Dataset:
  """`dataset.mean(dims, skipna)`: variables lacking a dim are averaged over
  the ones they have; results keep the input dtype, like xarray."""
  ds = xl.as_dataset(dataset)
  dims = (dims,) if isinstance(dims, str) else tuple(dims)
  out = xl.Dataset(coords=_coords_without(ds, dims), attrs=dict(ds.attrs))
  for name, da in ds.data_vars.items():
    if not any(d in da.dims for d in dims):
      out.data_vars[name] = da
      continue
    total, _, count, keep, dtype = _moments(da, dims, skipna)
    out.data_vars[name] = xl.DataArray((total / count).to(dtype), keep,
                                       out.coords, name)
  return out


@_like_input
def ensemble_mean(dataset, realization_name: str = 'realization',
                  skipna: bool = False) -> xl.Dataset:
  """The ensemble mean of every variable (compute_ensemble_mean.py:134)."""
  ds = xl.as_dataset(dataset)
  if realization_name not in ds.dims:
    raise ValueError(f'{realization_name!r} not found in {tuple(ds.dims)}')
  return mean(ds, (realization_name,), skipna)


@_like_input
def averages(dataset, averaging_dims: t.Sequence[str],
             skipna: bool = False) -> xl.Dataset:
  """Averages over `averaging_dims` (compute_averages.py:139-160).  Latitude is
  area-weighted the way the script does it: the data are multiplied by the
  mean-one latitude weights (metrics.py:55-60) before a plain mean, which makes
  the result float64 for the variables that have a latitude dim."""
  ds = xl.as_dataset(dataset)
  averaging_dims = tuple(averaging_dims)
  weights = None
  if 'latitude' in averaging_dims:
    weights = {'latitude': plan_lib.get_lat_weights(
        np.asarray(ds.coords['latitude']))}
  out = xl.Dataset(coords=_coords_without(ds, averaging_dims),
                   attrs=dict(ds.attrs))
  for name, da in ds.data_vars.items():
    if not any(d in da.dims for d in averaging_dims):
      out.data_vars[name] = da
      continue
    weighted = weights is not None and 'latitude' in da.dims
    total, _, count, keep, dtype = _moments(
        da, averaging_dims, skipna, weights if weighted else None)
    if weighted:
      # v * w is float64 (NumPy promotes float32 * float64)
      dtype = torch.promote_types(dtype, torch.float64)
    out.data_vars[name] = xl.DataArray((total / count).to(dtype), keep,
                                       out.coords, name)
  return out


@_like_input
def statistical_moments(dataset,
                        reduce_dims=('latitude', 'longitude')) -> xl.Dataset:
  """`<var>_zeroth` (fraction of non-NaN points), `<var>_first` (mean) and
  `<var>_second` (mean square) over `reduce_dims`, NaNs skipped
  (compute_statistical_moments.py:52-80) -- from ONE read of each variable."""
  ds = xl.as_dataset(dataset)
  reduce_dims = tuple(reduce_dims)
  out = xl.Dataset(coords=_coords_without(ds, reduce_dims),
                   attrs=dict(ds.attrs))
  for name, da in ds.data_vars.items():
    total, sq, count, keep, dtype = _moments(da, reduce_dims, True,
                                             want_sq=True)
    n_red = 1
    for d in da.dims:
      if d in reduce_dims:
        n_red *= da.sizes[d]
    out.data_vars[f'{name}_zeroth'] = xl.DataArray(
        count / float(n_red), keep, out.coords, f'{name}_zeroth')
    out.data_vars[f'{name}_first'] = xl.DataArray(
        (total / count).to(dtype), keep, out.coords, f'{name}_first')
    out.data_vars[f'{name}_second'] = xl.DataArray(
        (sq / count).to(dtype), keep, out.coords, f'{name}_second')
  return out
