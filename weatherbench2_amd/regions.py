"""Region selectors: same classes/fields as weatherbench2/regions.py.

The reference applies a region by slicing the chunk (`SliceRegion`, a fancy-index
COPY of the whole chunk per region, regions.py:72-95) or by multiplying the
weights with a mask (`ExtraTropicalRegion`/`LandRegion`, :98-138), once per
metric x region (evaluation.py:416-430).  Here a region is only *described*:
`decompose_region` turns it into

  lat_mult[n_lat], lon_mult[n_lon]   integer multiplicities (0 = excluded; a
                                     list of overlapping slices repeats rows,
                                     exactly like the reference's concat), and
  field[n_lat, n_lon] or None        a 2-D weight factor (land-sea mask),

and the fused kernel evaluates every region in one pass (see plan.py).  The
index/mask work is bit-exact: label-inclusive slices on the coordinate values
(pandas `slice_indexer` semantics) and `abs(lat) >= 20` on the stored labels.

Objects of the REFERENCE's own classes are accepted too (duck-typed on class
name + dataclass fields), so an existing `config.Eval.regions` dict works.
"""
from __future__ import annotations

import dataclasses
import typing as t

import numpy as np


def _xarray_of(dataset, weights):
  """`Region.apply` exists for FOREIGN metrics -- objects written against the
  reference's protocol, which call `region.apply(dataset, weights)` on xarray
  objects inside their own `_spatial_average` (metrics.py:157).  The GPU metrics
  never call it: they evaluate every region inside the fused pass
  (`decompose_region`)."""
  from weatherbench2_amd import xarray_lite as xl
  if not (xl.is_xarray(dataset) and xl.is_xarray(weights)):
    raise NotImplementedError(
        'Region.apply serves xarray-based (foreign) metrics; GPU metrics '
        'evaluate regions in the fused kernel, see decompose_region')
  return xl._xr


@dataclasses.dataclass
class Region:
  """regions.py:24-54."""

  def apply(self, dataset, weights):
    """(dataset, weights) restricted to the region, like the reference's
    `Region.apply` (regions.py:40-54)."""
    raise NotImplementedError


@dataclasses.dataclass
class SliceRegion(Region):
  """Latitude-longitude box selection (regions.py:57-95)."""

  lat_slice: t.Optional[t.Union[slice, list]] = dataclasses.field(
      default_factory=lambda: slice(None, None))
  lon_slice: t.Optional[t.Union[slice, list]] = dataclasses.field(
      default_factory=lambda: slice(None, None))

  def apply(self, dataset, weights):
    """Rows / columns of every slice, in the order the slices are listed
    (regions.py:72-95 concatenates the selected labels the same way)."""
    _xarray_of(dataset, weights)
    picked = {}
    for dim, spec in (('latitude', self.lat_slice),
                      ('longitude', self.lon_slice)):
      labels = np.asarray(dataset[dim].values)
      parts = spec if isinstance(spec, list) else [spec]
      picked[dim] = np.concatenate(
          [_slice_positions(labels, s) for s in parts]).astype(np.int64)
    return (dataset.isel(picked),
            weights.isel({d: i for d, i in picked.items()
                          if d in weights.dims}))


@dataclasses.dataclass
class ExtraTropicalRegion(Region):
  """|lat| >= 20; `threshold_lat` is ignored like in regions.py:102-109."""

  threshold_lat: t.Optional[float] = 20

  def apply(self, dataset, weights):
    xr = _xarray_of(dataset, weights)
    lat = np.asarray(dataset['latitude'].values)
    outside = xr.DataArray((np.abs(lat) >= 20).astype(float),
                           dims=('latitude',), coords={'latitude': lat})
    return dataset, weights * outside


@dataclasses.dataclass
class LandRegion(Region):
  """Weights x land-sea mask, optionally thresholded (regions.py:112-138).

  `land_sea_mask`: a DataArray (ours or xarray's) over latitude/longitude with
  those coordinates attached.
  """

  land_sea_mask: t.Any = None
  threshold: t.Optional[float] = None

  def apply(self, dataset, weights):
    """Weights x mask; the mask's labels are cast to the dataset's coordinate
    dtype first and xarray aligns the product by label (regions.py:125-138)."""
    xr = _xarray_of(dataset, weights)
    lsm = self.land_sea_mask
    values = np.asarray(lsm.values)
    labels = {}
    for name in ('latitude', 'longitude'):
      c = lsm.coords[name]
      labels[name] = np.asarray(getattr(c, 'values', c)).astype(
          np.asarray(dataset[name].values).dtype)
    if self.threshold is not None:
      values = (values > self.threshold).astype(float)
    land = xr.DataArray(values, dims=tuple(lsm.dims),
                        coords={d: labels[d] for d in lsm.dims})
    return dataset, weights * land


@dataclasses.dataclass
class CombinedRegion(Region):
  """Sequential application (regions.py:141-158)."""

  regions: list = dataclasses.field(default_factory=list)

  def apply(self, dataset, weights):
    for region in self.regions:
      dataset, weights = region.apply(dataset, weights)
    return dataset, weights


@dataclasses.dataclass
class RegionSpec:
  lat_mult: np.ndarray           # int64[n_lat]
  lon_mult: np.ndarray           # int64[n_lon]
  field: t.Optional[np.ndarray]  # float64[n_lat, n_lon] or None


def _slice_positions(labels: np.ndarray, s: slice) -> np.ndarray:
  """Positions of `labels.sel(slice(a, b))`: both ends inclusive."""
  if s.step not in (None, 1):
    raise NotImplementedError('stepped label slices are not supported')
  if len(labels) > 1 and not (np.diff(labels) >= 0).all():
    raise NotImplementedError(
        'label slice on a non-monotonic coordinate (the reference raises a '
        'KeyError here)')
  lo = 0 if s.start is None else int(np.searchsorted(labels, s.start, 'left'))
  hi = len(labels) if s.stop is None else int(
      np.searchsorted(labels, s.stop, 'right'))
  return np.arange(lo, max(lo, hi))


def _land_field(region, lat: np.ndarray, lon: np.ndarray) -> np.ndarray:
  lsm = region.land_sea_mask
  dims = tuple(lsm.dims)
  if set(dims) != {'latitude', 'longitude'}:
    raise ValueError(f'land_sea_mask must be (latitude, longitude), got {dims}')
  values = np.asarray(lsm.values)
  if dims == ('longitude', 'latitude'):
    values = values.T

  def coord(name):
    c = lsm.coords[name]
    return np.asarray(getattr(c, 'values', c))

  # regions.py:131-134: mask labels are cast to the dataset's coord dtype and
  # then aligned by label (inner join).
  out = values
  for axis, (name, want) in enumerate((('latitude', lat), ('longitude', lon))):
    have = coord(name).astype(want.dtype)
    if have.shape == want.shape and np.array_equal(have, want):
      continue
    pos = {v: i for i, v in enumerate(have.tolist())}
    try:
      idx = np.array([pos[v] for v in want.tolist()], dtype=np.int64)
    except KeyError as e:
      raise NotImplementedError(
          f'land_sea_mask has no {name} label {e}: an inner join that drops '
          'grid points is not supported') from e
    out = np.take(out, idx, axis=axis)
  out = np.asarray(out, dtype=np.float64)
  if region.threshold is not None:
    out = (np.asarray(out) > region.threshold).astype(np.float64)
  if not np.isfinite(out).all():
    raise ValueError('land_sea_mask weights must be finite')
  return np.ascontiguousarray(out)


def decompose_region(region, lat: np.ndarray, lon: np.ndarray) -> RegionSpec:
  """Region -> (row multiplicities, column multiplicities, 2-D factor)."""
  n_lat, n_lon = len(lat), len(lon)
  # Current selection as positions into the ORIGINAL coordinates (repeats kept).
  lat_sel = np.arange(n_lat)
  lon_sel = np.arange(n_lon)
  lat_mask = np.ones(n_lat, dtype=np.int64)  # 0/1 factors from weight masks
  field = None

  def visit(r):
    nonlocal lat_sel, lon_sel, lat_mask, field
    kind = type(r).__name__
    if r is None:
      return
    if kind == 'SliceRegion':
      lats = r.lat_slice if isinstance(r.lat_slice, list) else [r.lat_slice]
      lons = r.lon_slice if isinstance(r.lon_slice, list) else [r.lon_slice]
      cur_lat, cur_lon = lat[lat_sel], lon[lon_sel]
      lat_sel = lat_sel[np.concatenate(
          [_slice_positions(cur_lat, s) for s in lats]).astype(np.int64)]
      lon_sel = lon_sel[np.concatenate(
          [_slice_positions(cur_lon, s) for s in lons]).astype(np.int64)]
    elif kind == 'ExtraTropicalRegion':
      lat_mask = lat_mask * (np.abs(lat) >= 20).astype(np.int64)
    elif kind == 'LandRegion':
      f = _land_field(r, lat, lon)
      field = f if field is None else field * f
    elif kind == 'CombinedRegion':
      for sub in r.regions:
        visit(sub)
    else:
      raise NotImplementedError(
          f'region type {kind} cannot be decomposed for the fused kernel')

  visit(region)
  lat_mult = np.bincount(lat_sel, minlength=n_lat).astype(np.int64) * lat_mask
  lon_mult = np.bincount(lon_sel, minlength=n_lon).astype(np.int64)
  return RegionSpec(lat_mult, lon_mult, field)
