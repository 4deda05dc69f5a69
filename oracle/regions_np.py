"""NumPy restatement of weatherbench2/regions.py (TEST INFRASTRUCTURE).

Follows /root/reference/weatherbench2/regions.py line by line:
  Region.apply            regions.py:40-54
  SliceRegion             regions.py:57-95   (label-inclusive slices, lists are
                                               concatenated in list order)
  ExtraTropicalRegion     regions.py:98-109  (|lat| >= 20 hard-coded, :108)
  LandRegion              regions.py:112-138
  CombinedRegion          regions.py:141-158

Weights are an ``NA`` with dims ('latitude',) or ('latitude', 'longitude').
"""
from __future__ import annotations

import dataclasses
import typing as t

import numpy as np

from oracle.named import DS, NA


def _label_slice_indices(coord: np.ndarray, s: slice) -> np.ndarray:
  """Positions selected by ``coord.sel(slice(a, b))`` on an increasing index.

  xarray -> pandas ``Index.slice_indexer``: both ends inclusive,
  ``searchsorted(a, 'left') : searchsorted(b, 'right')``.
  """
  lo = 0 if s.start is None else int(np.searchsorted(coord, s.start, 'left'))
  hi = len(coord) if s.stop is None else int(
      np.searchsorted(coord, s.stop, 'right'))
  return np.arange(lo, max(hi, lo))


@dataclasses.dataclass
class Region:

  def apply(self, dataset: DS, weights: NA) -> tuple[DS, NA]:
    raise NotImplementedError


@dataclasses.dataclass
class SliceRegion(Region):
  """regions.py:57-95."""

  lat_slice: t.Union[slice, list] = dataclasses.field(
      default_factory=lambda: slice(None, None))
  lon_slice: t.Union[slice, list] = dataclasses.field(
      default_factory=lambda: slice(None, None))

  def apply(self, dataset, weights):
    lats = self.lat_slice if isinstance(self.lat_slice, list) else [
        self.lat_slice]
    lons = self.lon_slice if isinstance(self.lon_slice, list) else [
        self.lon_slice]
    # regions.py:79-84: concat of label selections, in list order.
    lat_idx = np.concatenate(
        [_label_slice_indices(dataset.coord('latitude'), s) for s in lats])
    lon_idx = np.concatenate(
        [_label_slice_indices(dataset.coord('longitude'), s) for s in lons])
    w_index = {}
    if 'latitude' in weights.dims:
      w_index['latitude'] = lat_idx
    if 'longitude' in weights.dims:
      w_index['longitude'] = lon_idx
    return (dataset.isel(latitude=lat_idx, longitude=lon_idx),
            weights.isel(**w_index))


@dataclasses.dataclass
class ExtraTropicalRegion(Region):
  """regions.py:98-109 (threshold_lat is ignored by the reference, :108)."""

  threshold_lat: t.Optional[float] = 20

  def apply(self, dataset, weights):
    lat = NA(dataset.coord('latitude'), ('latitude',))
    region_weights = NA((np.abs(lat.data) >= 20).astype(float), ('latitude',))
    return dataset, weights * region_weights


@dataclasses.dataclass
class LandRegion(Region):
  """regions.py:112-138.

  `land_sea_mask` is an NA over ('latitude', 'longitude') in any order and
  `latitude` / `longitude` are its coordinate labels.  xarray multiplies
  `weights * land_weights` with an inner join on labels (after the reference
  casts the mask's labels to the dataset's coord dtype, :131-134), so after a
  SliceRegion the mask is subset to the surviving labels; restated here as an
  exact-label lookup.
  """

  land_sea_mask: NA
  latitude: np.ndarray = None
  longitude: np.ndarray = None
  threshold: t.Optional[float] = None

  def apply(self, dataset, weights):
    land_weights = self.land_sea_mask
    index = {}
    for name, labels in (('latitude', self.latitude),
                         ('longitude', self.longitude)):
      if labels is None:
        continue  # positional alignment
      want = dataset.coord(name)
      have = np.asarray(labels).astype(want.dtype)
      pos = {v: i for i, v in enumerate(have.tolist())}
      index[name] = np.array([pos[v] for v in want.tolist()], dtype=int)
    land_weights = land_weights.isel(**index)
    if self.threshold is not None:
      land_weights = (land_weights > self.threshold).astype(float)
    return dataset, weights * land_weights


@dataclasses.dataclass
class CombinedRegion(Region):
  """regions.py:141-158."""

  regions: list = dataclasses.field(default_factory=list)

  def apply(self, dataset, weights):
    for region in self.regions:
      dataset, weights = region.apply(dataset, weights)
    return dataset, weights
