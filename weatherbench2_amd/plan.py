"""Host-side planning: grid + regions -> the tables the HIP kernels consume.

For one grid (latitude, longitude coordinates, memory layout) and one ordered
set of regions this builds, once, and keeps resident on the device:

  w_row / w_col     latitude weights (metrics.py:40-60, restated with the same
                    numpy ops so the weights are bit-identical, including the
                    "weights inherit the latitude dtype" quirk) along whichever
                    slab axis is latitude, ones along the other;
  bands / segs      maximal runs of rows / columns whose membership multiplicity
                    is the same in every region;
  chunks            bands cut into <= rows_per_chunk rows (the unit of work of
                    one workgroup), padded to a multiple of 8 so that chunk c
                    always lands on XCD c % 8 (keeps a land-sea-mask stripe in
                    that XCD's L2);
  coef_band/seg     region x band / region x seg multiplicities;
  region_wsum       sum of each region's weights = xarray's sum_of_weights when
                    nothing is NaN (metrics.py:161-163).
"""
from __future__ import annotations

import dataclasses
import os
import threading
import typing as t

import numpy as np
import torch

from weatherbench2_amd import regions as regions_lib

LATLON = 'latlon'  # slabs are (latitude, longitude): rows = latitude
LONLAT = 'lonlat'  # slabs are (longitude, latitude): rows = longitude

DEFAULT_ROWS_PER_CHUNK = 16


# K3 (64 columns per wave, ~1 100 VALU per row): 5 rows per chunk with
# non-temporal member loads and 2-wave workgroups (0.445 ms per 13-slab launch
# against 0.452 / 0.456 / 0.457 at 6 / 7 / 8 rows and 0.469 at 4, same box:
# profiles/r03_k3_ab10_summary.txt; 12 and 16 rows are 4-6 % slower).
ENSEMBLE_ROWS_PER_CHUNK = int(os.environ.get('WB2HIP_ENS_ROWS_PER_CHUNK', 5))


# The fused energy-score pass (one wave per block of 8 members: ~100 VALU per
# row, a 16-sum fold per chunk): longer chunks than K3's amortise the fold.
ENERGY_ROWS_PER_CHUNK = int(os.environ.get('WB2HIP_ENERGY_ROWS_PER_CHUNK', 16))


def auto_rows_per_chunk(n_row: int, n_outer: int) -> int:
  """Rows per workgroup-chunk of the streaming kernel K1 for a launch of
  `n_outer` slabs (the ensemble kernels use ENSEMBLE_ROWS_PER_CHUNK).

  Measured on MI355X (profiles/r01_rows_per_chunk.md): 24-32 rows per chunk
  is best once the launch has thousands of workgroups (bigger chunks amortise
  the per-workgroup prologue/fold, but 48+ starves the tail); small launches
  (one 13-level unit) want finer chunks so that all 256 CUs get work.
  """
  target_workgroups = 4096
  rows = (n_row * max(int(n_outer), 1)) // target_workgroups
  return int(min(32, max(8, rows)))


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


def get_lat_weights(latitude: np.ndarray) -> np.ndarray:
  """Latitude/area weights, mean 1 (metrics.py:55-60)."""
  weights = _cell_area_from_latitude(np.deg2rad(np.asarray(latitude)))
  weights = weights / np.mean(weights)
  return weights


def _runs(signature: np.ndarray) -> np.ndarray:
  """Start offsets (plus the end) of maximal runs of equal rows."""
  n = signature.shape[0]
  change = np.ones(n, dtype=bool)
  if n > 1:
    change[1:] = np.any(signature[1:] != signature[:-1], axis=1)
  return np.concatenate([np.nonzero(change)[0], [n]]).astype(np.int32)


@dataclasses.dataclass
class ReductionPlan:
  layout: str
  n_row: int
  n_col: int
  region_names: list
  # host copies (numpy)
  w_lat: np.ndarray
  band_row0: np.ndarray
  seg_col0_host: np.ndarray
  chunk_row0_host: np.ndarray
  chunk_nrow_host: np.ndarray
  region_wsum_host: np.ndarray
  # device tables (torch, on `device`)
  device: torch.device = None
  w_row: torch.Tensor = None
  w_col: t.Optional[torch.Tensor] = None  # None = all ones
  wfield: t.Optional[torch.Tensor] = None
  # the same field as float32 when every value is a float32 number (an ERA5
  # land-sea mask, a thresholded mask): K1 reads half the bytes, same results
  wfield32: t.Optional[torch.Tensor] = None
  chunk_row0: torch.Tensor = None
  chunk_nrow: torch.Tensor = None
  seg_col0: torch.Tensor = None
  band_chunk0: torch.Tensor = None
  coef_band: torch.Tensor = None
  coef_seg: torch.Tensor = None
  region_wf: torch.Tensor = None
  region_wsum: torch.Tensor = None
  _pins: tuple = ()
  _eoff: dict = dataclasses.field(default_factory=dict)

  def seg_entries(self, tile_cols: int):
    """(seg_eoff device tensor, n_ts) for a column-tile width (wb2hip.h)."""
    hit = self._eoff.get(tile_cols)
    if hit is None:
      c = self.seg_col0_host.astype(np.int64)
      ntile = (c[1:] - 1) // tile_cols - c[:-1] // tile_cols + 1
      eoff = np.concatenate([[0], np.cumsum(ntile)]).astype(np.int32)
      hit = (torch.as_tensor(eoff).to(self.device), int(eoff[-1]))
      self._eoff[tile_cols] = hit
    return hit

  @property
  def n_chunk(self): return int(self.chunk_row0_host.shape[0])
  @property
  def n_seg(self): return int(self.seg_col0_host.shape[0] - 1)
  @property
  def n_band(self): return int(self.band_row0.shape[0] - 1)
  @property
  def n_region(self): return len(self.region_names)
  @property
  def nwf(self): return 2 if self.wfield is not None else 1


def _as_float32_field(field_rc, up):
  """The weight field as a float32 device tensor if that loses nothing."""
  if field_rc is None:
    return None
  f64 = np.asarray(field_rc, dtype=np.float64)
  f32 = f64.astype(np.float32)
  if not np.array_equal(f32.astype(np.float64), f64):
    return None
  return up(f32, torch.float32)


def build_plan(latitude: np.ndarray, longitude: np.ndarray, layout: str,
               regions: t.Optional[dict], device,
               rows_per_chunk: int = DEFAULT_ROWS_PER_CHUNK) -> ReductionPlan:
  """`regions`: ordered {name: Region or None}; None/{} means global only."""
  latitude = np.asarray(latitude)
  longitude = np.asarray(longitude)
  if not regions:
    regions = {'global': None}
  names = list(regions)
  specs = [regions_lib.decompose_region(regions[k], latitude, longitude)
           for k in names]

  w_lat = np.asarray(get_lat_weights(latitude), dtype=np.float64)
  if not (np.isfinite(w_lat).all() and (w_lat > 0).all()):
    raise NotImplementedError(
        'degenerate latitude weights (<= 0 or non-finite) are not supported')

  fields = [s.field for s in specs if s.field is not None]
  field = None
  if fields:
    field = fields[0]
    for f in fields[1:]:
      if f is not fields[0] and not np.array_equal(f, field):
        raise NotImplementedError(
            'more than one distinct 2-D weight field per pass; evaluate the '
            'regions in separate groups')
    if (field < 0).any() or not np.isfinite(field).all():
      raise NotImplementedError('negative or non-finite 2-D region weights')

  lat_mult = np.stack([s.lat_mult for s in specs], axis=1)  # [n_lat, R]
  lon_mult = np.stack([s.lon_mult for s in specs], axis=1)  # [n_lon, R]
  if layout == LATLON:
    row_mult, col_mult = lat_mult, lon_mult
    w_row, w_col = w_lat, np.ones(len(longitude))
    field_rc = field
  elif layout == LONLAT:
    row_mult, col_mult = lon_mult, lat_mult
    w_row, w_col = np.ones(len(longitude)), w_lat
    field_rc = None if field is None else np.ascontiguousarray(field.T)
  else:
    raise ValueError(f'unknown layout {layout}')
  n_row, n_col = row_mult.shape[0], col_mult.shape[0]

  band_row0 = _runs(row_mult)
  seg_col0 = _runs(col_mult)
  n_band, n_seg = len(band_row0) - 1, len(seg_col0) - 1
  coef_band = np.ascontiguousarray(
      row_mult[band_row0[:-1]].T.astype(np.float64))  # [R, n_band]
  coef_seg = np.ascontiguousarray(
      col_mult[seg_col0[:-1]].T.astype(np.float64))   # [R, n_seg]

  chunk_row0, chunk_nrow, band_chunk0 = [], [], [0]
  for b in range(n_band):
    r0, r1 = int(band_row0[b]), int(band_row0[b + 1])
    for r in range(r0, r1, rows_per_chunk):
      chunk_row0.append(r)
      chunk_nrow.append(min(rows_per_chunk, r1 - r))
    band_chunk0.append(len(chunk_row0))
  while len(chunk_row0) % 8:
    chunk_row0.append(0)
    chunk_nrow.append(0)

  region_wf = np.array([0 if s.field is None else 1 for s in specs],
                       dtype=np.int32)
  wsum = np.zeros(len(specs))
  for i, s in enumerate(specs):
    wl = s.lat_mult * w_lat
    if s.field is None:
      wsum[i] = np.sum(wl) * np.sum(s.lon_mult)
    else:
      wsum[i] = np.einsum('i,ij,j->', wl, s.field, s.lon_mult.astype(float))

  dev = torch.device(device)
  if dev.type == 'cuda' and dev.index is None:
    dev = torch.device('cuda', torch.cuda.current_device())

  def up(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(dev)

  return ReductionPlan(
      layout=layout, n_row=n_row, n_col=n_col, region_names=names,
      w_lat=w_lat, band_row0=band_row0, seg_col0_host=seg_col0,
      chunk_row0_host=np.array(chunk_row0, dtype=np.int32),
      chunk_nrow_host=np.array(chunk_nrow, dtype=np.int32),
      region_wsum_host=wsum, device=dev,
      w_row=up(w_row, torch.float64),
      w_col=None if layout == LATLON else up(w_col, torch.float64),
      wfield=None if field_rc is None else up(field_rc, torch.float64),
      wfield32=_as_float32_field(field_rc, up),
      chunk_row0=up(chunk_row0, torch.int32),
      chunk_nrow=up(chunk_nrow, torch.int32),
      seg_col0=up(seg_col0, torch.int32),
      band_chunk0=up(band_chunk0, torch.int32),
      coef_band=up(coef_band, torch.float64),
      coef_seg=up(coef_seg, torch.float64),
      region_wf=up(region_wf, torch.int32),
      region_wsum=up(wsum, torch.float64),
      _pins=tuple(regions.values()))


_PLAN_CACHE: dict = {}
_PLAN_LOCK = threading.Lock()


def cached_plan(latitude, longitude, layout, regions, device,
                rows_per_chunk: int = DEFAULT_ROWS_PER_CHUNK) -> ReductionPlan:
  """Plans are keyed on the coordinate bytes and the identity of the regions."""
  latitude = np.asarray(latitude)
  longitude = np.asarray(longitude)
  rkey = tuple((k, id(v)) for k, v in (regions or {'global': None}).items())
  key = (latitude.tobytes(), str(latitude.dtype), longitude.tobytes(),
         str(longitude.dtype), layout, rkey, str(device), rows_per_chunk)
  hit = _PLAN_CACHE.get(key)
  if hit is None:
    plan = build_plan(latitude, longitude, layout, regions, device,
                      rows_per_chunk)
    with _PLAN_LOCK:
      if len(_PLAN_CACHE) > 64:
        _PLAN_CACHE.clear()
      # the regions are kept alive with the plan: the key holds their id()s
      _PLAN_CACHE[key] = hit = (plan, regions)
  return hit[0]
