"""GPU metric operators with the reference's `Metric` protocol.

Same class names, constructor fields, `compute_chunk(forecast, truth, region,
skipna)` / `compute(...)` signatures, result layout (every input variable, all
non-spatial dims kept, no latitude/longitude) and error behaviour as
weatherbench2/metrics.py -- so they are drop-in values for
`config.Eval.metrics` -- but the arithmetic is ONE fused HIP pass per
(forecast, truth) chunk:

  reference                                   here
  ------------------------------------------  ---------------------------------
  MSE/RMSE/MAE/Bias each recompute f - t and  one read of f, t (and the
  run 2 einsums (metrics.py:236-359), ACC     climatology for ACC) feeds all
  three more (:377-414), all of it again per  five metrics and EVERY region
  region (evaluation.py:416-430)              (csrc/stream_reduce.hip)
  truth.sel(time=valid_time) and              no copies: per-slab index tables
  climatology.sel(dayofyear, hour) copies     gather truth/climatology slabs
  (evaluation.py:474-475, metrics.py:398-404) inside the kernel

Because the reference's loop asks for one (metric, region) at a time, results
of a pass are cached per chunk (the reference does the same for CRPS with
`dataset_safe_lru_cache`, metrics.py:775-780): the first call computes all
metrics of the pass for all regions announced through `fused_regions(...)`
(or for just the requested one), later calls are lookups.
"""
from __future__ import annotations

import contextlib
import logging
import dataclasses
import functools
import os
import threading
import typing as t

import numpy as np
import torch

from weatherbench2_amd import _lib
from weatherbench2_amd import engine
from weatherbench2_amd import plan as plan_lib
from weatherbench2_amd import program
from weatherbench2_amd import xarray_lite as xl
from weatherbench2_amd.regions import Region

REALIZATION = 'realization'
_SPATIAL = ('latitude', 'longitude')


# ---------------------------------------------------------------------------
# region announcement + per-chunk result cache
# ---------------------------------------------------------------------------
class _Announced(threading.local):
  """Per-thread stacks: what the enclosing loop announced it will ask for, plus
  the per-chunk caches of that thread (Beam's DirectRunner may call
  compute_chunk from several threads; nothing here is shared between them, so
  no lock is needed and the passes of different threads overlap on the GPU)."""

  def __init__(self):
    self.regions: list = []       # (ordered {name: Region|None}, signature)
    self.climatology: list = []   # climatology Datasets of an ACC in the loop
    self.depth = 0                # nesting of chunk scopes (fused_regions)
    self.scoped: dict = {}        # cache name -> {key: (pins, value)}
    self.memo: dict = {}          # per-scope memo (stamps, coordinate signatures)
    self.groupings: list = []     # memo of fused_regions' grouping (identity)
    self.wind: list = []          # (u_name, v_name) pairs the loop will ask for
    self.rows_per_chunk: list = []  # pinned K1 chunking (evaluate_chunks)
    self.lru: dict = {}           # cache name -> _LRU (calls outside any scope)


_ANNOUNCED = _Announced()


@contextlib.contextmanager
def chunk_scope():
  """Marks one pass of a metric x region loop over ONE (forecast, truth) chunk.

  Inside the scope input arrays are taken not to change, so results, uploads
  and aligned views are cached by identity without bound; everything is dropped
  when the outermost scope exits -- the next chunk can reuse the same buffers
  in place and never sees stale results (the reference's loop-local
  `dataset_safe_lru_cache(maxsize=1)`, metrics.py:775-780, has the same
  lifetime).  `fused_regions` opens a scope by itself."""
  st = _ANNOUNCED
  st.depth += 1
  try:
    yield
  finally:
    st.depth -= 1
    if st.depth == 0:
      st.scoped.clear()
      st.memo.clear()
      # results leave the pass here: a worker thread's private stream is made
      # visible to the default stream (engine._adopt_thread_stream)
      engine.publish_thread_stream()


def _field_key(region):
  """Identity of the 2-D weight field a region brings along (land-sea masks);
  None for pure latitude/longitude regions.  One fused pass carries at most
  one such field, so regions are grouped by it."""
  kind = type(region).__name__
  if kind == 'LandRegion':
    mask = getattr(region.land_sea_mask, 'data', region.land_sea_mask)
    if isinstance(mask, np.ndarray):  # same buffer = same field
      ident = (mask.__array_interface__['data'][0], mask.shape, mask.strides)
    else:
      ident = id(mask)
    return (ident, getattr(region, 'threshold', None))
  if kind == 'CombinedRegion':
    keys = [k for k in (_field_key(r) for r in region.regions)
            if k is not None]
    # a slice combined with ONE land-sea mask carries that mask's field and
    # shares its pass (scripts/evaluate.py:378-395: global_land,
    # extra-tropics_land and tropics_land are one group)
    return None if not keys else keys[0] if len(keys) == 1 else tuple(keys)
  return None


def _grouping(regions: t.Optional[dict]):
  """(active, groups, member) of a region dict: one group per distinct 2-D
  weight field.  Memoised per thread on the identity of the dict and of its
  values (a loop announces the same dict a dozen times per chunk)."""
  st = _ANNOUNCED
  sig = (id(regions), tuple((k, id(v)) for k, v in regions.items())
         if regions else None)
  for known_sig, pinned, value in st.groupings:
    if known_sig == sig and pinned is regions:
      return value
  active = dict(regions) if regions else {'__none__': None}
  keys = {k: _field_key(v) for k, v in active.items()}
  distinct = []
  for fk in keys.values():
    if fk is not None and fk not in distinct:
      distinct.append(fk)
  groups, member = [], {}
  for gi, fk in enumerate(distinct or [None]):
    group = {k: v for k, v in active.items()
             if keys[k] == fk or (gi == 0 and keys[k] is None)}
    for k in group:
      member[k] = gi
    groups.append((group, tuple((k, id(v)) for k, v in group.items())))
  value = (active, groups, member)
  st.groupings.append((sig, regions, value))
  del st.groupings[:-8]
  return value


@contextlib.contextmanager
def fused_regions(regions: t.Optional[dict]):
  """Announce the regions a loop is about to iterate so one pass serves all
  (one pass per distinct 2-D weight field: field-free regions ride along with
  the first group).  Also a `chunk_scope`: wrap exactly one chunk's loop."""
  _ANNOUNCED.regions.append(_grouping(regions))
  try:
    with chunk_scope():
      yield
  finally:
    _ANNOUNCED.regions.pop()


@contextlib.contextmanager
def fused_climatology(climatology):
  """Announce that an ACC with this climatology is part of the loop: the first
  deterministic pass over a chunk then reads the climatology too (12 instead of
  8 + 12 bytes per point for MSE/MAE/Bias/RMSE followed by ACC)."""
  _ANNOUNCED.climatology.append(climatology)
  try:
    yield
  finally:
    _ANNOUNCED.climatology.pop()


@contextlib.contextmanager
def fused_wind_vectors(pairs):
  """Announce the (u_name, v_name) pairs of every wind-vector metric of a loop
  (scripts/evaluate.py:279-311: the pressure-level and the 10 m wind): the
  first one asked for reads them all in ONE launch."""
  _ANNOUNCED.wind.append(tuple(pairs))
  try:
    yield
  finally:
    _ANNOUNCED.wind.pop()


@contextlib.contextmanager
def pinned_rows_per_chunk(rows: t.Optional[int]):
  """Fixes the row-chunking of the streaming kernel for every pass inside:
  chunk partials are summed in chunk order, so results are bit-identical
  between launches of different sizes only when the chunking does not follow
  the launch size (`plan.auto_rows_per_chunk`).  `evaluation.evaluate_chunks`
  pins it, which makes its result independent of `batch_chunks`."""
  _ANNOUNCED.rows_per_chunk.append(rows)
  try:
    yield
  finally:
    _ANNOUNCED.rows_per_chunk.pop()


def _rows_per_chunk(n_row: int, n_outer: int) -> int:
  pinned = _ANNOUNCED.rows_per_chunk[-1] if _ANNOUNCED.rows_per_chunk else None
  return int(pinned) if pinned else plan_lib.auto_rows_per_chunk(n_row, n_outer)


def _region_set_for(region) -> tuple[dict, str]:
  """Returns (ordered region dict to evaluate, key of the requested one)."""
  return _region_set_sig(region)[:2]


def _region_set_sig(region) -> tuple[dict, str, tuple]:
  if _ANNOUNCED.regions:
    active, groups, member = _ANNOUNCED.regions[-1]
    for k, v in active.items():
      if v is region:
        group, sig = groups[member[k]]
        return group, k, sig
  return ({'__requested__': region}, '__requested__',
          (('__requested__', id(region)),))


_ALL = '__all_regions__'  # by_region key: (stacked tensor, region names)

# What a COMPANION variable of a fused launch (another variable of the chunk, a
# second wind pair) may raise while its inputs are prepared -- a foreign dtype,
# a grid that does not match, a missing label: such a variable is left out and
# reports when it is asked for.  Anything else (a failed launch, an allocator
# or library fault) is not an input error and propagates.
_INPUT_ERRORS = (TypeError, ValueError, KeyError, NotImplementedError)


def _fused(pass_fn, region, regions: t.Optional[dict]):
  """pass_fn(region) -> (geo, by_region, ...) for the requested region -- or,
  on the all-regions path (inside `_all_regions(regions)`), for one
  representative of every field group, with the by_region dicts merged."""
  if regions is None:
    return pass_fn(region)
  _, groups, _ = _ANNOUNCED.regions[-1]
  if len(groups) == 1:  # the common case: one pass answers every region
    return pass_fn(next(iter(groups[0][0].values())))
  out, merged = None, {}
  for group, _ in groups:
    res = pass_fn(next(iter(group.values())))
    merged.update(res[1].materialized() if isinstance(res[1], _ByRegion)
                  else res[1])
    out = out or res
  merged.pop(_ALL, None)  # the stacked tensor of ONE pass is not all regions
  return (out[0], merged) + tuple(out[2:])


class _LRU:
  """Small bounded cache for calls made outside any chunk scope."""

  def __init__(self, maxsize):
    self.maxsize = maxsize
    self.items: list = []  # (key, pins, value), most recent last

  def get(self, key):
    for i, (k, _, v) in enumerate(self.items):
      if k == key:
        self.items.append(self.items.pop(i))
        return v
    return None

  def put(self, key, pins, value):
    self.items = [it for it in self.items if it[0] != key]
    self.items.append((key, pins, value))
    while len(self.items) > self.maxsize:
      self.items.pop(0)

  def clear(self):
    self.items.clear()


class _Cache:
  """Per-thread cache: unbounded and identity-keyed inside a chunk scope
  (dropped when the scope ends), a small LRU outside."""

  def __init__(self, name: str, maxsize: int):
    self.name, self.maxsize = name, maxsize

  def _lru(self) -> _LRU:
    lru = _ANNOUNCED.lru.get(self.name)
    if lru is None:
      lru = _ANNOUNCED.lru[self.name] = _LRU(self.maxsize)
    return lru

  def get(self, key):
    st = _ANNOUNCED
    if st.depth > 0:
      hit = st.scoped.get(self.name, {}).get(key)
      return None if hit is None else hit[1]
    return self._lru().get(key)

  def put(self, key, pins, value):
    st = _ANNOUNCED
    if st.depth > 0:
      st.scoped.setdefault(self.name, {})[key] = (pins, value)
    else:
      self._lru().put(key, pins, value)

  def clear(self):
    _ANNOUNCED.scoped.pop(self.name, None)
    self._lru().clear()


_RESULTS = _Cache('results', 8)    # fused pass results
_DEVICE = _Cache('device', 12)     # host array -> device tensor uploads
_ALIGNED = _Cache('aligned', 4)    # (forecast, truth) -> label-aligned views


def _inputs(forecast, truth) -> tuple:
  """Datasets as the reference's `forecast - truth` would see them: converted
  at the boundary and inner-joined on their shared dimension coordinates.  The
  aligned pair is cached by identity so that every metric of a chunk sees the
  SAME array objects (the per-chunk result caches key on them)."""
  forecast, truth = xl.as_dataset(forecast), xl.as_dataset(truth)

  def members(ds):  # a Dataset object whose variables / coords were replaced
    return tuple((k, id(v.data)) for k, v in ds.data_vars.items()) + tuple(
        (k, id(c.data) if isinstance(c, xl.DataArray) else id(c))
        for k, c in ds.coords.items())
  key = (id(forecast), id(truth), members(forecast), members(truth))
  hit = _ALIGNED.get(key)
  if hit is None:
    hit = xl.align_inner(forecast, truth, exclude=_SPATIAL)
    _ALIGNED.put(key, (forecast, truth), hit)
  return hit


def clear_caches():
  _RESULTS.clear()
  _DEVICE.clear()
  _ALIGNED.clear()


def _stamp(a) -> tuple:
  """Cache identity of an input array.

  Torch tensors carry a version counter that every in-place op bumps: (id,
  version, pointer) is exact.  NumPy arrays have no such counter: inside a
  chunk scope (where inputs do not change by contract and everything cached is
  dropped at the end) identity + layout is the stamp; OUTSIDE a scope the whole
  buffer is hashed (xxh3, GB/s -- comparable to the upload the cache saves), so
  a caller that refills a buffer in place between two bare `compute_chunk`
  calls can never get the previous chunk's result (the reference's
  dataset_safe_lru_cache compares whole arrays too, utils.py:322-350)."""
  if isinstance(a, torch.Tensor):
    return (id(a), a._version, a.data_ptr())
  if isinstance(a, xl.SlabGather):
    # a gather is what it reads: the index and the base (a host base is hashed
    # over the slabs the gather touches only -- _to_device compacts it first)
    base = a.base
    if isinstance(base, np.ndarray) and _ANNOUNCED.depth == 0:
      small, _ = a.compact_host()
      base_stamp = ('gathered', _stamp(small)[-1])
    else:
      base_stamp = _stamp(base)
    return ('gather', base_stamp, a.index.shape, engine.digest(a.index))
  if isinstance(a, xl.SlabConcat):
    if _ANNOUNCED.depth > 0:  # inputs do not change inside a chunk scope
      return ('concat', id(a))
    return ('concat', tuple(_stamp(b) for b in a.bases), a.index.shape,
            engine.digest(a.index))
  if isinstance(a, np.ndarray):
    ident = (id(a), a.__array_interface__['data'][0], a.shape, a.strides,
             a.dtype.str)
    if _ANNOUNCED.depth > 0 or a.size == 0:
      return ident
    if a.dtype.kind not in 'biufc':  # object / string arrays: never cached
      return ident + (object(),)
    return ident + (engine.digest(a),)
  return (id(a),)


def _coord_sig(*datasets) -> tuple:
  """Identity of everything besides the data that shapes a result: variable
  dims and the values of all coordinates (small arrays, hashed in full).
  Memoised per Dataset object inside a chunk scope."""
  st = _ANNOUNCED
  out = []
  for ds in datasets:
    key = ('coord_sig', id(ds))
    hit = st.memo.get(key) if st.depth > 0 else None
    if hit is None or hit[0] is not ds:
      parts = [tuple((k, v.dims) for k, v in ds.data_vars.items())]
      for k, c in ds.coords.items():
        if isinstance(c, xl.DataArray):
          v, dims = np.asarray(c.values), tuple(c.dims)
        else:
          v, dims = np.asarray(c), None
        if v.dtype == object:
          parts.append((k, dims, tuple(v.ravel().tolist())))
        else:
          parts.append((k, dims, v.shape, v.dtype.str,
                        np.ascontiguousarray(v).tobytes()))
      hit = (ds, hash(tuple(parts)))
      if st.depth > 0:
        st.memo[key] = hit
    out.append(hit[1])
  return tuple(out)


_BIG_UPLOAD_BYTES = 2 << 30


def _to_device(data, device, allow_gather: bool = False):
  """`data` on `device`.  A SlabGather stays a gather (over a device base)
  when the caller can read through slab tables (`allow_gather`), otherwise it is
  materialised on the device; a host base crosses PCIe as the DISTINCT slabs
  the gather touches, never as the whole array."""
  if isinstance(data, torch.Tensor) and data.device == device:
    return data
  if isinstance(data, xl.SlabGather):
    base, index = data.base, data.index
    if isinstance(base, np.ndarray):
      base, index = data.compact_host()
    dev_base = _to_device(base, device)
    gathered = xl.SlabGather(dev_base, index)
    if allow_gather and not gathered.has_missing:
      return gathered
    return gathered.materialize()
  if isinstance(data, xl.SlabConcat):
    if not data.on_device:
      data = xl.SlabConcat([_to_device(b, device) for b in data.bases],
                           data.index)
    return data if allow_gather else data.materialize(device)
  if isinstance(data, np.ndarray) and data.nbytes >= _BIG_UPLOAD_BYTES:
    import warnings
    warnings.warn(
        f'uploading a {data.nbytes / 2**30:.1f} GiB host array for one chunk; '
        'inputs that recur across chunks (climatology, truth) are better made '
        'resident once (evaluation.make_resident)', stacklevel=3)
  key = _stamp(data)
  hit = _DEVICE.get(key)
  if hit is not None:
    return hit
  ten = engine.as_device_tensor(data, device)
  _DEVICE.put(key, (data,), ten)
  return ten


# ---------------------------------------------------------------------------
# chunk geometry
# ---------------------------------------------------------------------------
@dataclasses.dataclass
class _Geometry:
  layout: str
  out_dims: tuple
  out_shape: tuple
  latitude: np.ndarray
  longitude: np.ndarray

  @property
  def n_outer(self):
    return int(np.prod(self.out_shape, dtype=np.int64))


def _spatial_last(da: xl.DataArray, layout: t.Optional[str]):
  """Returns (data with the two spatial dims last, non-spatial dims, layout)."""
  for d in _SPATIAL:
    if d not in da.dims:
      raise ValueError(f'{d!r} missing from dims {da.dims}')
  rest = tuple(d for d in da.dims if d not in _SPATIAL)
  tail = tuple(d for d in da.dims if d in _SPATIAL)
  if layout is None:
    layout = plan_lib.LATLON if tail == _SPATIAL else plan_lib.LONLAT
  want = _SPATIAL if layout == plan_lib.LATLON else _SPATIAL[::-1]
  if da.dims[-2:] == want:
    return da.data, rest, layout
  moved = da.transpose(*rest, *want)
  data = moved.data
  if isinstance(data, (xl.SlabGather, xl.SlabConcat)):
    data = data.materialize() if getattr(
        data, 'on_device', False) else data.materialize_host()
  data = data.contiguous() if isinstance(data, torch.Tensor) else (
      np.ascontiguousarray(data))
  return data, rest, layout


def _slab_table(out_dims, out_shape, in_dims, in_shape, extra=None):
  """int64[n_outer] slab index of an input broadcast (by name) to out_dims."""
  if tuple(in_dims) == tuple(out_dims) and extra is None:
    if tuple(in_shape) != tuple(out_shape):
      raise ValueError(f'shape mismatch {in_shape} vs {out_shape}')
    return None
  table = np.zeros(out_shape, dtype=np.int64)
  stride = 1
  for d, n in reversed(list(zip(in_dims, in_shape))):
    if d not in out_dims:
      raise ValueError(f'dimension {d!r} not in {out_dims}')
    ax = out_dims.index(d)
    if out_shape[ax] != n:
      raise ValueError(f'size mismatch on {d!r}: {n} vs {out_shape[ax]}')
    shape = [1] * len(out_shape)
    shape[ax] = n
    table = table + (np.arange(n, dtype=np.int64) * stride).reshape(shape)
    stride *= n
  if extra is not None:
    table = table + extra
  return np.ascontiguousarray(table).ravel()


def _coord_values(ds: xl.Dataset, name: str) -> np.ndarray:
  c = ds.coords[name]
  return np.asarray(c.values if isinstance(c, xl.DataArray) else c)


def _geometry(forecast: xl.Dataset, fvar: xl.DataArray, others) -> tuple:
  """Common output dims (xarray broadcast order) for one variable."""
  fdata, frest, layout = _spatial_last(fvar, None)
  out_dims = list(frest)
  sizes = dict(fvar.sizes)
  prepared = [(fdata, frest)]
  for da in others:
    data, rest, _ = _spatial_last(da, layout)
    for d in rest:
      if d not in out_dims:
        out_dims.append(d)
        sizes[d] = da.sizes[d]
    prepared.append((data, rest))
  out_dims = tuple(out_dims)
  out_shape = tuple(sizes[d] for d in out_dims)
  geo = _Geometry(layout, out_dims, out_shape,
                  _coord_values(forecast, 'latitude'),
                  _coord_values(forecast, 'longitude'))
  return geo, prepared


def _check_grid(geo: _Geometry, data) -> None:
  n_lat, n_lon = len(geo.latitude), len(geo.longitude)
  want = (n_lat, n_lon) if geo.layout == plan_lib.LATLON else (n_lon, n_lat)
  if tuple(data.shape[-2:]) != want:
    raise ValueError(f'spatial shape {tuple(data.shape[-2:])} does not match '
                     f'the coordinates {want}')


# ---------------------------------------------------------------------------
# climatology gather tables (metrics.py:63-81, 394-404)
# ---------------------------------------------------------------------------
def _get_climatology_chunk(climatology: xl.Dataset, truth: xl.Dataset) -> dict:
  """Maps each truth variable to its climatology DataArray (KeyError like ref)."""
  names = list(truth.keys())
  if all(k in climatology for k in names):
    return {k: climatology[k] for k in names}
  clim_var_dict = {str(k) + '_mean': k for k in names}
  not_found = set(names).difference(climatology.keys())
  not_found_means = set(clim_var_dict).difference(climatology.keys())
  if not_found and not_found_means:
    raise KeyError(f'Did not find {not_found} keys in climatology. Appending '
                   "'mean' did not help.")
  return {v: climatology[k] for k, v in clim_var_dict.items()}


def _label_positions(have: np.ndarray, want: np.ndarray, what: str):
  """Positions of the labels `want` in the index `have` (KeyError like .sel).
  Numeric labels (dayofyear, hour, level) by binary search -- this runs once per
  chunk; a dict of the whole index, rebuilt per call, cost more than the
  lookup."""
  have, want = np.asarray(have), np.asarray(want)
  if have.dtype.kind in 'iuf' and want.dtype.kind in 'iuf' and have.size:
    sorter = np.argsort(have, kind='stable')
    at = np.searchsorted(have, want.ravel(), sorter=sorter)
    at = np.minimum(at, have.size - 1)
    pos = sorter[at]
    miss = have[pos] != want.ravel()
    if miss.any():
      raise KeyError(f'{what} label {want.ravel()[miss][0]!r} not found in '
                     'climatology')
    return pos.astype(np.int64).reshape(np.shape(want))
  pos = {v: i for i, v in enumerate(xl.label_list(have))}
  try:
    return np.array([pos[v] for v in xl.label_list(want)],
                    dtype=np.int64).reshape(np.shape(want))
  except KeyError as e:
    raise KeyError(f'{what} label {e} not found in climatology') from e


def _dayofyear_hour(vt: np.ndarray):
  """(dayofyear 1.., hour of day) of datetime64 values, like pandas'
  DatetimeIndex.dayofyear / .hour (metrics.py:71-75) -- in NumPy: a
  DatetimeIndex per chunk cost 0.13 ms."""
  vt = np.asarray(vt)
  if vt.dtype.kind != 'M':
    import pandas as pd
    idx = pd.DatetimeIndex(vt.ravel())
    return (np.asarray(idx.dayofyear).reshape(vt.shape),
            np.asarray(idx.hour).reshape(vt.shape))
  days = vt.astype('datetime64[D]')
  year0 = vt.astype('datetime64[Y]').astype('datetime64[D]')
  doy = (days - year0).astype(np.int64) + 1
  hour = (vt.astype('datetime64[h]') - days.astype('datetime64[h]')).astype(
      np.int64)
  return doy, hour


_CLIM_TABLES: dict = {}  # content key -> table (bounded; see _climatology_slabs)


def _climatology_slabs(climatology: xl.Dataset, cvar: xl.DataArray,
                       forecast: xl.Dataset, geo: _Geometry, crest: tuple):
  """int64[n_outer]: which climatology slab each output slab subtracts.

  Pure label work on small coordinate arrays; memoised on their CONTENT (valid
  times, levels, the climatology's dayofyear / hour / level labels, the output
  layout), so a stream of chunks with recurring time stamps -- or the five
  metrics of one loop -- builds each table once."""
  st = _ANNOUNCED
  memo_key = None
  if st.depth > 0:  # the variables of one chunk share their tables
    memo_key = ('clim_slabs', id(climatology), id(forecast), crest,
                geo.out_dims, geo.out_shape,
                tuple(cvar.sizes[d] for d in crest))
    hit = st.memo.get(memo_key)
    if hit is not None and hit[0] is climatology and hit[1] is forecast:
      return hit[2]
    table = _climatology_slabs_by_content(climatology, cvar, forecast, geo,
                                          crest)
    st.memo[memo_key] = (climatology, forecast, table)
    return table
  return _climatology_slabs_by_content(climatology, cvar, forecast, geo, crest)


_LABEL_BYTES: dict = {}  # id(coordinate array) -> (array, content key)


def _valid_times(forecast: xl.Dataset):
  """(valid times of the chunk, the dims they run over)."""
  if forecast.has_dim('init_time'):
    vt = forecast.coords['valid_time']
    if not isinstance(vt, xl.DataArray):
      raise ValueError('valid_time must be a coordinate over (init_time, lead)')
    time_dims = tuple(vt.dims)  # e.g. (init_time, lead_time / prediction_timedelta)
    vt = vt.values
  else:
    time_dims = ('time',)
    vt = _coord_values(forecast, 'time')
  return np.asarray(vt), time_dims


def _climatology_slabs_by_content(climatology, cvar, forecast, geo, crest):
  vt, time_dims = _valid_times(forecast)

  def label_bytes(ds, name):
    if name not in ds.coords:
      return None
    v = _coord_values(ds, name)
    # coordinate arrays recur (the climatology's always, the chunks' level
    # array usually): remembered per array OBJECT, kept alive here
    hit = _LABEL_BYTES.get(id(v))
    if hit is None or hit[0] is not v:
      if len(_LABEL_BYTES) >= 64:
        _LABEL_BYTES.clear()
      hit = _LABEL_BYTES[id(v)] = (
          v, (v.dtype.str, v.shape, np.ascontiguousarray(v).tobytes()))
    return hit[1]
  key = (vt.dtype.str, vt.shape, np.ascontiguousarray(vt).tobytes(), time_dims,
         geo.out_dims, geo.out_shape, crest,
         tuple(cvar.sizes[d] for d in crest),
         label_bytes(climatology, 'dayofyear'), label_bytes(climatology, 'hour'),
         label_bytes(climatology, 'level'), label_bytes(forecast, 'level'))
  hit = _CLIM_TABLES.get(key)
  if hit is None:
    hit = _climatology_slabs_build(climatology, cvar, forecast, geo, crest, vt,
                                   time_dims)
    if len(_CLIM_TABLES) >= 256:
      _CLIM_TABLES.clear()
    _CLIM_TABLES[key] = hit
  return hit


def _climatology_strides(crest, sizes) -> dict:
  unknown = set(crest) - {'dayofyear', 'hour', 'level'}
  if unknown:
    raise ValueError(f'unsupported climatology dims {unknown}')
  stride, strides = 1, {}
  for d, n in reversed(list(zip(crest, sizes))):
    strides[d] = stride
    stride *= n
  return strides


_LABEL_LUTS: dict = {}  # id(label array) -> (array, lookup table or None)


def _small_label_positions(have: np.ndarray, want: np.ndarray, what: str):
  """_label_positions for an index of small non-negative integers (dayofyear
  1..366, hour 0..23) through a dense lookup table remembered per label array
  OBJECT: the per-chunk cost of a climatology gather is two tiny fancy
  indexings instead of two argsorts."""
  hit = _LABEL_LUTS.get(id(have))
  if hit is None or hit[0] is not have:
    lut = None
    if have.dtype.kind in 'iu' and have.size and have.min() >= 0 and (
        have.max() < 4096) and len(np.unique(have)) == have.size:
      lut = np.full(int(have.max()) + 2, -1, dtype=np.int64)
      lut[have] = np.arange(have.size, dtype=np.int64)
    if len(_LABEL_LUTS) >= 64:
      _LABEL_LUTS.clear()
    hit = _LABEL_LUTS[id(have)] = (have, lut)
  lut = hit[1]
  if lut is None or want.dtype.kind not in 'iu':
    return _label_positions(have, want, what)
  pos = lut[np.clip(want, -1, lut.size - 1)]
  if (pos < 0).any():
    bad = np.asarray(want).ravel()[(pos < 0).ravel()][0]
    raise KeyError(f'{what} label {bad!r} not found in climatology')
  return pos


def _climatology_time_values(climatology, crest, sizes, vt,
                             memo: t.Optional[dict] = None) -> np.ndarray:
  """The time part of the climatology slab number of every valid time in `vt`
  (same shape): position of its dayofyear (and hour) label times the stride of
  that dim (metrics.py:398-404: climatology.sel(dayofyear=..., hour=...)).
  `memo`: shared by the gathers of one chunk (one calendar conversion)."""
  strides = _climatology_strides(crest, sizes)
  when = None if memo is None else memo.get('doy_hour')
  if when is None or when[0] is not vt:
    when = (vt,) + tuple(_dayofyear_hour(vt))
    if memo is not None:
      memo['doy_hour'] = when
  _, doy, hour = when
  tpart = _small_label_positions(_coord_values(climatology, 'dayofyear'), doy,
                                 'dayofyear') * strides['dayofyear']
  if 'hour' in climatology.coords and 'hour' in crest:
    tpart = tpart + _small_label_positions(
        _coord_values(climatology, 'hour'), hour, 'hour') * strides['hour']
  return tpart


def _climatology_structure(climatology, cvar, forecast, geo, crest, vt_shape,
                           time_dims):
  """(cell, base): output slab k of a variable subtracts climatology slab
  values.ravel()[cell[k]] + base[k], `values` = _climatology_time_values of
  the chunk's valid times.  Which (time, lead) cell a slab belongs to and its
  level part do not depend on the valid times themselves: the same for every
  chunk of one structure (program.py replays with new `values` only)."""
  strides = _climatology_strides(crest, [cvar.sizes[d] for d in crest])
  shape = [1] * len(geo.out_shape)
  for d, n in zip(time_dims, vt_shape):
    if d not in geo.out_dims:
      raise ValueError(f'time dim {d!r} missing from the forecast variable')
    shape[geo.out_dims.index(d)] = n
  order = [d for d in geo.out_dims if d in time_dims]
  n_cell = int(np.prod(vt_shape, dtype=np.int64))
  cells = np.arange(n_cell, dtype=np.int64).reshape(vt_shape)
  cells = np.transpose(cells, [time_dims.index(d) for d in order])
  cell = np.zeros(geo.out_shape, dtype=np.int64) + cells.reshape(shape)
  base = np.zeros(geo.out_shape, dtype=np.int64)
  if 'level' in crest:
    if 'level' not in geo.out_dims:
      raise ValueError('climatology has a level dim but the forecast has none')
    lv = _label_positions(_coord_values(climatology, 'level'),
                          _coord_values(forecast, 'level'), 'level')
    shape = [1] * len(geo.out_shape)
    shape[geo.out_dims.index('level')] = len(lv)
    base = base + (lv * strides['level']).reshape(shape)
  return (np.ascontiguousarray(cell).ravel(),
          np.ascontiguousarray(base).ravel())


def _climatology_slabs_build(climatology, cvar, forecast, geo, crest, vt,
                             time_dims):
  values = _climatology_time_values(
      climatology, crest, [cvar.sizes[d] for d in crest], vt)
  cell, base = _climatology_structure(climatology, cvar, forecast, geo, crest,
                                      np.shape(vt), time_dims)
  return np.ascontiguousarray(values).ravel()[cell] + base


def _climatology_gather(climatology, cvar, forecast, geo, crest):
  """What a chunk program needs to read this climatology variable for OTHER
  chunks of the structure: {'cell', 'base'} (structural) and values(other) ->
  int64[n_cell] (per chunk: label work on the chunk's valid times only)."""
  vt, time_dims = _valid_times(forecast)
  sizes = [cvar.sizes[d] for d in crest]
  cell, base = _climatology_structure(climatology, cvar, forecast, geo, crest,
                                      np.shape(vt), time_dims)

  seen: dict = {}   # valid times -> values (a valid time comes back with
                    # every lead that reaches it; read-only for the callers)

  def values(other, memo=None, shape=np.shape(vt), dims=time_dims):
    when = None if memo is None else memo.get('valid_times')
    if when is None or when[0] is not other:
      when = (other,) + _valid_times(other)
      if memo is not None:
        memo['valid_times'] = when
    _, vt2, dims2 = when
    if np.shape(vt2) != shape or dims2 != dims:
      raise ValueError('the chunk\'s valid times have another layout')
    key = np.ascontiguousarray(vt2).tobytes()
    hit = seen.get(key)
    if hit is None:
      if len(seen) >= 4096:
        seen.clear()
      hit = seen[key] = np.ascontiguousarray(_climatology_time_values(
          climatology, crest, sizes, vt2, memo), dtype=np.int64).ravel()
      hit.setflags(write=False)
    return hit
  return {'cell': cell, 'base': base, 'values': values,
          'key': (id(climatology), tuple(crest), tuple(sizes))}


def _physical_slabs(x: torch.Tensor, table, n_row: int, n_col: int):
  """(tensor for the engine, slab table) of an input whose logical layout is
  [..., n_row, n_col] with `table` indexing its slabs as if it were contiguous.

  Contiguous tensors are simply flattened.  A strided VIEW whose 2-D slabs are
  intact (overlapping `as_strided` windows -- evaluation.
  select_truth_at_valid_time --, `expand`ed broadcasts, slices with a step) is
  handed over as it is: the table is rewritten to physical slab offsets from
  the view's strides, so the selection costs no copy at all."""
  if isinstance(x, xl.SlabGather):
    index = x.index.ravel()
    return (x.base.reshape(-1, n_row, n_col),
            index if table is None else index[table])
  if x.is_contiguous():
    return x.reshape(-1, n_row, n_col), table
  se = n_row * n_col
  st, sh = x.stride(), tuple(x.shape)
  ok = (x.dim() >= 2 and st[-1] == 1 and st[-2] == n_col
        and all(s >= 0 and s % se == 0 for s in st[:-2]))
  if not ok:
    return x.contiguous().reshape(-1, n_row, n_col), table
  phys = np.zeros(sh[:-2], dtype=np.int64)
  for ax, (n, s) in enumerate(zip(sh[:-2], st[:-2])):
    shape = [1] * (len(sh) - 2)
    shape[ax] = n
    phys = phys + (np.arange(n, dtype=np.int64) * (s // se)).reshape(shape)
  phys = np.ascontiguousarray(phys).ravel()
  return x, (phys if table is None else phys[table])


# ---------------------------------------------------------------------------
# the fused passes
# ---------------------------------------------------------------------------
def _prepare_inputs(geo, arrays, tables, device):
  """The inputs of one variable's pass on the device, in one dtype:
  (tensors or lazy slab arrays, slab tables, dtype)."""
  arrays, tables = list(arrays), list(tables)
  for i, (a, tb) in enumerate(zip(arrays, tables)):
    # a host array read through a table that touches only some of its slabs (a
    # climatology gathered by dayofyear / hour): only those cross PCIe
    if isinstance(a, np.ndarray) and tb is not None and a.ndim >= 2 and (
        a.flags.c_contiguous):
      n_slab = a.size // max(a.shape[-1] * a.shape[-2], 1)
      if len(np.unique(tb)) < n_slab:
        arrays[i], tables[i] = xl.SlabGather(a, tb), None
  tensors = [_to_device(a, device, allow_gather=True) for a in arrays]
  dtype = tensors[0].dtype
  for x in tensors[1:]:
    dtype = torch.promote_types(dtype, x.dtype)
  if dtype not in (torch.float32, torch.float64):
    dtype = torch.float64

  def cast(x):
    if x.dtype == dtype:
      return x
    if isinstance(x, (xl.SlabGather, xl.SlabConcat)):
      x = x.materialize()  # never convert a whole resident base
    return x.to(dtype)
  tensors = [cast(x) for x in tensors]
  for x in tensors:
    _check_grid(geo, x)
  return tensors, tables, dtype


def _slab_addresses(x, table, n_row: int, n_col: int, n_outer: int):
  """(int64[n_outer] device byte addresses of the slabs of input `x` read
  through `table`, what must stay alive until the launch is enqueued)."""
  if isinstance(x, xl.SlabConcat):
    flat = x.addresses().ravel()
    return (flat if table is None else flat[table]), x.bases
  flat, tb = _physical_slabs(x, table, n_row, n_col)
  step = n_row * n_col * flat.element_size()
  if tb is None:
    tb = np.arange(n_outer, dtype=np.int64)
  return flat.data_ptr() + np.asarray(tb, dtype=np.int64) * step, flat


class _ByRegion(dict):
  """{region key: metrics[NMETRIC, ...]} over the stacked result of a pass,
  sliced on first use (a loop over all regions only ever takes `_ALL`).

  `cast(torch dtype)` gives the same stack in another dtype out of ONE
  conversion of the whole launch's result (shared by every variable of the
  launch): the reference's float32 results are views of it instead of one
  `.to()` kernel per (variable, metric)."""

  def __init__(self, dev, names, cast=None):
    super().__init__()
    self._dev, self._names = dev, list(names)
    self._cast, self._as = cast, {}
    dict.__setitem__(self, _ALL, (dev, self._names))

  def __missing__(self, key):
    try:
      value = self._dev[:, self._names.index(key)]
    except ValueError:
      raise KeyError(key) from None
    self[key] = value
    return value

  def __contains__(self, key):
    return dict.__contains__(self, key) or key in self._names

  def get(self, key, default=None):
    try:
      return self[key]
    except (KeyError, ValueError):
      return default

  def materialized(self) -> dict:
    return {k: self[k] for k in self._names}

  def as_dtype(self, dtype) -> '_ByRegion':
    want = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
    if want == self._dev.dtype or self._cast is None:
      return self
    if want not in self._as:
      self._as[want] = _ByRegion(self._cast(want), self._names)
    return self._as[want]


_AUX: dict = {}   # id(host field) -> (host field, device tensor)


def _resident_aux(aux, device) -> torch.Tensor:
  """A mode's auxiliary 2-D field (SEEPS: the masked dry fraction, one per
  climatology) on the device, uploaded once per host array OBJECT (kept alive
  here; the metric objects cache their fields)."""
  if isinstance(aux, torch.Tensor):
    return aux
  hit = _AUX.get((id(aux), str(device)))
  if hit is not None and hit[0] is aux:
    return hit[1]
  ten = torch.as_tensor(np.ascontiguousarray(aux, dtype=np.float64)).to(device)
  if len(_AUX) >= 16:
    _AUX.clear()
  _AUX[(id(aux), str(device))] = (aux, ten)
  return ten


def _run_group(mode, entries, region, skipna, aux=None, scalar=0.0,
               pairs=None, wind_out=None):
  """One fused pass (K1 + K2) per launch signature over the variables in
  `entries` = [(geo, arrays, tables)]: variables that share grid, layout and
  dtype -- every variable of an ERA5-style chunk -- are read by ONE launch
  whose slabs are addressed one by one (wb2_stream_partials_addr); the fold is
  per slab, so every variable's numbers are those of a launch of its own.
  Returns the list of {region_key: metrics[NMETRIC, ...]} dicts, in order.

  `pairs` = [(i_u, i_v)]: entries that are the u and v of a wind-vector metric
  (same output dims).  Where the launch has a pair kernel their wind-vector
  numbers come from the same read (wb2_det_wind_suite_step: the reference
  derives them from the same `diff`, metrics.py:283-301) and are filed in
  `wind_out[(i_u, i_v)]` as {region_key: metrics[NMETRIC, ...]}; pairs the
  launch cannot take are simply not filed (the caller runs MODE_WIND for them).

  Nothing here waits for the GPU: results stay on the device (tiny fp64
  tensors), `.values` of the returned DataArrays is where the copy (and the
  sync) happens, so a caller streaming chunks keeps the queue full."""
  device = engine.require_gpu()
  regions, _ = _region_set_for(region)
  prepped = [_prepare_inputs(geo, arrays, tables, device)
             for geo, arrays, tables in entries]
  groups: dict = {}
  for i, ((geo, _, _), (_, _, dtype)) in enumerate(zip(entries, prepped)):
    sig = (geo.layout, geo.latitude.tobytes(), str(geo.latitude.dtype),
           geo.longitude.tobytes(), str(geo.longitude.dtype), dtype)
    groups.setdefault(sig, []).append(i)
  if aux is not None:
    aux = _resident_aux(aux, device)
  out: list = [None] * len(entries)
  for members in groups.values():
    geo0 = entries[members[0]][0]
    n_row = len(geo0.latitude if geo0.layout == plan_lib.LATLON
                else geo0.longitude)
    n_total = sum(entries[i][0].n_outer for i in members)
    pl = plan_lib.cached_plan(geo0.latitude, geo0.longitude, geo0.layout,
                              regions, device, _rows_per_chunk(n_row, n_total))
    lazy = any(isinstance(x, xl.SlabConcat)
               for i in members for x in prepped[i][0])
    rec = program.recorder()
    # wind-vector pairs inside this launch: their slabs go last (u slabs,
    # then v slabs), where the pair kernel expects them
    group_pairs, n_pair, wind = [], 0, None
    if pairs and mode in (_lib.MODE_DET, _lib.MODE_DET_ACC) and (
        engine.pairs_supported(pl, mode, prepped[members[0]][2], skipna)):
      taken: set = set()
      for a, b in pairs:
        if a in members and b in members and a != b and not (
            {a, b} & taken) and (entries[a][0].out_shape ==
                                 entries[b][0].out_shape):
          group_pairs.append((a, b))
          taken |= {a, b}
      if group_pairs:
        members = ([i for i in members if i not in taken] +
                   [a for a, _ in group_pairs] + [b for _, b in group_pairs])
        n_pair = sum(entries[a][0].n_outer for a, _ in group_pairs)
    if rec is not None and rec.probe:
      # program.py's probe pass: no launch, index-valued results
      metrics = rec.fake_metrics(_lib.GENERIC_KQ.get(mode, _lib.NMETRIC),
                                 pl.n_region, n_total, device)
      if n_pair:
        wind = rec.fake_metrics(_lib.NMETRIC, pl.n_region, n_pair, device,
                                extend=True)
    elif len(members) == 1 and not lazy:
      # one variable in one allocation: slab NUMBERS (wb2_stream_partials_ex)
      i = members[0]
      tensors, tables, _ = prepped[i]
      flat, tables = zip(*[_physical_slabs(x, tb, pl.n_row, pl.n_col)
                           for x, tb in zip(tensors, tables)])
      slabs = [None if tb is None else engine.upload_table(tb, device)
               for tb in tables]
      metrics, _ = engine.stream_reduce(pl, mode, flat, slabs, n_total, skipna,
                                        aux=aux, scalar=scalar)
    else:
      n_in = len(prepped[members[0]][0])
      addr = np.empty((n_in, n_total), dtype=np.int64)
      keep, off = [], 0
      for i in members:
        tensors, tables, _ = prepped[i]
        n = entries[i][0].n_outer
        for j, (x, tb) in enumerate(zip(tensors, tables)):
          addr[j, off:off + n], alive = _slab_addresses(x, tb, pl.n_row,
                                                        pl.n_col, n)
          keep.append(alive)
        off += n
      aligned = not (addr % 16).any()
      dev_addr = engine.upload_table(addr, device)
      if n_pair:
        metrics, wind = engine.stream_reduce_pairs(
            pl, mode, prepped[members[0]][2], list(dev_addr), aligned, n_total,
            n_pair, skipna)
      else:
        metrics, _ = engine.stream_reduce_addr(
            pl, mode, prepped[members[0]][2], list(dev_addr), aligned, n_total,
            skipna, aux=aux, scalar=scalar)
      del keep  # the launch is enqueued: the allocator orders any reuse after it
    if rec is not None and not rec.probe:
      rec.record(plan=pl, mode=mode, dtype=prepped[members[0]][2],
                 n_total=n_total, skipna=bool(skipna), aux=aux, scalar=scalar,
                 members=[(entries[i][0], list(entries[i][1]),
                           list(prepped[i][1]), list(prepped[i][0]))
                          for i in members], metrics=metrics, n_pair=n_pair,
                 wind_metrics=wind)
    if n_pair and wind_out is not None:
      wcasts = {wind.dtype: wind}

      def wview(dtype, off, n, shape, wcasts=wcasts, wind=wind):
        if dtype not in wcasts:
          wcasts[dtype] = wind.to(dtype)
        return wcasts[dtype][:, :, off:off + n].reshape(shape)
      woff = 0
      for a, b in group_pairs:
        geo = entries[a][0]
        n = geo.n_outer
        shape = (wind.shape[0], pl.n_region) + geo.out_shape
        wind_out[(a, b)] = _ByRegion(
            wview(wind.dtype, woff, n, shape), pl.region_names,
            lambda dtype, off=woff, n=n, shape=shape, view=wview: view(
                dtype, off, n, shape))
        woff += n
    casts = {metrics.dtype: metrics}

    def view(dtype, off, n, shape, casts=casts, metrics=metrics):
      if dtype not in casts:
        casts[dtype] = metrics.to(dtype)
      return casts[dtype][:, :, off:off + n].reshape(shape)
    off = 0
    for i in members:
      geo = entries[i][0]
      n = geo.n_outer
      shape = (metrics.shape[0], pl.n_region) + geo.out_shape
      out[i] = _ByRegion(
          view(metrics.dtype, off, n, shape), pl.region_names,
          lambda dtype, off=off, n=n, shape=shape, view=view: view(
              dtype, off, n, shape))
      off += n
  return out


def _run_pass(mode, geo, arrays, tables, region, skipna, aux=None, scalar=0.0):
  """Uploads (if needed), launches, returns {region_key: metrics[NMETRIC, ...]}."""
  _, rkey = _region_set_for(region)
  by_region, = _run_group(mode, [(geo, arrays, tables)], region, skipna, aux,
                          scalar)
  return by_region, rkey


def _result_key(kind, arrays, region_key_obj, skipna, datasets=()):
  """Key of a fused pass result: what was read (stamps of the arrays), how it
  was laid out and labelled (dims + coordinate values of `datasets`), for which
  regions, with which NaN rule."""
  sig = _region_set_sig(region_key_obj)[2]
  return (kind, tuple(_stamp(a) for a in arrays), _coord_sig(*datasets), sig,
          bool(skipna))


def _fuse_variables() -> bool:
  """Inside a chunk scope the first variable asked for brings every variable
  of the chunk along (WB2HIP_FUSE_VARIABLES=0: one launch per variable, the
  behaviour of rounds 1-3, for A/B runs)."""
  return _ANNOUNCED.depth > 0 and os.environ.get(
      'WB2HIP_FUSE_VARIABLES', '1') != '0'


def _det_plan(forecast, truth, name, climatology):
  """Host part of one variable's deterministic pass (label work only):
  (geo, arrays, tables, mode, pins, climatology stamp)."""
  fvar, tvar = forecast[name], truth[name]
  cvar, announced = None, False
  if climatology is not None:
    cvar = _get_climatology_chunk(climatology, truth)[name]
  elif _ANNOUNCED.climatology:
    # an ACC over this chunk is coming (fused_climatology): read its
    # climatology now, the ACC call then becomes a cache hit
    try:
      climatology = xl.as_dataset(_ANNOUNCED.climatology[-1])
      cvar = _get_climatology_chunk(climatology, truth)[name]
      announced = True
    except (KeyError, ValueError):
      climatology, cvar = None, None
  pins = [fvar.data, tvar.data]
  geo, prepared = _geometry(forecast, fvar, [tvar])
  tables = [_slab_table(geo.out_dims, geo.out_shape, p[1], p[0].shape[:-2])
            for p in prepared]
  mode = _lib.MODE_DET
  if cvar is not None:
    try:
      crest = tuple(d for d in cvar.dims if d not in _SPATIAL)
      cdata, _, _ = _spatial_last(cvar, geo.layout)
      ctable = _climatology_slabs(climatology, cvar, forecast, geo, crest)
      rec = program.recorder()
      if rec is not None and not rec.probe:
        # how another chunk of this structure gets ITS table (program.py);
        # variables with the same climatology dims share one table per chunk
        sizes = tuple(cvar.sizes[d] for d in crest)

        def recompute(other, memo, climatology=climatology, cvar=cvar,
                      geo=geo, crest=crest, sizes=sizes):
          key = (id(climatology), crest, sizes, geo.out_dims, geo.out_shape)
          if key not in memo:
            memo[key] = _climatology_slabs_by_content(climatology, cvar, other,
                                                      geo, crest)
          return memo[key]
        rec.note_table(ctable, recompute, _climatology_gather(
            climatology, cvar, forecast, geo, crest))
    except (KeyError, ValueError):
      if not announced:
        raise
      cvar = None  # the ACC call itself will report what is wrong
  if cvar is not None:
    pins.append(cvar.data)
    prepared.append((cdata, crest))
    tables.append(ctable)
    mode = _lib.MODE_DET_ACC
  return (geo, [p[0] for p in prepared], tables, mode, tuple(pins),
          None if cvar is None else _stamp(cvar.data))


def _det_key(forecast, truth, name, region, skipna):
  st = _ANNOUNCED
  if st.depth == 0:
    return _result_key('det', [forecast[name].data, truth[name].data], region,
                       skipna, (forecast, truth))
  # inside a chunk scope the key of (datasets, variable, region, announcement)
  # is computed once: a loop asks for it once per metric and variable
  announced = st.regions[-1] if st.regions else None
  memo_key = ('det_key', id(forecast), id(truth), name, id(region),
              id(announced), bool(skipna))
  hit = st.memo.get(memo_key)
  if hit is not None and hit[0] is forecast and hit[1] is truth and (
      hit[2] is region and hit[3] is announced):
    return hit[4]
  key = _result_key('det', [forecast[name].data, truth[name].data], region,
                    skipna, (forecast, truth))
  st.memo[memo_key] = (forecast, truth, region, announced, key)
  return key


def _det_pass(forecast, truth, name, region, skipna, climatology=None):
  """All five deterministic metrics of one variable, for the active regions.

  Inside a chunk scope a miss runs the pass for EVERY variable the two
  datasets share (the reference's Dataset arithmetic does the same, one
  variable after the other): one launch per (grid, dtype, mode) instead of one
  per variable, the other variables' calls are cache hits."""
  key = _det_key(forecast, truth, name, region, skipna)
  hit = _RESULTS.get(key)
  # A cached ACC pass also answers MSE/RMSE/MAE/Bias queries.
  if hit is not None and (climatology is None or hit['clim'] == _stamp(
      _get_climatology_chunk(climatology, truth)[name].data)):
    return hit['geo'], hit['by_region']
  plans = {name: _det_plan(forecast, truth, name, climatology)}
  if _fuse_variables():
    for other in _common_vars(forecast, truth):
      if other in plans:
        continue
      try:
        if _RESULTS.get(_det_key(forecast, truth, other, region, skipna)):
          continue  # an earlier (narrower) pass already answered it
        plans[other] = _det_plan(forecast, truth, other, climatology)
      except _INPUT_ERRORS:  # reported when that variable is asked for
        continue
  by_mode: dict = {}
  for n, pl in plans.items():
    by_mode.setdefault(pl[3], []).append(n)
  for mode, names in by_mode.items():
    # the announced wind-vector pairs whose u and v are both in this launch:
    # their numbers come from the same read (metrics.py:283-301)
    wind_plans, wind_out = {}, {}
    if _fuse_variables() and _ANNOUNCED.wind:
      for u, v in _ANNOUNCED.wind[-1]:
        if u in names and v in names and u != v:
          try:
            if _RESULTS.get(_wind_key(forecast, truth, u, v, region, skipna)):
              continue
            wplan = _wind_plan(forecast, truth, u, v)
          except _INPUT_ERRORS:
            continue
          if all(plans[k][0].layout == wplan[0].layout and
                 plans[k][0].out_dims == wplan[0].out_dims and
                 plans[k][0].out_shape == wplan[0].out_shape for k in (u, v)):
            wind_plans[(names.index(u), names.index(v))] = (u, v, wplan)
    try:
      results = _run_group(mode, [plans[n][:3] for n in names], region, skipna,
                           pairs=list(wind_plans), wind_out=wind_out)
    except _INPUT_ERRORS as e:
      if len(names) == 1 and names[0] == name:
        raise
      logging.getLogger(__name__).info(
          'fused launch over %s fell back to %r alone: %s: %s', names, name,
          type(e).__name__, e)
      # a companion variable cannot be read (foreign dtype, bad grid): the
      # requested one goes alone, the others report when they are asked for
      names = [n for n in names if n == name]
      if not names:
        continue
      results = _run_group(mode, [plans[name][:3]], region, skipna)
      wind_out = {}
    for n, by_region in zip(names, results):
      geo, _, _, _, pins, clim = plans[n]
      _RESULTS.put(_det_key(forecast, truth, n, region, skipna), pins,
                   {'geo': geo, 'by_region': by_region, 'clim': clim})
    for idx, by_region in wind_out.items():
      u, v, wplan = wind_plans[idx]
      _RESULTS.put(_wind_key(forecast, truth, u, v, region, skipna), wplan[3],
                   (wplan[0], by_region))
  hit = _RESULTS.get(key)
  return hit['geo'], hit['by_region']


def _result_coords(forecast: xl.Dataset, out_dims) -> dict:
  coords = {}
  for k, c in forecast.coords.items():
    if k in _SPATIAL:
      continue
    if isinstance(c, xl.DataArray):
      if all(d in out_dims for d in c.dims):
        coords[k] = c
    elif k in out_dims:
      coords[k] = c
  return coords


def _stack(arrays):
  """np.stack / torch.stack: fused results are device tensors."""
  if isinstance(arrays[0], torch.Tensor):
    return torch.stack(list(arrays))
  return np.stack([np.asarray(a, dtype=np.float64) for a in arrays])


def _nansum_leading(values):
  """xarray's `.sum(dim)` with its default skipna=None: NaNs are skipped for
  float data, an all-NaN slice sums to 0.  This is what the reference's
  `result.sum("quantile")` does to the per-threshold terms of the ranked
  probability scores (metrics.py:1158, 1868, 1891): a NaN term -- an empty
  region, a NaN input without skipna -- drops out of the sum."""
  if isinstance(values, torch.Tensor):
    return torch.nansum(values, 0)
  return np.nansum(values, 0)


def _transpose(values, axes):
  if isinstance(values, torch.Tensor):
    return values.permute(*axes)
  return np.transpose(values, axes)


def _pick(by_region: dict, region, index, regions: t.Optional[dict],
          dtype=None):
  """The requested region's row of a fused result -- or, for the all-regions
  fast path, every announced region stacked along a leading `region` dim --
  in the result dtype `dtype` when the pass can give it (a view of one shared
  conversion, _ByRegion.as_dtype; `_assemble` converts whatever is left)."""
  if dtype is not None and isinstance(by_region, _ByRegion):
    by_region = by_region.as_dtype(dtype)
  if regions is None:
    _, rkey = _region_set_for(region)
    return (), by_region[rkey][index]
  whole = by_region.get(_ALL)
  if whole is not None and whole[1] == list(regions):
    return ('region',), whole[0][index]  # already [region, ...]: a view
  return ('region',), _stack([by_region[k][index] for k in regions])


@contextlib.contextmanager
def _all_regions(regions: t.Optional[dict]):
  """An announcement of exactly `regions` (no-op for a single-region call)."""
  if regions is None:
    yield
  else:
    with fused_regions(regions):
      yield


def _assemble(forecast, per_var: dict,
              regions: t.Optional[dict] = None) -> xl.Dataset:
  """{var: (out_dims, array)} -> Dataset without spatial dims."""
  out = xl.Dataset()
  if regions is not None:
    out.coords['region'] = np.array(list(regions), dtype=object)
  for name, entry in per_var.items():
    out.coords.update(_result_coords(forecast, entry[0]))
  for name, entry in per_var.items():
    dims, arr = entry[0], entry[1]
    # third entry: the dtype the reference's weighted mean returns for this
    # variable (_reference_result_dtype); float64 when not given
    dtype = np.dtype(entry[2]) if len(entry) > 2 else np.dtype(np.float64)
    if isinstance(arr, torch.Tensor):
      data = arr.to(torch.float32 if dtype == np.float32 else torch.float64)
    else:
      data = np.array(arr, dtype=dtype)
    out.data_vars[name] = xl.DataArray(data, dims, out.coords, name)
  return out


def _np_dtype(x) -> np.dtype:
  """NumPy dtype of a numpy / torch / SlabGather array."""
  dt = x.dtype
  if isinstance(dt, torch.dtype):
    return np.dtype(str(dt).replace('torch.', ''))
  return np.dtype(dt)


def _region_weight_dtype(w: np.dtype, region) -> np.dtype:
  """dtype of the weights after `region.apply` (regions.py:56-158): slices keep
  it, the extra-tropical mask and a thresholded land-sea mask are
  `.astype(float)` fields (float64), an unthresholded land-sea mask brings its
  own dtype, a combined region applies its parts in turn."""
  if region is None:
    return w
  kind = type(region).__name__
  if kind == 'SliceRegion':
    return w
  if kind == 'LandRegion':
    if getattr(region, 'threshold', None) is not None:
      return np.promote_types(w, np.float64)
    mask = getattr(region.land_sea_mask, 'data', region.land_sea_mask)
    return np.promote_types(w, _np_dtype(mask))
  if kind == 'CombinedRegion':
    for r in region.regions:
      w = _region_weight_dtype(w, r)
    return w
  return np.promote_types(w, np.float64)  # ExtraTropicalRegion, foreign regions


def _reference_result_dtype(forecast: xl.Dataset, data_dtypes, region,
                            regions: t.Optional[dict]) -> np.dtype:
  """The dtype `_spatial_average` returns in the reference (metrics.py:141-163):
  xarray's weighted mean is `dot(data, weights) / dot(notnull, weights)`, both
  in promote(data dtype, weights dtype), and the weights inherit the dtype of
  the LATITUDE COORDINATE (metrics.py:41, 57) unless a mask region multiplies a
  float64 field in.  So float32 data on a grid with float32 coordinates (0.25
  degree ERA5) comes back float32 for slice regions and float64 for mask
  regions; with float64 coordinates everything is float64.  A loop over several
  regions concatenates along `region` (evaluation.py:430): NumPy promotion over
  the regions' dtypes.  The VALUES here are the float64 sums of the fused pass
  rounded once to that dtype (the reference's float32 einsum carries ~1e-5 of
  summation noise at 10^6 points; see profiles/NOTES.md 4)."""
  st = _ANNOUNCED
  data_dtypes = tuple(data_dtypes)
  memo_key = None
  if st.depth > 0:
    memo_key = ('result_dtype', id(forecast), data_dtypes, id(region),
                id(regions))
    hit = st.memo.get(memo_key)
    if hit is not None and hit[0] is forecast and hit[1] is region and (
        hit[2] is regions):
      return hit[3]
  out = _reference_result_dtype_uncached(forecast, data_dtypes, region, regions)
  if memo_key is not None:
    st.memo[memo_key] = (forecast, region, regions, out)
  return out


def _reference_result_dtype_uncached(forecast, data_dtypes, region, regions):
  w = np.sin(np.deg2rad(_coord_values(forecast, 'latitude')[:1])).dtype
  data = np.dtype(np.float32)
  first = True
  for dt in data_dtypes:
    data = dt if first else np.promote_types(data, dt)
    first = False
  if data.kind not in 'f':
    data = np.promote_types(data, np.float32)
  out = None
  for r in (regions.values() if regions is not None else [region]):
    dt = np.promote_types(data, _region_weight_dtype(w, r))
    out = dt if out is None else np.promote_types(out, dt)
  if out not in (np.dtype(np.float32), np.dtype(np.float64)):
    out = np.dtype(np.float64)
  return out


def _common_vars(forecast, truth):
  return [k for k in forecast.keys() if k in truth]


# ---------------------------------------------------------------------------
# Metric classes (metrics.py:84-414)
# ---------------------------------------------------------------------------
def get_lat_weights(ds) -> xl.DataArray:
  """Latitude/area weights of a dataset's latitude coordinate, normalised to
  mean 1 (metrics.py:55-60; same signature: a Dataset in, a DataArray over
  `latitude` out).  Host work, bit-identical NumPy ops (`plan.get_lat_weights`)."""
  given = ds
  ds = xl.as_dataset(ds)
  lat = np.asarray(ds.coords['latitude'])
  out = xl.DataArray(plan_lib.get_lat_weights(lat), ('latitude',),
                     {'latitude': lat}, 'latitude')
  return xl.like_input(out, given)


def _returns_like_input(fn):
  """xarray in -> xarray out at the public entry points (xarray_lite.like_input);
  calls between metrics pass lite Datasets and skip the conversion."""
  @functools.wraps(fn)
  def wrapper(self, forecast, truth, *args, **kwargs):
    result = fn(self, forecast, truth, *args, **kwargs)
    if _ANNOUNCED.depth == 0:  # a bare call outside any chunk scope
      engine.publish_thread_stream()
    return xl.like_input(result, forecast, truth)
  wrapper._wb2_like_input = True
  return wrapper


@dataclasses.dataclass
class Metric:
  """Base class for metrics (metrics.py:84-138)."""

  def __init_subclass__(cls, **kwargs):
    super().__init_subclass__(**kwargs)
    for name in ('compute_chunk', 'compute', 'compute_chunk_regions',
                 'compute_regions'):
      fn = cls.__dict__.get(name)
      if callable(fn) and not getattr(fn, '_wb2_like_input', False):
        setattr(cls, name, _returns_like_input(fn))

  def compute_chunk(self, forecast, truth, region: t.Optional[Region] = None,
                    skipna: bool = False) -> xl.Dataset:
    raise NotImplementedError

  def compute(self, forecast, truth, region: t.Optional[Region] = None,
              skipna: bool = False) -> xl.Dataset:
    """Evaluate this metric on datasets with full temporal coverages."""
    forecast, truth = xl.as_dataset(forecast), xl.as_dataset(truth)
    if 'time' in forecast.dims:
      avg_dim = 'time'
    elif 'init_time' in forecast.dims:
      avg_dim = 'init_time'
    else:
      raise ValueError(
          f'Forecast has neither valid_time or init_time dimension {forecast}')
    return self.compute_chunk(forecast, truth, region=region,
                              skipna=skipna).mean(avg_dim, skipna=skipna)

  # All regions of a loop at once, with a leading `region` dim: what
  # evaluation.py:416-430 builds with one call + expand_dims + concat per
  # region.  The generic versions do exactly that; metrics whose fused pass
  # already holds every region override them with a single stack.
  def compute_chunk_regions(self, forecast, truth, regions: dict,
                            skipna: bool = False) -> xl.Dataset:
    return self._fan_out(self.compute_chunk, forecast, truth, regions, skipna)

  def compute_regions(self, forecast, truth, regions: dict,
                      skipna: bool = False) -> xl.Dataset:
    return self._fan_out(self.compute, forecast, truth, regions, skipna)

  def _fan_out(self, fn, forecast, truth, regions, skipna):
    forecast, truth = xl.as_dataset(forecast), xl.as_dataset(truth)
    with fused_regions(regions):
      parts = [xl.as_dataset(fn(forecast=forecast, truth=truth, region=region,
                                skipna=skipna)).expand_dims({'region': [name]})
               for name, region in regions.items()]
    return xl.concat(parts, 'region')

  def _mean_regions(self, forecast, truth, regions, skipna):
    """Metric.compute on the stacked result (same checks, same mean)."""
    forecast, truth = xl.as_dataset(forecast), xl.as_dataset(truth)
    if 'time' in forecast.dims:
      avg_dim = 'time'
    elif 'init_time' in forecast.dims:
      avg_dim = 'init_time'
    else:
      raise ValueError(
          f'Forecast has neither valid_time or init_time dimension {forecast}')
    return self.compute_chunk_regions(forecast, truth, regions,
                                      skipna).mean(avg_dim, skipna=skipna)


for _name in ('compute', 'compute_chunk_regions', 'compute_regions'):
  setattr(Metric, _name, _returns_like_input(Metric.__dict__[_name]))
del _name


class _DetMetric(Metric):
  _index: int = -1
  # reads concatenated chunks (xarray_lite.SlabConcat) through address tables,
  # without materialising them: evaluate_chunks may batch chunks for it
  _reads_slabs_in_place = True

  def _scalar(self, forecast, truth, region, skipna,
              regions: t.Optional[dict] = None) -> xl.Dataset:
    forecast, truth = _inputs(forecast, truth)
    per_var = {}
    with _all_regions(regions):
      for name in _common_vars(forecast, truth):
        geo, by_region = _fused(
            lambda r, name=name: _det_pass(forecast, truth, name, r, skipna),
            region, regions)
        dtype = _reference_result_dtype(
            forecast, [_np_dtype(forecast[name].data),
                       _np_dtype(truth[name].data)], region, regions)
        lead, values = _pick(by_region, region, self._index, regions, dtype)
        per_var[name] = (lead + geo.out_dims, values, dtype)
    return _assemble(forecast, per_var, regions)

  def compute_chunk_regions(self, forecast, truth, regions, skipna=False):
    return self._scalar(forecast, truth, None, skipna, regions)

  def compute_regions(self, forecast, truth, regions, skipna=False):
    return self._mean_regions(forecast, truth, regions, skipna)


def _wind_plan(forecast, truth, u_name, v_name):
  fu, fv, tu, tv = (forecast[u_name], forecast[v_name], truth[u_name],
                    truth[v_name])
  pins = (fu.data, tu.data, fv.data, tv.data)
  geo, prepared = _geometry(forecast, fu, [tu, fv, tv])
  tables = [_slab_table(geo.out_dims, geo.out_shape, p[1], p[0].shape[:-2])
            for p in prepared]
  return geo, [p[0] for p in prepared], tables, pins


def _wind_key(forecast, truth, u_name, v_name, region, skipna):
  pins = [forecast[u_name].data, truth[u_name].data, forecast[v_name].data,
          truth[v_name].data]
  return _result_key('wind', pins, region, skipna, (forecast, truth))


def _wind_pass(forecast, truth, u_name, v_name, region, skipna):
  """Wind-vector MSE / RMSE of one (u, v) pair; inside a chunk scope every
  announced pair (`fused_wind_vectors`) is read by the same launch."""
  key = _wind_key(forecast, truth, u_name, v_name, region, skipna)
  hit = _RESULTS.get(key)
  if hit is not None:
    return hit
  pairs = [(u_name, v_name)]
  if _fuse_variables() and _ANNOUNCED.wind:
    pairs += [p for p in _ANNOUNCED.wind[-1] if p not in pairs
              and all(k in forecast and k in truth for k in p)]
  plans = {}
  for pair in pairs:
    try:
      if pair != pairs[0] and _RESULTS.get(
          _wind_key(forecast, truth, *pair, region, skipna)):
        continue
      plans[pair] = _wind_plan(forecast, truth, *pair)
    except _INPUT_ERRORS:
      if pair == pairs[0]:
        raise
  names = list(plans)
  try:
    results = _run_group(_lib.MODE_WIND, [plans[p][:3] for p in names], region,
                         skipna)
  except _INPUT_ERRORS as e:
    if len(names) == 1:
      raise
    logging.getLogger(__name__).info(
        'fused wind-vector launch over %s fell back to %s alone: %s: %s',
        names, names[0], type(e).__name__, e)
    names = names[:1]
    results = _run_group(_lib.MODE_WIND, [plans[names[0]][:3]], region, skipna)
  for pair, by_region in zip(names, results):
    _RESULTS.put(_wind_key(forecast, truth, *pair, region, skipna),
                 plans[pair][3], (plans[pair][0], by_region))
  return _RESULTS.get(key)


@dataclasses.dataclass
class WindVectorMSE(Metric):
  """Wind vector mean square error (metrics.py:175-202)."""

  u_name: str
  v_name: str
  vector_name: str
  _index = _lib.METRIC_INDEX['mse']
  _reads_slabs_in_place = True

  def compute_chunk(self, forecast, truth, region=None, skipna=False,
                    regions: t.Optional[dict] = None):
    forecast, truth = _inputs(forecast, truth)
    with _all_regions(regions):
      geo, by_region = _fused(
          lambda r: _wind_pass(forecast, truth, self.u_name, self.v_name, r,
                               skipna), region, regions)
    dtype = _reference_result_dtype(
        forecast, [_np_dtype(ds[k].data) for ds in (forecast, truth)
                   for k in (self.u_name, self.v_name)], region, regions)
    lead, values = _pick(by_region, region, self._index, regions, dtype)
    return _assemble(forecast, {self.vector_name: (lead + geo.out_dims,
                                                   values, dtype)}, regions)

  def compute_chunk_regions(self, forecast, truth, regions, skipna=False):
    return self.compute_chunk(forecast, truth, None, skipna, regions)

  def compute_regions(self, forecast, truth, regions, skipna=False):
    return self._mean_regions(forecast, truth, regions, skipna)


@dataclasses.dataclass
class WindVectorRMSESqrtBeforeTimeAvg(WindVectorMSE):
  """Wind vector RMSE, sqrt before time averaging (metrics.py:205-233)."""

  _index = _lib.METRIC_INDEX['rmse']


@dataclasses.dataclass
class RMSESqrtBeforeTimeAvg(_DetMetric):
  """Root mean squared error, sqrt before time averaging (metrics.py:236-269)."""

  wind_vector_rmse: t.Optional[list] = None
  _index = _lib.METRIC_INDEX['rmse']

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    # converted once, here: the wind-vector metrics below are then called with
    # lite Datasets and answer with lite Datasets (xarray callers get their
    # xarray result from the wrapper around THIS method)
    forecast, truth = xl.as_dataset(forecast), xl.as_dataset(truth)
    results = self._scalar(forecast, truth, region, skipna)
    if self.wind_vector_rmse is not None:
      for wv in self.wind_vector_rmse:
        results[wv.vector_name] = wv.compute_chunk(
            forecast, truth, region=region, skipna=skipna)[wv.vector_name]
    return results

  def compute_chunk_regions(self, forecast, truth, regions, skipna=False):
    forecast, truth = xl.as_dataset(forecast), xl.as_dataset(truth)
    results = self._scalar(forecast, truth, None, skipna, regions)
    if self.wind_vector_rmse is not None:
      for wv in self.wind_vector_rmse:
        results[wv.vector_name] = wv.compute_chunk_regions(
            forecast, truth, regions, skipna)[wv.vector_name]
    return results


@dataclasses.dataclass
class MSE(_DetMetric):
  """Mean squared error (metrics.py:272-301)."""

  wind_vector_mse: t.Optional[list] = None
  _index = _lib.METRIC_INDEX['mse']

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    # converted once, here: the wind-vector metrics below are then called with
    # lite Datasets and answer with lite Datasets (xarray callers get their
    # xarray result from the wrapper around THIS method)
    forecast, truth = xl.as_dataset(forecast), xl.as_dataset(truth)
    results = self._scalar(forecast, truth, region, skipna)
    if self.wind_vector_mse is not None:
      for wv in self.wind_vector_mse:
        results[wv.vector_name] = wv.compute_chunk(
            forecast, truth, region=region, skipna=skipna)[wv.vector_name]
    return results

  def compute_chunk_regions(self, forecast, truth, regions, skipna=False):
    forecast, truth = xl.as_dataset(forecast), xl.as_dataset(truth)
    results = self._scalar(forecast, truth, None, skipna, regions)
    if self.wind_vector_mse is not None:
      for wv in self.wind_vector_mse:
        results[wv.vector_name] = wv.compute_chunk_regions(
            forecast, truth, regions, skipna)[wv.vector_name]
    return results


@dataclasses.dataclass
class MAE(_DetMetric):
  """Mean absolute error (metrics.py:319-330)."""

  _index = _lib.METRIC_INDEX['mae']

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    return self._scalar(forecast, truth, region, skipna)


@dataclasses.dataclass
class Bias(_DetMetric):
  """Bias (metrics.py:348-359)."""

  _index = _lib.METRIC_INDEX['bias']

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    return self._scalar(forecast, truth, region, skipna)


@dataclasses.dataclass
class ACC(Metric):
  """Anomaly correlation coefficient (metrics.py:377-414).

  Attribute:
    climatology: Climatology for computing anomalies (Dataset with dims
      [hour,] dayofyear[, level], latitude, longitude).
  """

  climatology: t.Any = None
  _reads_slabs_in_place = True

  def compute_chunk(self, forecast, truth, region=None, skipna=False,
                    regions: t.Optional[dict] = None):
    forecast, truth = _inputs(forecast, truth)
    climatology = xl.as_dataset(self.climatology)
    per_var = {}
    _get_climatology_chunk(climatology, truth)  # KeyError like the reference
    with _all_regions(regions):
      for name in _common_vars(forecast, truth):
        geo, by_region = _fused(
            lambda r, name=name: _det_pass(forecast, truth, name, r, skipna,
                                           climatology), region, regions)
        cvar = _get_climatology_chunk(climatology, truth)[name]
        dtype = _reference_result_dtype(
            forecast, [_np_dtype(forecast[name].data),
                       _np_dtype(truth[name].data), _np_dtype(cvar.data)],
            region, regions)
        lead, values = _pick(by_region, region, _lib.METRIC_INDEX['acc'],
                             regions, dtype)
        per_var[name] = (lead + geo.out_dims, values, dtype)
    return _assemble(forecast, per_var, regions)

  def compute_chunk_regions(self, forecast, truth, regions, skipna=False):
    return self.compute_chunk(forecast, truth, None, skipna, regions)

  def compute_regions(self, forecast, truth, regions, skipna=False):
    return self._mean_regions(forecast, truth, regions, skipna)


# ---------------------------------------------------------------------------
# Ensemble metrics (metrics.py:532-846, 1161-1363): one fused pass per chunk
# ---------------------------------------------------------------------------
def _get_n_ensemble(ds: xl.Dataset, ensemble_dim: str,
                    expect_n_ensemble_at_least: int = 1) -> int:
  """metrics.py:568-582 (same messages)."""
  if ensemble_dim not in ds.dims:
    raise ValueError(f'{ensemble_dim=} not found in {ds.dims=}')
  n_ensemble = ds.sizes[ensemble_dim]
  if n_ensemble < expect_n_ensemble_at_least:
    raise ValueError(f'{n_ensemble=} is less than expected size of '
                     f'{expect_n_ensemble_at_least}')
  return n_ensemble


def _torch_dtype_of(data) -> t.Optional[torch.dtype]:
  """torch dtype of a tensor / NumPy array / lazy container; None if torch has
  no such dtype."""
  dtype = getattr(data, 'dtype', None)
  if isinstance(dtype, torch.dtype):
    return dtype
  try:
    return torch.from_numpy(np.empty(0, dtype=np.dtype(dtype))).dtype
  except (TypeError, ValueError):
    return None


def _ens_layout(forecast, fvar, tvar, ensemble_dim, allow_gather=False):
  """Device tensors + slab tables of one ensemble variable: member m of outer
  index o is slab  m * stride_member + ens_table[o]  of the forecast array --
  or, for a gathered forecast when the caller can take it (`allow_gather`: K3),
  the slab at address member_ptrs[o, m] (last entry of the result, else None)."""
  if ensemble_dim not in fvar.dims:
    raise ValueError(f'{ensemble_dim=} not found in {fvar.dims=}')
  fdata, frest, layout = _spatial_last(fvar, None)
  tdata, trest, _ = _spatial_last(tvar, layout)
  if ensemble_dim in trest:
    raise ValueError(f'truth must not have the ensemble dim {ensemble_dim!r}')
  fsizes = dict(zip(frest, fdata.shape[:-2]))
  n_member = fsizes[ensemble_dim]
  out_dims = [d for d in frest if d != ensemble_dim]
  sizes = {d: fsizes[d] for d in out_dims}
  for d, n in zip(trest, tdata.shape[:-2]):
    if d not in out_dims:
      out_dims.append(d)
      sizes[d] = n
  out_dims = tuple(out_dims)
  out_shape = tuple(sizes[d] for d in out_dims)
  geo = _Geometry(layout, out_dims, out_shape,
                  _coord_values(forecast, 'latitude'),
                  _coord_values(forecast, 'longitude'))
  # member m, outer index o  ->  slab  m * stride_member + ens_table[o]
  stride, strides = 1, {}
  for d in reversed(frest):
    strides[d] = stride
    stride *= fsizes[d]
  strided = False
  if isinstance(fdata, torch.Tensor) and not fdata.is_contiguous():
    # a VIEW with intact 2-D slabs (an (init_time=1, lead_time=1) chunk sliced
    # out of a resident forecast): addressed through its own strides, no copy
    # -- but only if the kernel reads the view AS IT IS: a cast to the common
    # dtype or an upload below returns a compact copy, and the view's strides
    # would then address memory the copy does not have
    se = int(fdata.shape[-2]) * int(fdata.shape[-1])
    st = fdata.stride()
    tdtype = _torch_dtype_of(tdata)
    common = None if tdtype is None else torch.promote_types(fdata.dtype,
                                                             tdtype)
    as_is = (fdata.device.type == 'cuda' and common == fdata.dtype and
             common in (torch.float32, torch.float64))
    if as_is and fdata.dim() >= 2 and st[-1] == 1 and (
        st[-2] == fdata.shape[-1]) and all(
        x >= 0 and x % se == 0 for x in st[:-2]) and se > 0:
      strides = {d: x // se for d, x in zip(frest, st[:-2])}
      strided = True
    else:
      fdata = fdata.contiguous()
  ens_table = np.zeros(out_shape, dtype=np.int64)
  for d in frest:
    if d == ensemble_dim:
      continue
    shape = [1] * len(out_shape)
    shape[out_dims.index(d)] = fsizes[d]
    ens_table = ens_table + (np.arange(fsizes[d], dtype=np.int64)
                             * strides[d]).reshape(shape)
  ens_table = np.ascontiguousarray(ens_table).ravel()
  identity = (not strided) and np.array_equal(ens_table,
                                              np.arange(ens_table.size))
  truth_table = _slab_table(out_dims, out_shape, trest, tdata.shape[:-2])

  device = engine.require_gpu()
  tten = _to_device(tdata, device)
  to_dev = lambda tb: None if tb is None else engine.upload_table(tb, device)
  member_ptrs = None
  if isinstance(fdata, xl.SlabGather) and not allow_gather:
    ften = _to_device(fdata, device)
  elif isinstance(fdata, xl.SlabGather):
    # forecast := probabilistic climatology (evaluation.py:458-470): every
    # member slab is read where it lives; K3 gets one address per (outer,
    # member), holes point at a resident NaN slab.  Anything the gather kernels
    # do not cover (dtype conversion, more members than the register sort
    # takes) is materialised below, as before.
    base, index = fdata.base, fdata.index
    if isinstance(base, np.ndarray):
      base, index = fdata.compact_host()
    dev_base = _to_device(base, device)
    if (dev_base.dtype == tten.dtype and dev_base.is_contiguous() and
        n_member <= engine.GATHER_MAX_MEMBERS.get(dev_base.dtype, 0)):
      ax = frest.index(ensemble_dim)
      moved = np.moveaxis(index, ax, -1)  # [frest without the ensemble dim, M]
      n_extra = len(out_dims) - (len(frest) - 1)
      moved = moved.reshape(moved.shape[:-1] + (1,) * n_extra + (n_member,))
      table = np.broadcast_to(moved, out_shape + (n_member,))
      slab_elems = int(dev_base.shape[-2]) * int(dev_base.shape[-1])
      ptrs = engine.gather_pointers(dev_base, table, slab_elems)
      member_ptrs = to_dev(np.ascontiguousarray(ptrs).ravel())
      ften = dev_base
      _check_grid(geo, ften)
      _check_grid(geo, tten)
      return (geo, ften, tten, None, to_dev(truth_table), 0, n_member, device,
              member_ptrs)
    ften = xl.SlabGather(dev_base, index).materialize()
  else:
    ften = _to_device(fdata, device)
  dtype = torch.promote_types(ften.dtype, tten.dtype)
  if dtype not in (torch.float32, torch.float64):
    dtype = torch.float64
  ften = ften if ften.dtype == dtype else ften.to(dtype)
  tten = tten if tten.dtype == dtype else tten.to(dtype)
  _check_grid(geo, ften)
  _check_grid(geo, tten)
  return (geo, ften, tten, None if identity else to_dev(ens_table),
          to_dev(truth_table), strides[ensemble_dim], n_member, device, None)


def _ens_concat_layout(forecast, fvar, tvar, ensemble_dim):
  """The layout of one ensemble variable of a WINDOW of chunks
  (evaluation.concat_chunks: `fvar.data` is an xarray_lite.SlabConcat over the
  chunks' own arrays) for K3 by address: member 0's slab and the truth slab of
  every outer index, one member stride for all of them.  None where the window
  cannot be read where it lies (a cast, host data, members that do not follow
  each other at one stride): the caller materialises it as before."""
  if ensemble_dim not in fvar.dims:
    raise ValueError(f'{ensemble_dim=} not found in {fvar.dims=}')
  fdata, frest, layout = _spatial_last(fvar, None)
  in_hbm = lambda x: isinstance(x, torch.Tensor) and x.device.type == 'cuda'
  if not isinstance(fdata, xl.SlabConcat) or not in_hbm(fdata.bases[0]):
    return None
  tdata, trest, _ = _spatial_last(tvar, layout)
  if ensemble_dim in trest:
    raise ValueError(f'truth must not have the ensemble dim {ensemble_dim!r}')
  dtype = fdata.dtype
  on_device = in_hbm(tdata.bases[0] if isinstance(tdata, xl.SlabConcat)
                     else tdata)
  if dtype not in (torch.float32, torch.float64) or not on_device or (
      _torch_dtype_of(tdata) != dtype):
    return None
  fsizes = dict(zip(frest, fdata.shape[:-2]))
  n_member = fsizes[ensemble_dim]
  out_dims = [d for d in frest if d != ensemble_dim]
  sizes = {d: fsizes[d] for d in out_dims}
  n_own = len(out_dims)
  for d, n in zip(trest, tdata.shape[:-2]):
    if d not in out_dims:
      out_dims.append(d)
      sizes[d] = n
  out_dims = tuple(out_dims)
  out_shape = tuple(sizes[d] for d in out_dims)
  geo = _Geometry(layout, out_dims, out_shape,
                  _coord_values(forecast, 'latitude'),
                  _coord_values(forecast, 'longitude'))
  _check_grid(geo, fdata)
  _check_grid(geo, tdata)
  # slab numbers (in the virtual concatenation of the chunks) [outer..., member]
  index = np.moveaxis(fdata.index, frest.index(ensemble_dim), -1)
  first = index[..., 0]
  stride = 0
  if n_member > 1:
    steps = index[..., 1:] - index[..., :-1]
    stride = int(steps.flat[0]) if steps.size else 0
    home = lambda ix: np.searchsorted(fdata.offsets, ix, side='right') - 1
    if stride < 0 or (steps != stride).any() or (
        home(first) != home(index[..., -1])).any():
      return None
  wide = lambda a: np.ascontiguousarray(np.broadcast_to(
      a.reshape(a.shape + (1,) * (len(out_dims) - n_own)), out_shape)).ravel()
  first = wide(first)
  n_outer = geo.n_outer
  item = fdata.bases[0].element_size()
  step = fdata.slab_shape[0] * fdata.slab_shape[1] * item
  base_of = np.searchsorted(fdata.offsets, first, side='right') - 1
  starts = np.array([b.data_ptr() for b in fdata.bases], dtype=np.int64)
  ens_addr = starts[base_of] + (first - fdata.offsets[base_of]) * step
  truth_table = _slab_table(out_dims, out_shape, trest, tdata.shape[:-2])
  truth_addr, keep = _slab_addresses(tdata, truth_table, fdata.slab_shape[0],
                                     fdata.slab_shape[1], n_outer)
  return dict(geo=geo, dtype=dtype, n_member=n_member,
              member_stride=stride * fdata.slab_shape[0] * fdata.slab_shape[1],
              ens_addr=ens_addr, truth_addr=truth_addr,
              keep=(fdata.bases, keep), ens_first=first,
              truth_table=truth_table, truth_data=tdata)


def _ens_pass_concat(fvar, tvar, lay, region, skipna):
  """_ens_pass over a window read in place: (geo, device, plan, metrics)."""
  geo = lay['geo']
  device = engine.require_gpu()
  regions, _ = _region_set_for(region)
  pl = plan_lib.cached_plan(
      geo.latitude, geo.longitude, geo.layout, regions, device,
      plan_lib.ENSEMBLE_ROWS_PER_CHUNK)
  rec = program.recorder()
  if rec is not None and rec.probe:
    return geo, pl, rec.fake_metrics(_lib.NMETRIC_ENS, pl.n_region, geo.n_outer,
                                     device)
  table = engine.upload_table(
      np.concatenate([lay['ens_addr'], lay['truth_addr']]), device)
  like = lay['keep'][0][0]     # (a tensor of the dtype: the kernel is told
  metrics, _ = engine.ensemble_reduce(   # addresses, not tensors)
      pl, like, lay['member_stride'], lay['n_member'], None, like, None,
      geo.n_outer, skipna, addresses=table)
  if rec is not None:
    rec.record(kind='ens', plan=pl, ens=None, ens_raw=fvar.data,
               truth=None, truth_raw=tvar.data, concat=lay,
               member_stride=lay['member_stride'], n_member=lay['n_member'],
               ens_table=None, truth_table=lay['truth_table'],
               n_outer=geo.n_outer, skipna=bool(skipna), metrics=metrics,
               dtype=lay['dtype'])
  return geo, pl, metrics


def _ens_pass(forecast, truth, name, ensemble_dim, region, skipna,
              want_maps: bool = False):
  """All ensemble metrics of one variable for the active regions (and, with
  `want_maps`, the six pointwise maps as a device tensor)."""
  fvar, tvar = forecast[name], truth[name]
  pins = [fvar.data, tvar.data]
  key = _result_key(('ens', ensemble_dim, want_maps), pins, region, skipna,
                    (forecast, truth))
  hit = _RESULTS.get(key)
  if hit is not None:
    return hit
  lay = None
  if isinstance(fvar.data, xl.SlabConcat) and not want_maps:
    # a window of chunks (evaluate_chunks): read where the chunks lie
    lay = _ens_concat_layout(forecast, fvar, tvar, ensemble_dim)
  if lay is not None:
    geo, pl, metrics = _ens_pass_concat(fvar, tvar, lay, region, skipna)
    dev = metrics.reshape((_lib.NMETRIC_ENS, pl.n_region) + geo.out_shape)
    geo.region_wsum = dict(zip(pl.region_names,
                               (float(w) for w in pl.region_wsum_host)))
    value = (geo, {nm: dev[:, i] for i, nm in enumerate(pl.region_names)},
             lay['n_member'])
    _RESULTS.put(key, tuple(pins), value)
    return value
  (geo, ften, tten, ens_table, truth_table, member_slabs, n_member, device,
   member_ptrs) = _ens_layout(forecast, fvar, tvar, ensemble_dim,
                              allow_gather=True)
  out_shape = geo.out_shape
  regions, _ = _region_set_for(region)
  pl = plan_lib.cached_plan(
      geo.latitude, geo.longitude, geo.layout, regions, device,
      plan_lib.ENSEMBLE_ROWS_PER_CHUNK)
  slab_elems = pl.n_row * pl.n_col
  maps = (torch.empty((6, geo.n_outer, slab_elems), dtype=torch.float64,
                      device=device) if want_maps else None)
  rec = program.recorder()
  plain = maps is None and member_ptrs is None
  if rec is not None and rec.probe and plain:
    # program.py's probe pass: no launch, index-valued results
    metrics = rec.fake_metrics(_lib.NMETRIC_ENS, pl.n_region, geo.n_outer,
                               device)
  else:
    truth3 = tten.reshape(-1, pl.n_row, pl.n_col)
    metrics, _ = engine.ensemble_reduce(
        pl, ften, member_slabs * slab_elems, n_member, ens_table, truth3,
        truth_table, geo.n_outer, skipna, maps=maps,
        member_ptrs=(None if member_ptrs is None
                     else member_ptrs.reshape(geo.n_outer, n_member)))
    if rec is not None and not rec.probe:
      if plain:
        rec.record(kind='ens', plan=pl, ens=ften, ens_raw=fvar.data,
                   truth=truth3, truth_raw=tvar.data,
                   member_stride=member_slabs * slab_elems, n_member=n_member,
                   ens_table=ens_table, truth_table=truth_table,
                   n_outer=geo.n_outer, skipna=bool(skipna), metrics=metrics)
      else:
        rec.record(kind='unsupported')
  dev = metrics.reshape((_lib.NMETRIC_ENS, pl.n_region) + geo.out_shape)
  # total weight of each region: a spatial average of zeros is 0/0 = NaN over
  # an empty region (CRPSSpread with one member, metrics.py:689-693, 783-784)
  geo.region_wsum = dict(zip(pl.region_names,
                             (float(w) for w in pl.region_wsum_host)))
  value = (geo, {nm: dev[:, i] for i, nm in enumerate(pl.region_names)},
           n_member)
  if want_maps:
    value = value + (maps.reshape((6,) + out_shape + (pl.n_row, pl.n_col)),
                     ften.dtype)
  _RESULTS.put(key, tuple(pins), value)
  return value


@dataclasses.dataclass
class EnsembleMetric(Metric):
  """Ensemble metric base class (metrics.py:585-607)."""

  ensemble_dim: str = REALIZATION
  _metric = ''
  _zero_if_single = False  # metrics.py:1196-1204, 1228-1235, 783-784
  # zeros_like(spatial average) is 0 everywhere (:1197-1204); the spatial
  # average OF zeros (CRPSSpread, :689-693) is NaN where a region has no weight
  _zero_is_spatial_average = False
  # xarray puts the dims of the LEFT operand first: metrics built from
  # `truth - forecast...` (skill, CRPS, mean MSE/RMSE, debiased) come out in
  # truth-first order, the ones built from the forecast alone in forecast order.
  _truth_first = True

  def compute_chunk(self, forecast, truth, region=None, skipna=False,
                    regions: t.Optional[dict] = None):
    forecast, truth = _inputs(forecast, truth)
    _get_n_ensemble(forecast, self.ensemble_dim)  # raises like the reference
    per_var = {}
    with _all_regions(regions):
      for name in _common_vars(forecast, truth):
        geo, by_region, n_member = _fused(
            lambda r, name=name: _ens_pass(forecast, truth, name,
                                           self.ensemble_dim, r, skipna)[:3],
            region, regions)
        lead, values = _pick(by_region, region,
                             _lib.ENS_METRIC_INDEX[self._metric], regions)
        if self._zero_if_single and n_member == 1:
          values = (torch.zeros_like(values)
                    if isinstance(values, torch.Tensor)
                    else np.zeros_like(values))
          if self._zero_is_spatial_average:
            # _spatial_average(zeros): 0 / sum(w) -- NaN over an empty region
            names = list(regions) if regions is not None else [
                _region_set_for(region)[1]]
            empty = [geo.region_wsum[k] == 0.0 for k in names]
            if any(empty):
              values = values.clone() if isinstance(
                  values, torch.Tensor) else values.copy()
              if regions is not None:
                for i, e in enumerate(empty):
                  if e:
                    values[i] = float('nan')
              else:
                values[...] = float('nan')
        dims = lead + geo.out_dims
        if self._truth_first:
          tdims = [d for d in truth[name].dims if d in dims]
          order = lead + tuple(
              tdims + [d for d in dims if d not in tdims and d not in lead])
          values = _transpose(values, [dims.index(d) for d in order])
          dims = order
        per_var[name] = (dims, values)
    return _assemble(forecast, per_var, regions)

  def compute(self, forecast, truth, region=None, skipna=False):
    """Evaluate this metric on datasets with full temporal coverages."""
    forecast, truth = xl.as_dataset(forecast), xl.as_dataset(truth)
    result = super().compute(forecast, truth, region=region, skipna=skipna)
    return result.assign_attrs(ensemble_size=forecast.sizes[self.ensemble_dim])

  def _uses_fused_scalars(self) -> bool:
    """Subclasses with their own compute_chunk / compute (energy scores, maps,
    rank histograms) keep the generic per-region fan-out."""
    return (type(self).compute_chunk is EnsembleMetric.compute_chunk
            and type(self).compute is EnsembleMetric.compute)

  @property
  def _reads_slabs_in_place(self) -> bool:
    """K3 reads a window of chunks through slab addresses (_ens_pass):
    evaluate_chunks may batch chunks for the scalar ensemble metrics."""
    return self._uses_fused_scalars()

  def compute_chunk_regions(self, forecast, truth, regions, skipna=False):
    if not self._uses_fused_scalars():
      return super().compute_chunk_regions(forecast, truth, regions, skipna)
    return self.compute_chunk(forecast, truth, None, skipna, regions)

  def compute_regions(self, forecast, truth, regions, skipna=False):
    if not self._uses_fused_scalars():
      return super().compute_regions(forecast, truth, regions, skipna)
    forecast = xl.as_dataset(forecast)
    return self._mean_regions(forecast, truth, regions, skipna).assign_attrs(
        ensemble_size=forecast.sizes[self.ensemble_dim])


@dataclasses.dataclass
class CRPS(EnsembleMetric):
  """Continuous Ranked Probability Score = skill - 0.5 spread (metrics.py:610-675)."""
  _metric = 'crps'


@dataclasses.dataclass
class CRPSSpread(EnsembleMetric):
  """E|X - X'| (metrics.py:678-694); zero for one member (:783-784)."""
  _metric = 'crps_spread'
  _zero_if_single = True
  _zero_is_spatial_average = True
  _truth_first = False


@dataclasses.dataclass
class CRPSSkill(EnsembleMetric):
  """E|X - Y| (metrics.py:697-715)."""
  _metric = 'crps_skill'


@dataclasses.dataclass
class EnsembleStddevSqrtBeforeTimeAvg(EnsembleMetric):
  """sqrt(spatial mean of std(ddof=1)^2) (metrics.py:1161-1210)."""
  _metric = 'ensemble_stddev'
  _zero_if_single = True
  _truth_first = False


@dataclasses.dataclass
class EnsembleVariance(EnsembleMetric):
  """Spatial mean of var(ddof=1) (metrics.py:1213-1241)."""
  _metric = 'ensemble_variance'
  _zero_if_single = True
  _truth_first = False


@dataclasses.dataclass
class EnsembleMeanRMSESqrtBeforeTimeAvg(EnsembleMetric):
  """RMSE of the ensemble mean (metrics.py:1269-1307)."""
  _metric = 'ensemble_mean_rmse'


@dataclasses.dataclass
class EnsembleMeanMSE(EnsembleMetric):
  """MSE of the ensemble mean (metrics.py:1310-1333)."""
  _metric = 'ensemble_mean_mse'


@dataclasses.dataclass
class DebiasedEnsembleMeanMSE(EnsembleMetric):
  """(t - mean)^2 - var / n (metrics.py:1336-1363); NaN for one member."""
  _metric = 'debiased_ensemble_mean_mse'


# ---------------------------------------------------------------------------
# Spatial* metrics (metrics.py:304-374): maps, no spatial reduction
# ---------------------------------------------------------------------------
def _spatial_inputs(forecast, truth, name, time_first: t.Optional[str] = None):
  """Device tensors + slab tables of one variable for the K5 kernels."""
  fvar, tvar = forecast[name], truth[name]
  geo, prepared = _geometry(forecast, fvar, [tvar])
  out_dims, out_shape = geo.out_dims, geo.out_shape
  if time_first is not None:
    if time_first not in out_dims:
      raise ValueError(f'{time_first!r} missing from {out_dims}')
    order = (time_first,) + tuple(d for d in out_dims if d != time_first)
    out_shape = tuple(out_shape[out_dims.index(d)] for d in order)
    out_dims = order
  tables = [_slab_table(out_dims, out_shape, p[1], p[0].shape[:-2])
            for p in prepared]
  device = engine.require_gpu()
  tensors = [_to_device(p[0], device) for p in prepared]
  dtype = torch.promote_types(tensors[0].dtype, tensors[1].dtype)
  if dtype not in (torch.float32, torch.float64):
    dtype = torch.float64
  tensors = [x if x.dtype == dtype else x.to(dtype) for x in tensors]
  for x in tensors:
    _check_grid(geo, x)
  slabs = [None if tb is None else engine.upload_table(tb, device)
           for tb in tables]
  spatial = _SPATIAL if geo.layout == plan_lib.LATLON else _SPATIAL[::-1]
  n_point = len(geo.latitude) * len(geo.longitude)
  spatial_shape = tuple(tensors[0].shape[-2:])
  return tensors, slabs, out_dims, out_shape, spatial, spatial_shape, n_point


def _spatial_coords(forecast, dims):
  coords = _result_coords(forecast, dims)
  for d in _SPATIAL:
    coords[d] = forecast.coords[d]
  return coords


@dataclasses.dataclass
class _SpatialMetric(Metric):
  """Elementwise map of forecast - truth; `skipna` is ignored by
  compute_chunk exactly like the reference (`del skipna`)."""

  _map = ''

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    del skipna  # Ignored
    forecast, truth = _inputs(forecast, truth)
    out = xl.Dataset()
    for name in _common_vars(forecast, truth):
      pins = (forecast[name].data, truth[name].data)
      key = ('spatial', _stamp(pins[0]), _stamp(pins[1]),
             _coord_sig(forecast, truth))
      hit = _RESULTS.get(key)
      if hit is None:
        (tensors, slabs, out_dims, out_shape, spatial, spatial_shape,
         n_point) = _spatial_inputs(forecast, truth, name)
        n_outer = int(np.prod(out_shape, dtype=np.int64))
        maps = engine.spatial_maps(
            tensors[0].reshape(-1, n_point), slabs[0],
            tensors[1].reshape(-1, n_point), slabs[1], n_outer, n_point)
        dims = tuple(out_dims) + tuple(spatial)
        hit = (dims, {k: v.reshape(out_shape + spatial_shape)
                      for k, v in maps.items()})
        _RESULTS.put(key, pins, hit)
      dims, maps = hit
      out.coords.update(_spatial_coords(forecast, dims))
      out.data_vars[name] = xl.DataArray(maps[self._map], dims, out.coords,
                                         name)
    return out

  def compute(self, forecast, truth, region=None, skipna=False):
    """Temporal mean of the map, fused: no per-time map is materialised."""
    forecast, truth = _inputs(forecast, truth)
    if 'time' in forecast.dims:
      avg_dim = 'time'
    elif 'init_time' in forecast.dims:
      avg_dim = 'init_time'
    else:
      raise ValueError(
          f'Forecast has neither valid_time or init_time dimension {forecast}')
    out = xl.Dataset()
    for name in _common_vars(forecast, truth):
      pins = (forecast[name].data, truth[name].data)
      key = ('spatial_mean', _stamp(pins[0]), _stamp(pins[1]),
             _coord_sig(forecast, truth), bool(skipna))
      hit = _RESULTS.get(key)
      if hit is None:
        (tensors, slabs, out_dims, out_shape, spatial, spatial_shape,
         n_point) = _spatial_inputs(forecast, truth, name, time_first=avg_dim)
        n_time = out_shape[0]
        n_rest = int(np.prod(out_shape[1:], dtype=np.int64))
        dev = tensors[0].device
        total = torch.zeros((3, n_rest, n_point), dtype=torch.float64,
                            device=dev)
        count = torch.zeros_like(total) if skipna else None
        engine.spatial_accumulate(
            tensors[0].reshape(-1, n_point), slabs[0],
            tensors[1].reshape(-1, n_point), slabs[1], n_time, n_rest,
            n_point, skipna, total, count)
        mean = total / (count if skipna else float(n_time))
        mean = mean.to(tensors[0].dtype).reshape(
            (3,) + tuple(out_shape[1:]) + spatial_shape)
        dims = tuple(out_dims[1:]) + tuple(spatial)
        hit = (dims, {'bias': mean[0], 'mse': mean[1], 'mae': mean[2]})
        _RESULTS.put(key, pins, hit)
      dims, maps = hit
      out.coords.update(_spatial_coords(forecast, dims))
      out.data_vars[name] = xl.DataArray(maps[self._map], dims, out.coords,
                                         name)
    return out


@dataclasses.dataclass
class SpatialMSE(_SpatialMetric):
  """MSE without spatial averaging (metrics.py:304-316)."""
  _map = 'mse'


@dataclasses.dataclass
class SpatialMAE(_SpatialMetric):
  """Mean absolute error without spatial averaging (metrics.py:333-345)."""
  _map = 'mae'


@dataclasses.dataclass
class SpatialBias(_SpatialMetric):
  """Bias without spatial averaging (metrics.py:362-374)."""
  _map = 'bias'


def compute_spread_skill_ratio(results):
  """ensemble_stddev / ensemble_mean_rmse along the `metric` dim, NaN at lead
  time 0 (weatherbench2/visualization.py:136-141: `ratio.where(ratio.lead_time
  > np.timedelta64(0))`).  Takes the merged result of the loop -- a Dataset
  (every variable) or, like the reference, one DataArray of it -- with the two
  metrics under those names (EnsembleStddevSqrtBeforeTimeAvg /
  EnsembleMeanRMSESqrtBeforeTimeAvg); xarray in, xarray out.  Results without
  a `lead_time` coordinate (the reference requires one) are returned unmasked."""
  given = results
  if xl.is_xarray(results) and not hasattr(results, 'data_vars'):
    name = results.name if results.name is not None else '__ratio__'
    ds = xl.as_dataset(results.to_dataset(name=name))
    ratio = compute_spread_skill_ratio(ds)[name]
    return xl.like_input(xl.DataArray(ratio.values, ratio.dims, ratio.coords,
                                      results.name), given)
  if isinstance(results, xl.DataArray):
    name = results.name if results.name is not None else '__ratio__'
    ds = xl.Dataset({name: results}, results.coords)
    ratio = compute_spread_skill_ratio(ds)[name]
    return xl.DataArray(ratio.values, ratio.dims, ratio.coords, results.name)
  results = xl.as_dataset(results)
  labels = [str(m) for m in np.atleast_1d(results.coords['metric'])]
  i_std, i_rmse = labels.index('ensemble_stddev'), labels.index(
      'ensemble_mean_rmse')
  out = xl.Dataset(coords={k: v for k, v in results.coords.items()
                           if k != 'metric'})
  lead = results.coords.get('lead_time')
  for name, da in results.data_vars.items():
    ax = da.dims.index('metric')
    a = np.take(np.asarray(da.values), i_std, axis=ax)
    b = np.take(np.asarray(da.values), i_rmse, axis=ax)
    dims = tuple(d for d in da.dims if d != 'metric')
    with np.errstate(all='ignore'):
      ratio = a / b
    if lead is not None and 'lead_time' in dims:
      positive = np.asarray(lead) > np.timedelta64(0)
      shape = [1] * ratio.ndim
      shape[dims.index('lead_time')] = len(positive)
      ratio = np.where(positive.reshape(shape), ratio, np.nan)
    out.data_vars[name] = xl.DataArray(ratio, dims, out.coords, name)
  return xl.like_input(out, given)


# ---------------------------------------------------------------------------
# Tier 2: Gaussian forecasts (metrics.py:849-937) and the energy score
# (metrics.py:1402-1517)
# ---------------------------------------------------------------------------
def _gauss_pass(forecast, truth, name, region, skipna):
  mvar, svar, tvar = forecast[name], forecast[f'{name}_std'], truth[name]
  pins = [mvar.data, svar.data, tvar.data]
  key = _result_key('gauss', pins, region, skipna, (forecast, truth))
  hit = _RESULTS.get(key)
  if hit is not None:
    return hit
  geo, prepared = _geometry(forecast, mvar, [svar, tvar])
  tables = [_slab_table(geo.out_dims, geo.out_shape, p[1], p[0].shape[:-2])
            for p in prepared]
  by_region, _ = _run_pass(_lib.MODE_GAUSS, geo, [p[0] for p in prepared],
                           tables, region, skipna)
  value = (geo, by_region)
  _RESULTS.put(key, tuple(pins), value)
  return value


class _GaussianMetric(Metric):
  _row = 0

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    forecast, truth = _inputs(forecast, truth)
    per_var = {}
    for name in [v for v in forecast.keys() if f'{v}_std' in forecast.keys()]:
      if name not in truth:
        raise KeyError(name)
      geo, by_region = _gauss_pass(forecast, truth, name, region, skipna)
      _, rkey = _region_set_for(region)
      per_var[name] = (geo.out_dims, by_region[rkey][self._row])
    return _assemble(forecast, per_var)


@dataclasses.dataclass
class GaussianCRPS(_GaussianMetric):
  """The analytical CRPS of a Gaussian forecast given as `<var>` (mean) and
  `<var>_std` (metrics.py:849-905)."""
  _row = 0


@dataclasses.dataclass
class GaussianVariance(_GaussianMetric):
  """The variance of a Gaussian forecast (metrics.py:908-937)."""
  _row = 1


def _energy_pass(forecast, truth, name, ensemble_dim, region, skipna):
  """(score, spread, skill) of one variable for the active regions from ONE
  read of the ensemble (wb2_energy_score: the 2 M - 1 per-member sums of
  metrics.py:1468-1517 accumulated by blocks of members)."""
  fvar, tvar = forecast[name], truth[name]
  pins = [fvar.data, tvar.data]
  key = _result_key(('energy', ensemble_dim), pins, region, skipna,
                    (forecast, truth))
  hit = _RESULTS.get(key)
  if hit is not None:
    return hit
  (geo, ften, tten, ens_table, truth_table, member_slabs, n_member, device,
   _) = _ens_layout(forecast, fvar, tvar, ensemble_dim)
  regions, _ = _region_set_for(region)
  pl = plan_lib.cached_plan(
      geo.latitude, geo.longitude, geo.layout, regions, device,
      plan_lib.ENERGY_ROWS_PER_CHUNK)
  slab_elems = pl.n_row * pl.n_col
  out = engine.energy_score(
      pl, ften, member_slabs * slab_elems, n_member, ens_table,
      tten.reshape(-1, pl.n_row, pl.n_col), truth_table, geo.n_outer, skipna)
  dev = out.reshape((3, pl.n_region) + geo.out_shape)
  value = (geo, {nm: dev[:, i] for i, nm in enumerate(pl.region_names)},
           n_member)
  _RESULTS.put(key, tuple(pins), value)
  return value


@dataclasses.dataclass
class _EnergyMetric(EnsembleMetric):
  """Shared body of the three energy-score metrics: rows of the fused pass."""
  _row = 0
  _truth_first = False  # `forecast - truth` / the forecast alone come first

  def compute_chunk(self, forecast, truth, region=None, skipna=False,
                    regions: t.Optional[dict] = None):
    forecast, truth = _inputs(forecast, truth)
    _get_n_ensemble(forecast, self.ensemble_dim)  # raises like the reference
    per_var = {}
    with _all_regions(regions):
      for name in _common_vars(forecast, truth):
        geo, by_region, _ = _fused(
            lambda r, name=name: _energy_pass(forecast, truth, name,
                                              self.ensemble_dim, r, skipna),
            region, regions)
        dtype = _reference_result_dtype(
            forecast, [_np_dtype(forecast[name].data),
                       _np_dtype(truth[name].data)], region, regions)
        lead, values = _pick(by_region, region, self._row, regions)
        per_var[name] = (lead + geo.out_dims, values, dtype)
    return _assemble(forecast, per_var, regions)

  def compute_chunk_regions(self, forecast, truth, regions, skipna=False):
    return self.compute_chunk(forecast, truth, None, skipna, regions)

  def compute_regions(self, forecast, truth, regions, skipna=False):
    forecast = xl.as_dataset(forecast)
    return self._mean_regions(forecast, truth, regions, skipna).assign_attrs(
        ensemble_size=forecast.sizes[self.ensemble_dim])


@dataclasses.dataclass
class EnergyScoreSkill(_EnergyMetric):
  """E||X - Y||: member-wise area-weighted L2 norms, averaged over members
  (metrics.py:1501-1517)."""
  _row = 2


@dataclasses.dataclass
class EnergyScoreSpread(_EnergyMetric):
  """E||X - X'|| from the N-1 adjacent member differences
  (metrics.py:1468-1498); zeros for one member."""
  _row = 1


@dataclasses.dataclass
class EnergyScore(_EnergyMetric):
  """ES = E||X - Y|| - 0.5 E||X - X'|| (metrics.py:1402-1465)."""
  _row = 0


# ---------------------------------------------------------------------------
# Tier 2: threshold metrics (metrics.py:940-1158 Gaussian, 1524-1891 ensemble)
# ---------------------------------------------------------------------------
def _stack_quantiles(forecast, per_threshold: list, quantiles, method: str,
                     sum_over_quantile: bool) -> xl.Dataset:
  """metrics.py:966-972: concat over a new leading `quantile` dim (+ attrs)."""
  out = xl.Dataset()
  first = per_threshold[0]
  for name, (dims, _) in first.items():
    out.coords.update(_result_coords(forecast, dims))
  if not sum_over_quantile:
    out.coords['quantile'] = np.array(list(quantiles), dtype=np.float64)
  for name, (dims, _) in first.items():
    data = _stack([p[name][1] for p in per_threshold])
    if sum_over_quantile:
      out.data_vars[name] = xl.DataArray(_nansum_leading(data), dims,
                                         out.coords, name)
    else:
      out.data_vars[name] = xl.DataArray(data, ('quantile',) + tuple(dims),
                                         out.coords, name)
  return out.assign_attrs(threshold_method=method)


def _gauss_threshold_pass(forecast, truth, threshold_ds, name, region, skipna):
  mvar, svar = forecast[name], forecast[f'{name}_std']
  tvar, hvar = truth[name], threshold_ds[name]
  geo, prepared = _geometry(forecast, mvar, [svar, tvar, hvar])
  tables = [_slab_table(geo.out_dims, geo.out_shape, p[1], p[0].shape[:-2])
            for p in prepared]
  by_region, _ = _run_pass(_lib.MODE_GAUSS_THR, geo, [p[0] for p in prepared],
                           tables, region, skipna)
  return geo, by_region


@dataclasses.dataclass
class ThresholdMetric(Metric):
  """Base of the threshold metrics (metrics.py:940-972): one score per
  threshold, stacked along `quantile`."""

  thresholds: t.Sequence = ()
  _row = 0
  _sum_over_quantile = False
  # dims of the LEFT operand come first in xarray: scores whose outermost
  # operand is derived from truth / the threshold are truth-first.
  _truth_first = True


def _truth_first_order(dims, values, truth_dims):
  tdims = [d for d in truth_dims if d in dims]
  order = tuple(tdims + [d for d in dims if d not in tdims])
  if order == tuple(dims):
    return dims, values
  return order, _transpose(values, [dims.index(d) for d in order])


def _order_by(dims, values, precedence):
  """Reorders (dims, values) like xarray broadcasting would: dims in order of
  first appearance over the operand dim tuples in `precedence`."""
  order = []
  for operand in precedence:
    order += [d for d in operand if d in dims and d not in order]
  order += [d for d in dims if d not in order]
  order = tuple(order)
  if order == tuple(dims):
    return dims, values
  return order, _transpose(values, [dims.index(d) for d in order])


class _GaussianThresholdMetric(ThresholdMetric):
  _truth_leads = False

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    forecast, truth = _inputs(forecast, truth)
    per_threshold = []
    for threshold in self.thresholds:
      threshold_ds = threshold.compute(truth)
      per_var = {}
      for name in [v for v in forecast.keys()
                   if f'{v}_std' in forecast.keys()]:
        geo, by_region = _gauss_threshold_pass(forecast, truth, threshold_ds,
                                               name, region, skipna)
        _, rkey = _region_set_for(region)
        # xarray keeps the dims of the LEFT operand first.  Brier / RPS are
        # ((threshold - mean) / std -> cdf) - truth_indicator: threshold dims,
        # then the forecast's, then the truth's (:990-1000, 1111-1121); the
        # ignorance score is xr.where(truth_indicator, ...): truth (with the
        # threshold it was compared to) first (:1057-1066).
        thr, mean, std = (threshold_ds[name].dims, forecast[name].dims,
                          forecast[f'{name}_std'].dims)
        if self._truth_leads:
          precedence = (truth[name].dims, thr, mean, std)
        else:
          precedence = (thr, mean, std, truth[name].dims)
        per_var[name] = _order_by(geo.out_dims, by_region[rkey][self._row],
                                  precedence)
      per_threshold.append(per_var)
    return _stack_quantiles(forecast, per_threshold,
                            [th.quantile for th in self.thresholds],
                            type(self.thresholds[0]).__name__,
                            self._sum_over_quantile)


@dataclasses.dataclass
class GaussianBrierScore(_GaussianThresholdMetric):
  """Brier score of a Gaussian forecast (metrics.py:1003-1040)."""
  _row = 0


@dataclasses.dataclass
class GaussianIgnoranceScore(_GaussianThresholdMetric):
  """Ignorance (log) score of a Gaussian forecast (metrics.py:1069-1101)."""
  _row = 1
  _truth_leads = True


@dataclasses.dataclass
class GaussianRPS(_GaussianThresholdMetric):
  """Ranked probability score of a Gaussian forecast: the per-threshold
  (cdf - truth_ecdf)^2 summed over the thresholds (metrics.py:1124-1158)."""
  _row = 2
  _sum_over_quantile = True


def _ens_threshold_layout(forecast, truth, threshold_ds, name, ensemble_dim):
  """Device tensors and slab tables (members, truth, threshold) of one
  variable; the threshold may carry any subset of the output dims."""
  fvar, tvar, hvar = forecast[name], truth[name], threshold_ds[name]
  if ensemble_dim not in fvar.dims:
    raise ValueError(f'{ensemble_dim=} not found in {fvar.dims=}')
  fdata, frest, layout = _spatial_last(fvar, None)
  tdata, trest, _ = _spatial_last(tvar, layout)
  hdata, hrest, _ = _spatial_last(hvar, layout)
  fsizes = dict(zip(frest, fdata.shape[:-2]))
  n_member = fsizes[ensemble_dim]
  out_dims = [d for d in frest if d != ensemble_dim]
  sizes = {d: fsizes[d] for d in out_dims}
  for rest, data in ((trest, tdata), (hrest, hdata)):
    for d, n in zip(rest, data.shape[:-2]):
      if d not in out_dims:
        out_dims.append(d)
        sizes[d] = n
  out_dims = tuple(out_dims)
  out_shape = tuple(sizes[d] for d in out_dims)
  geo = _Geometry(layout, out_dims, out_shape,
                  _coord_values(forecast, 'latitude'),
                  _coord_values(forecast, 'longitude'))
  stride, strides = 1, {}
  for d in reversed(frest):
    strides[d] = stride
    stride *= fsizes[d]
  ens_table = np.zeros(out_shape, dtype=np.int64)
  for d in frest:
    if d == ensemble_dim:
      continue
    shape = [1] * len(out_shape)
    shape[out_dims.index(d)] = fsizes[d]
    ens_table = ens_table + (np.arange(fsizes[d], dtype=np.int64)
                             * strides[d]).reshape(shape)
  ens_table = np.ascontiguousarray(ens_table).ravel()
  identity = np.array_equal(ens_table, np.arange(ens_table.size))
  t_table = _slab_table(out_dims, out_shape, trest, tdata.shape[:-2])
  h_table = _slab_table(out_dims, out_shape, hrest, hdata.shape[:-2])
  device = engine.require_gpu()
  tens = [_to_device(x, device) for x in (fdata, tdata, hdata)]
  dtype = tens[0].dtype
  for x in tens[1:]:
    dtype = torch.promote_types(dtype, x.dtype)
  if dtype not in (torch.float32, torch.float64):
    dtype = torch.float64
  tens = [x if x.dtype == dtype else x.to(dtype) for x in tens]
  for x in tens:
    _check_grid(geo, x)
  to_dev = lambda tb: None if tb is None else engine.upload_table(tb, device)
  tables = [None if identity else to_dev(ens_table), to_dev(t_table),
            to_dev(h_table)]
  return geo, tens, tables, strides[ensemble_dim], n_member, device


def _ens_threshold_pass(forecast, truth, threshold_ds, name, ensemble_dim,
                        region, skipna):
  geo, tens, tables, member_slabs, n_member, device = _ens_threshold_layout(
      forecast, truth, threshold_ds, name, ensemble_dim)
  regions, _ = _region_set_for(region)
  pl = plan_lib.cached_plan(geo.latitude, geo.longitude, geo.layout, regions,
                            device, plan_lib.ENSEMBLE_ROWS_PER_CHUNK)
  slab_elems = pl.n_row * pl.n_col
  metrics = engine.ensemble_threshold_reduce(
      pl, tens[0], member_slabs * slab_elems, n_member, tables[0],
      tens[1].reshape(-1, pl.n_row, pl.n_col), tables[1],
      tens[2].reshape(-1, pl.n_row, pl.n_col), tables[2], geo.n_outer, skipna)
  dev = metrics.reshape((4, pl.n_region) + geo.out_shape)
  return geo, {nm: dev[:, i] for i, nm in enumerate(pl.region_names)}


def _ens_threshold_maps(forecast, truth, threshold_ds, name, ensemble_dim,
                        skipna):
  """The four pointwise score maps [4, *out_shape, n_row, n_col] (device)."""
  geo, tens, tables, member_slabs, n_member, _ = _ens_threshold_layout(
      forecast, truth, threshold_ds, name, ensemble_dim)
  n_point = tens[0].shape[-2] * tens[0].shape[-1]
  maps = engine.ensemble_threshold_maps(
      tens[0], member_slabs * n_point, n_member, tables[0],
      tens[1].reshape(-1, n_point), tables[1],
      tens[2].reshape(-1, n_point), tables[2], geo.n_outer, n_point, skipna)
  return geo, maps.reshape((4,) + geo.out_shape + tuple(tens[0].shape[-2:]))


@dataclasses.dataclass
class _EnsembleThresholdMetric(ThresholdMetric):
  ensemble_dim: str = REALIZATION

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    forecast, truth = _inputs(forecast, truth)
    _get_n_ensemble(forecast, self.ensemble_dim)
    per_threshold = []
    for threshold in self.thresholds:
      threshold_ds = threshold.compute(truth)
      per_var = {}
      for name in _common_vars(forecast, truth):
        key = _result_key(('ens_thr', self.ensemble_dim, id(threshold)),
                          [forecast[name].data, truth[name].data], region,
                          skipna, (forecast, truth))
        hit = _RESULTS.get(key)
        if hit is None:
          hit = _ens_threshold_pass(forecast, truth, threshold_ds, name,
                                    self.ensemble_dim, region, skipna)
          _RESULTS.put(key, (forecast[name].data, truth[name].data, threshold),
                       hit)
        geo, by_region = hit
        _, rkey = _region_set_for(region)
        # dims in xarray's left-operand-first order (:1524-1560, 532-565,
        # 1722-1738, 1805-1817): forecast-led for the plain Brier score and
        # the RPS part, truth-led (with its threshold) for the debiased Brier
        # and the ignorance score
        fdims = tuple(d for d in forecast[name].dims if d != self.ensemble_dim)
        tdims, hdims = truth[name].dims, threshold_ds[name].dims
        precedence = ((tdims, hdims, fdims) if self._truth_first
                      else (fdims, hdims, tdims))
        dims, values = _order_by(geo.out_dims, by_region[rkey][self._row],
                                 precedence)
        per_var[name] = (dims, values)
      per_threshold.append(per_var)
    return _stack_quantiles(forecast, per_threshold,
                            [th.quantile for th in self.thresholds],
                            type(self.thresholds[0]).__name__,
                            self._sum_over_quantile)

  def compute(self, forecast, truth, region=None, skipna=False):
    forecast = xl.as_dataset(forecast)
    result = super().compute(forecast, truth, region=region, skipna=skipna)
    return result.assign_attrs(ensemble_size=forecast.sizes[self.ensemble_dim])


@dataclasses.dataclass
class EnsembleBrierScore(_EnsembleThresholdMetric):
  """Brier score of an ensemble forecast (metrics.py:1563-1617)."""
  _row = 0
  _truth_first = False


@dataclasses.dataclass
class DebiasedEnsembleBrierScore(_EnsembleThresholdMetric):
  """Debiased ensemble Brier score (metrics.py:1644-1704)."""
  _row = 1


@dataclasses.dataclass
class EnsembleIgnoranceScore(_EnsembleThresholdMetric):
  """Ignorance score of an ensemble forecast (metrics.py:1741-1788)."""
  _row = 2


@dataclasses.dataclass
class EnsembleRPS(_EnsembleThresholdMetric):
  """Ranked probability score of an ensemble forecast (metrics.py:1805-1868)."""
  _row = 3
  _sum_over_quantile = True
  _truth_first = False


@dataclasses.dataclass
class SEEPS(Metric):
  """Spatially averaged Stable Equitable Error in Probability Space
  (metrics.py:417-524).

  Attributes as in the reference: `climatology` holds
  `<precip_name>_seeps_threshold` [same units as the data] and
  `<precip_name>_seeps_dry_fraction` over (hour, dayofyear, lat, lon).
  NaNs are always skipped (the p1 mask makes that mandatory, :513, 523).
  """

  climatology: t.Any = None
  dry_threshold_mm: float = 0.25
  precip_name: str = 'total_precipitation_24hr'
  min_p1: float = 0.1
  max_p1: float = 0.85

  def _p1(self, climatology) -> xl.DataArray:
    frac = climatology[f'{self.precip_name}_seeps_dry_fraction']
    # metrics.py:443-444 takes xarray's default skipna=None: NaN dry fractions
    # are skipped (a point is masked only if it is NaN at every hour and day)
    return frac.mean(('hour', 'dayofyear'), skipna=True)

  # reads concatenated chunks in place like the deterministic suite (one fused
  # pass through slab addresses): evaluate_chunks may batch chunks for it
  _reads_slabs_in_place = True

  def _masked_p1(self, climatology, layout) -> np.ndarray:
    """p1 (the climatological dry fraction averaged over hour and day of year,
    metrics.py:443-444) with NaN outside (min_p1, max_p1) (:504-506), in slab
    orientation -- a property of the climatology, computed once per metric
    object and climatology (the mean runs over the whole
    (hour, dayofyear, lat, lon) array: it must not be redone per chunk)."""
    key = (id(climatology), layout, self.precip_name, self.min_p1, self.max_p1)
    hit = getattr(self, '_p1_cache', None)
    if hit is not None and hit[0] == key and hit[1] is climatology:
      return hit[2]
    p1 = self._p1(climatology)
    want = _SPATIAL if layout == plan_lib.LATLON else _SPATIAL[::-1]
    p1v = p1.transpose(*want).values
    with np.errstate(invalid='ignore'):
      keep = np.logical_and(p1v < self.max_p1, p1v > self.min_p1)
    aux = np.where(keep, p1v.astype(np.float64), np.nan)
    object.__setattr__(self, '_p1_cache', (key, climatology, aux))
    return aux

  def _prepare(self, forecast, truth):
    """(geo, [forecast, truth, wet threshold], slab tables, masked p1)."""
    climatology = xl.as_dataset(self.climatology)
    name = self.precip_name
    fvar, tvar = forecast[name], truth[name]
    wvar = climatology[f'{name}_seeps_threshold']
    geo, prepared = _geometry(forecast, fvar, [tvar])
    tables = [_slab_table(geo.out_dims, geo.out_shape, p[1], p[0].shape[:-2])
              for p in prepared]
    wrest = tuple(d for d in wvar.dims if d not in _SPATIAL)
    wdata, _, _ = _spatial_last(wvar, geo.layout)
    prepared.append((wdata, wrest))
    wtable = _climatology_slabs(climatology, wvar, forecast, geo, wrest)
    rec = program.recorder()
    if rec is not None and not rec.probe:
      sizes = tuple(wvar.sizes[d] for d in wrest)

      def recompute(other, memo, climatology=climatology, wvar=wvar, geo=geo,
                    wrest=wrest, sizes=sizes):
        key = (id(climatology), wrest, sizes, geo.out_dims, geo.out_shape)
        if key not in memo:
          memo[key] = _climatology_slabs_by_content(climatology, wvar, other,
                                                    geo, wrest)
        return memo[key]
      rec.note_table(wtable, recompute, _climatology_gather(
          climatology, wvar, forecast, geo, wrest))
    tables.append(wtable)
    aux = self._masked_p1(climatology, geo.layout)
    return geo, [p[0] for p in prepared], tables, aux

  def _pass(self, forecast, truth, region):
    """(geo, by_region) of the fused SEEPS pass for the active regions: one
    launch per weight-field group answers every region of it (cached for the
    chunk, like the deterministic passes)."""
    name = self.precip_name
    pins = [forecast[name].data, truth[name].data]
    key = _result_key(('seeps', name, self.dry_threshold_mm, self.min_p1,
                       self.max_p1, id(self.climatology)), pins, region, True,
                      (forecast, truth))
    hit = _RESULTS.get(key)
    if hit is not None:
      return hit
    geo, arrays, tables, aux = self._prepare(forecast, truth)
    by_region, _ = _run_pass(
        _lib.MODE_SEEPS, geo, arrays, tables, region, True,
        aux=aux, scalar=self.dry_threshold_mm / 1000.0)
    value = (geo, by_region)
    _RESULTS.put(key, tuple(pins), value)
    return value

  def compute_chunk(self, forecast, truth, region=None, skipna=False,
                    regions: t.Optional[dict] = None):
    del skipna  # Ignored, must be effectively True because of p1 mask.
    forecast, truth = _inputs(forecast, truth)
    with _all_regions(regions):
      geo, by_region = _fused(lambda r: self._pass(forecast, truth, r), region,
                              regions)
      lead, values = _pick(by_region, region, 0, regions)
    return _assemble(forecast, {self.precip_name: (lead + geo.out_dims,
                                                   values)}, regions)

  def compute_chunk_regions(self, forecast, truth, regions, skipna=False):
    return self.compute_chunk(forecast, truth, None, skipna, regions)

  def compute_regions(self, forecast, truth, regions, skipna=False):
    return self._mean_regions(forecast, truth, regions, skipna)


# ---------------------------------------------------------------------------
# Spatial* ensemble metrics (metrics.py:718-772, 1244-1266, 1366-1399): the
# pointwise maps of the fused ensemble pass, kept on the device
# ---------------------------------------------------------------------------
_ENS_MAP_SLOT = {'skill': 0, 'spread': 1, 'mse': 2, 'var': 3, 'debiased': 5}


def _device_temporal_mean(metric, forecast, truth, region, skipna,
                          ensemble_dim=None) -> xl.Dataset:
  """Metric.compute (metrics.py:117-138) for map-valued metrics whose chunk
  result lives on the device: (sum, count) over the time dim via
  wb2_time_accumulate, no host round trip of the per-time maps."""
  forecast_ds = xl.as_dataset(forecast)
  avg_dim = 'time' if 'time' in forecast_ds.dims else 'init_time'
  if avg_dim not in forecast_ds.dims:
    raise ValueError(
        'Forecast has neither valid_time or init_time dimension '
        f'{forecast_ds}')
  chunk = metric.compute_chunk(forecast, truth, region=region, skipna=skipna)
  attrs = dict(chunk.attrs)
  if ensemble_dim is not None:
    attrs['ensemble_size'] = forecast_ds.sizes[ensemble_dim]
  out = xl.Dataset(coords={k: v for k, v in chunk.coords.items()
                           if k != avg_dim}, attrs=attrs)
  for name, da in chunk.data_vars.items():
    axis = da.dims.index(avg_dim)
    values = da.data.to(torch.float64).contiguous()
    shape = tuple(n for i, n in enumerate(values.shape) if i != axis)
    total = torch.zeros(shape, dtype=torch.float64, device=values.device)
    count = torch.zeros_like(total)
    engine.time_accumulate(values, axis, skipna, total, count)
    mean = (total / count).to(da.data.dtype)
    out.data_vars[name] = xl.DataArray(
        mean, tuple(d for d in da.dims if d != avg_dim), out.coords, name)
  return out


@dataclasses.dataclass
class _SpatialEnsembleMetric(EnsembleMetric):
  _slot = ''
  _truth_first = True
  _zero_if_single = False

  def _map(self, maps, dtype, n_member):
    m = maps[_ENS_MAP_SLOT[self._slot]]
    if self._zero_if_single and n_member == 1:
      return torch.zeros_like(m, dtype=dtype)  # zeros_like(forecast): input dtype
    # everything but the rank-weighted spread lives in the input dtype
    return m if self._slot == 'spread' else m.to(dtype)

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    forecast, truth = _inputs(forecast, truth)
    _get_n_ensemble(forecast, self.ensemble_dim)
    out = xl.Dataset()
    for name in _common_vars(forecast, truth):
      geo, _, n_member, maps, dtype = _ens_pass(
          forecast, truth, name, self.ensemble_dim, None, skipna,
          want_maps=True)
      spatial = _SPATIAL if geo.layout == plan_lib.LATLON else _SPATIAL[::-1]
      dims = tuple(geo.out_dims) + tuple(spatial)
      data = self._map(maps, dtype, n_member)
      if self._truth_first:
        tdims = [d for d in truth[name].dims if d in dims]
        order = tuple(tdims + [d for d in dims if d not in tdims])
        data = data.permute(*[dims.index(d) for d in order])
        dims = order
      out.coords.update(_spatial_coords(forecast, dims))
      out.data_vars[name] = xl.DataArray(data, dims, out.coords, name)
    return out

  def compute(self, forecast, truth, region=None, skipna=False):
    """Temporal mean of the map, accumulated on the device."""
    return _device_temporal_mean(self, forecast, truth, region, skipna,
                                 self.ensemble_dim)


@dataclasses.dataclass
class SpatialCRPSSkill(_SpatialEnsembleMetric):
  """CRPSSkill without spatial averaging (metrics.py:757-772)."""
  _slot = 'skill'


@dataclasses.dataclass
class SpatialCRPSSpread(_SpatialEnsembleMetric):
  """CRPSSpread without spatial averaging (metrics.py:742-754)."""
  _slot = 'spread'
  _truth_first = False
  _zero_if_single = True


@dataclasses.dataclass
class SpatialCRPS(_SpatialEnsembleMetric):
  """CRPS without spatial averaging: skill - 0.5 spread (metrics.py:718-739)."""
  _slot = 'skill'

  def _map(self, maps, dtype, n_member):
    skill = maps[_ENS_MAP_SLOT['skill']].to(dtype)
    if n_member == 1:
      return skill  # spread is zeros_like(forecast): stays in the input dtype
    # float32 skill - 0.5 * float64 spread -> float64, like numpy promotes
    return skill.to(torch.float64) - 0.5 * maps[_ENS_MAP_SLOT['spread']]


@dataclasses.dataclass
class SpatialEnsembleVariance(_SpatialEnsembleMetric):
  """Ensemble variance without spatial averaging (metrics.py:1244-1266)."""
  _slot = 'var'
  _truth_first = False
  _zero_if_single = True


@dataclasses.dataclass
class SpatialEnsembleMeanMSE(_SpatialEnsembleMetric):
  """(truth - ensemble mean)^2 as a map (metrics.py:1366-1381)."""
  _slot = 'mse'


@dataclasses.dataclass
class DebiasedSpatialEnsembleMeanMSE(_SpatialEnsembleMetric):
  """Debiased (truth - ensemble mean)^2 as a map (metrics.py:1384-1399)."""
  _slot = 'debiased'



# ---------------------------------------------------------------------------
# Spatial* threshold metrics and SpatialSEEPS: unreduced float64 maps on the
# device (`region` is ignored, like the reference: spatial_agg=False)
# ---------------------------------------------------------------------------
@dataclasses.dataclass
class _SpatialEnsembleThresholdMetric(ThresholdMetric):
  ensemble_dim: str = REALIZATION

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    forecast, truth = _inputs(forecast, truth)
    _get_n_ensemble(forecast, self.ensemble_dim)
    out = xl.Dataset()
    names = _common_vars(forecast, truth)
    stacks = {name: [] for name in names}
    for threshold in self.thresholds:
      threshold_ds = threshold.compute(truth)
      for name in names:
        geo, maps = _ens_threshold_maps(forecast, truth, threshold_ds, name,
                                        self.ensemble_dim, skipna)
        spatial = _SPATIAL if geo.layout == plan_lib.LATLON else _SPATIAL[::-1]
        dims = tuple(geo.out_dims) + tuple(spatial)
        data = maps[self._row]
        if self._truth_first:
          tdims = [d for d in truth[name].dims if d in dims]
          order = tuple(tdims + [d for d in dims if d not in tdims])
          data = data.permute(*[dims.index(d) for d in order])
          dims = order
        stacks[name].append((dims, data))
    for name in names:
      dims = stacks[name][0][0]
      data = torch.stack([d for _, d in stacks[name]])
      out.coords.update(_spatial_coords(forecast, dims))
      if self._sum_over_quantile:
        out.data_vars[name] = xl.DataArray(_nansum_leading(data), dims,
                                           out.coords, name)
      else:
        out.coords['quantile'] = np.array(
            [th.quantile for th in self.thresholds], dtype=np.float64)
        out.data_vars[name] = xl.DataArray(data, ('quantile',) + tuple(dims),
                                           out.coords, name)
    return out.assign_attrs(
        threshold_method=type(self.thresholds[0]).__name__)

  def compute(self, forecast, truth, region=None, skipna=False):
    return _device_temporal_mean(self, forecast, truth, region, skipna,
                                 self.ensemble_dim)


@dataclasses.dataclass
class SpatialEnsembleBrierScore(_SpatialEnsembleThresholdMetric):
  """Spatial map of ensemble Brier score (metrics.py:1615-1638)."""
  _row = 0
  _truth_first = False


@dataclasses.dataclass
class SpatialDebiasedEnsembleBrierScore(_SpatialEnsembleThresholdMetric):
  """Spatial map of ensemble debiased Brier score (metrics.py:1697-1719)."""
  _row = 1


@dataclasses.dataclass
class SpatialEnsembleIgnoranceScore(_SpatialEnsembleThresholdMetric):
  """Spatial map of ensemble ignorance score (metrics.py:1780-1802)."""
  _row = 2


@dataclasses.dataclass
class SpatialEnsembleRPS(_SpatialEnsembleThresholdMetric):
  """Spatial map of ensemble RPS (metrics.py:1870-1891)."""
  _row = 3
  _sum_over_quantile = True
  _truth_first = False


@dataclasses.dataclass
class SpatialSEEPS(SEEPS):
  """SEEPS without spatial averaging (metrics.py:418-509): float64 map."""

  _reads_slabs_in_place = False  # the map kernel reads whole arrays
  # maps have no regions: the generic per-region fan-out of the base class
  compute_chunk_regions = Metric.compute_chunk_regions
  compute_regions = Metric.compute_regions

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    del region, skipna
    forecast, truth = _inputs(forecast, truth)
    geo, arrays, tables, aux = self._prepare(forecast, truth)
    device = engine.require_gpu()
    tensors = [_to_device(a, device) for a in arrays]
    dtype = torch.result_type(tensors[0], tensors[1])
    dtype = torch.promote_types(dtype, tensors[2].dtype)
    if dtype not in (torch.float32, torch.float64):
      dtype = torch.float64
    tensors = [x if x.dtype == dtype else x.to(dtype) for x in tensors]
    for x in tensors:
      _check_grid(geo, x)
    n_point = tensors[0].shape[-2] * tensors[0].shape[-1]
    slabs = [None if tb is None else engine.upload_table(tb, device)
             for tb in tables]
    aux_dev = _resident_aux(aux, device).reshape(-1)  # (one copy per p1)
    out_map = engine.seeps_map(
        [x.reshape(-1, n_point) for x in tensors], slabs, geo.n_outer, n_point,
        aux_dev, self.dry_threshold_mm / 1000.0)
    spatial = _SPATIAL if geo.layout == plan_lib.LATLON else _SPATIAL[::-1]
    dims = tuple(geo.out_dims) + tuple(spatial)
    out = xl.Dataset()
    out.coords.update(_spatial_coords(forecast, dims))
    out.data_vars[self.precip_name] = xl.DataArray(
        out_map.reshape(geo.out_shape + tuple(tensors[0].shape[-2:])), dims,
        out.coords, self.precip_name)
    return out

  def compute(self, forecast, truth, region=None, skipna=False):
    # the reference's Metric.compute passes skipna on to the time mean
    return _device_temporal_mean(self, forecast, truth, region, skipna)

# ---------------------------------------------------------------------------
# RankHistogram / central_reliability (metrics.py:1894-2126)
# ---------------------------------------------------------------------------
class RankHistogram(EnsembleMetric):
  """Histogram of truth's rank among the ensemble members (metrics.py:1894-2042).

  compute_chunk returns the float64 one-hot encoding with a trailing `bins`
  dim; compute() accumulates the temporal mean on the device without
  materialising the per-time one-hots.  Ranks come from wb2_rank_histogram
  (counting, no sort): identical to the reference wherever truth differs from
  every member.  With a `seed` the reference's random tie breaking itself is
  reproduced: the kernel jumps NumPy's PCG64 (`np.random.default_rng(seed)`) to
  the position each element of the reference's concatenated [truth, members]
  array has in the stream and applies the same perturbation
  (metrics.py:1955-1980); `seed=None` (fresh entropy in the reference too)
  draws from a counter-based stream with the same distribution.
  NaNs rank highest and `skipna` is ignored, like the reference.
  """

  def __init__(self, ensemble_dim: str = REALIZATION,
               num_bins: t.Optional[int] = None,
               break_ties_randomly: bool = True,
               seed: t.Optional[int] = None):
    super().__init__(ensemble_dim=ensemble_dim)
    self.num_bins = num_bins
    self._break_ties_randomly = break_ties_randomly
    self._seed = seed

  def _num_bins_actual(self, ensemble_size: int) -> int:
    """metrics.py:1927-1939 (same message)."""
    default_n_bins = ensemble_size + 1
    if self.num_bins is None:
      return default_n_bins
    if default_n_bins % self.num_bins:
      raise ValueError(
          f'Cannot bin data with {ensemble_size=} into {self.num_bins} bins')
    return self.num_bins

  def _concat_dims(self, fvar, tvar) -> tuple:
    """Dim order of the reference's `xr.concat([truth, forecast],
    dim=ensemble_dim)` (metrics.py:2014): xarray lists the dims of the first
    object, then the unseen dims of the next ones in their order
    (core/concat.py, ensure_common_dims) -- the truth's dims, then the ensemble
    dim, then whatever only the forecast has.  It fixes both the result's dim
    order and the order in which the seeded perturbation stream is consumed."""
    cdims = list(tvar.dims) + [d for d in fvar.dims if d not in tvar.dims]
    return tuple(cdims)

  def _numpy_stream(self, fvar, tvar, geo, n_member, device):
    """Where in np.random.default_rng(seed)'s stream every element of the
    reference's concatenated array gets its perturbation from (C order of the
    concat layout), as strides for wb2_rank_histogram_seeded."""
    cdims = self._concat_dims(fvar, tvar)
    sizes = {**dict(tvar.sizes), **dict(fvar.sizes)}
    sizes[self.ensemble_dim] = n_member + 1
    stride, acc = {}, 1
    for d in reversed(cdims):
      stride[d] = acc
      acc *= sizes[d]
    off = np.zeros(geo.out_shape, dtype=np.int64)
    for ax, (d, n) in enumerate(zip(geo.out_dims, geo.out_shape)):
      shape = [1] * len(geo.out_shape)
      shape[ax] = n
      off = off + (np.arange(n, dtype=np.int64) * stride[d]).reshape(shape)
    row_dim, col_dim = (_SPATIAL if geo.layout == plan_lib.LATLON
                        else _SPATIAL[::-1])
    state = np.random.PCG64(self._seed).state['state']
    n_col = len(geo.longitude if geo.layout == plan_lib.LATLON
                else geo.latitude)
    return (int(state['state']), int(state['inc']),
            engine.upload_table(np.ascontiguousarray(off).ravel(), device),
            (stride[row_dim], stride[col_dim], stride[self.ensemble_dim]), n_col)

  def _histogram(self, forecast, truth, name, avg_dim=None):
    fvar, tvar = forecast[name], truth[name]
    (geo, ften, tten, ens_table, truth_table, member_slabs, n_member, device,
     _) = _ens_layout(forecast, fvar, tvar, self.ensemble_dim)
    n_bins = self._num_bins_actual(n_member)
    n_point = ften.shape[-2] * ften.shape[-1]
    seed = self._seed
    numpy_stream = None
    if seed is None:
      seed = int.from_bytes(os.urandom(8), 'little')
    elif self._break_ties_randomly:
      # a seeded run reproduces the reference's own draws (NumPy's PCG64)
      numpy_stream = self._numpy_stream(fvar, tvar, geo, n_member, device)
      seed = 0
    dims, shape = geo.out_dims, geo.out_shape
    mean_over, acc_row, n_acc = None, None, 0
    if avg_dim is not None:
      axis = dims.index(avg_dim)
      out_shape = tuple(n for i, n in enumerate(shape) if i != axis)
      if 1 <= n_bins <= engine.RANK_MEAN_MAX_BINS and shape[axis] >= 1:
        # the mean of the one-hots over `avg_dim`, formed by the kernel itself
        # (the outer index is C order over `shape`)
        mean_over = (int(np.prod(shape[:axis], dtype=np.int64)), shape[axis],
                     int(np.prod(shape[axis + 1:], dtype=np.int64)))
      else:  # counts by atomic adds, divided below
        n_acc = int(np.prod(out_shape, dtype=np.int64))
        rows = np.arange(n_acc, dtype=np.int64).reshape(out_shape)
        rows = np.broadcast_to(np.expand_dims(rows, axis), shape)
        acc_row = torch.from_numpy(np.array(rows).ravel()).to(device)
      dims = tuple(d for d in dims if d != avg_dim)
    else:
      out_shape = shape
    hist = engine.rank_histogram(
        ften, member_slabs * n_point, n_member, ens_table,
        tten.reshape(-1, n_point), truth_table, geo.n_outer, n_point, n_bins,
        self._break_ties_randomly, seed, acc_row, n_acc,
        numpy_stream=numpy_stream, mean_over=mean_over)
    if acc_row is not None:
      # tensor / tensor: a true division like NumPy's mean (torch multiplies by
      # the reciprocal when the divisor is a Python scalar)
      hist = hist / torch.full((), float(shape[axis]), dtype=hist.dtype,
                               device=hist.device)
    spatial = _SPATIAL if geo.layout == plan_lib.LATLON else _SPATIAL[::-1]
    hist = hist.reshape(tuple(out_shape) + tuple(ften.shape[-2:]) + (n_bins,))
    have = tuple(dims) + tuple(spatial) + ('bins',)
    # the reference's result has the dims of its concatenated array (minus the
    # ensemble dim, `bins` last): a permuted VIEW where that order differs
    want = tuple(d for d in self._concat_dims(fvar, tvar)
                 if d not in (self.ensemble_dim, avg_dim)) + ('bins',)
    if want != have and sorted(want) == sorted(have):
      hist = hist.permute(*[have.index(d) for d in want])
      have = want
    return hist, have, n_bins

  def _dataset(self, forecast, truth, avg_dim=None):
    forecast, truth = _inputs(forecast, truth)
    _get_n_ensemble(forecast, self.ensemble_dim)
    out = xl.Dataset()
    for name in _common_vars(forecast, truth):
      hist, dims, n_bins = self._histogram(forecast, truth, name, avg_dim)
      out.coords.update(_spatial_coords(forecast, dims[:-1]))
      out.coords['bins'] = np.arange(n_bins)
      out.data_vars[name] = xl.DataArray(hist, dims, out.coords, name)
    return out

  def compute_chunk(self, forecast, truth, region=None, skipna=False):
    """One-hot encoding of rank on a chunk of forecast/truth."""
    return self._dataset(forecast, truth)

  def compute(self, forecast, truth, region=None, skipna=False):
    """Rank histogram: the temporal mean of the one-hot encoding."""
    forecast_ds = xl.as_dataset(forecast)
    avg_dim = 'time' if 'time' in forecast_ds.dims else 'init_time'
    if avg_dim not in forecast_ds.dims:
      raise ValueError(
          'Forecast has neither valid_time or init_time dimension '
          f'{forecast_ds}')
    out = self._dataset(forecast, truth, avg_dim)
    return out.assign_attrs(
        ensemble_size=forecast_ds.sizes[self.ensemble_dim])


def central_reliability(hist):
  """Reliability of central prediction intervals (metrics.py:2045-2126).

  `hist` is a DataArray/Dataset with a `bins` dim (RankHistogram's temporal
  and/or spatial mean).  Returns the cumulative probability of the intervals
  grown outward from the centre bin(s), on a new `prob_index` dim with the
  `desired_prob` coordinate.  Tiny host work on an already reduced histogram.
  """
  if xl.is_xarray(hist):
    given = hist
    if not hasattr(hist, 'data_vars'):  # an xarray.DataArray
      name = hist.name or 'hist'
      return xl.like_input(
          central_reliability(xl.as_dataset(hist.to_dataset(name=name))[name]),
          given)
    return xl.like_input(central_reliability(xl.as_dataset(hist)), given)
  hist = xl.as_dataset(hist) if not isinstance(hist, xl.DataArray) else hist
  if isinstance(hist, xl.Dataset):
    out = xl.Dataset(attrs=dict(hist.attrs))
    for name, da in hist.data_vars.items():
      r = central_reliability(da)
      out.coords.update(r.coords)
      out.data_vars[name] = r
    return out
  if 'bins' not in hist.dims:
    raise ValueError(f"hist has no 'bins' dim: {hist.dims}")
  n_bins = hist.sizes['bins']
  if n_bins < 3:
    raise ValueError(f'Too few bins. {n_bins=} but should be >= 3')
  axis = hist.dims.index('bins')
  values = hist.data
  if isinstance(values, torch.Tensor):
    values = values.cpu().numpy()
  values = np.moveaxis(np.asarray(values), axis, -1)
  left = values[..., :n_bins // 2]
  right = values[..., n_bins // 2 + n_bins % 2:]
  probs = np.cumsum(left[..., ::-1] + right, axis=-1)
  desired = np.ones(probs.shape[-1])
  if n_bins % 2:
    center = values[..., n_bins // 2][..., None]
    probs = np.concatenate([center, center + probs], axis=-1)
    desired = np.concatenate(([0.5], desired))
  desired = np.cumsum(desired)
  desired = desired / desired[-1]
  # reference dim order: the new dim replaces `bins` for even n_bins; for odd
  # n_bins the xr.concat with the centre bin puts it first (:2100-2106).  The
  # final swap_dims names it `desired_prob` (:2126).
  rest = tuple(d for d in hist.dims if d != 'bins')
  if n_bins % 2:
    probs = np.moveaxis(probs, -1, 0)
    dims = ('desired_prob',) + rest
  else:
    probs = np.moveaxis(probs, -1, axis)
    dims = tuple('desired_prob' if d == 'bins' else d for d in hist.dims)
  coords = {k: v for k, v in hist.coords.items() if k != 'bins'}
  coords['desired_prob'] = desired
  return xl.DataArray(probs, dims, coords, hist.name)
