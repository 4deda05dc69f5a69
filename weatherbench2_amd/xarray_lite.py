"""Minimal labelled arrays: the slice of the xarray API this path touches.

xarray is not installable in the build image, and the reference's operator
protocol (`Metric.compute_chunk(forecast: xr.Dataset, ...) -> xr.Dataset`,
weatherbench2/metrics.py:88-115) is expressed in it.  `Dataset`/`DataArray`
here mirror the few members that protocol and `_metric_and_region_loop`
(evaluation.py:388-438) use -- `dims`, `sizes`, `coords`, `data_vars`, item
access, `mean`, `expand_dims`, `concat`, `merge` -- so that the GPU metrics can
be driven, and tested, without xarray.  When real xarray IS importable,
`from_xarray` / `to_xarray` convert at the boundary.

`DataArray.data` may be a numpy array (host) or a torch tensor (device, e.g.
obtained through DLPack); results of scalar metrics are small host arrays.
"""
from __future__ import annotations

import typing as t

import numpy as np

try:  # optional
  import xarray as _xr  # type: ignore
except Exception:  # pragma: no cover - xarray is absent in the build image
  _xr = None


def _is_torch(x) -> bool:
  return type(x).__module__.startswith('torch')


def label_list(values) -> list:
  """Labels as hashable Python values for exact matching (`.sel` semantics).
  datetime64 / timedelta64 of any unit are brought to nanoseconds first:
  `tolist()` gives ints for [ns] but datetime objects for [s] / [h], so the
  same instant in two units would not match (pandas indexes compare them
  equal)."""
  a = np.asarray(values)
  if a.dtype.kind == 'M':
    a = a.astype('datetime64[ns]')
  elif a.dtype.kind == 'm':
    a = a.astype('timedelta64[ns]')
  return a.ravel().tolist()


class SlabGather:
  """A gather that has not happened: element [o..., r, c] of the array is
  `base[index[o...], r, c]`.

  `base` is a C-contiguous numpy array or torch tensor whose last two dims are
  one 2-D slab (n_row, n_col) and whose leading dims, flattened, number the
  slabs; `index` is an int64 numpy array over the OUTER dims of the gathered
  array, -1 marking a slab that does not exist (NaN-filled, like the holes of
  an xarray outer join).  This is how `forecast := climatology.sel(dayofyear,
  hour)` (evaluation.py:452-460), the persistence forecast (:165-193, 651-675)
  and the probabilistic climatology (utils.py:47-70) are handed to the fused
  passes without copying anything: the deterministic passes read the base
  through `index` as their slab table (metrics._physical_slabs); every other
  consumer materialises (`materialize` on the device, `__array__` on the
  host) and is merely correct."""

  def __init__(self, base, index):
    index = np.asarray(index, dtype=np.int64)
    if base.ndim < 2:
      raise ValueError('base needs (n_row, n_col) as its last two dims')
    self.base = base
    self.index = index
    self.slab_shape = tuple(int(n) for n in base.shape[-2:])
    n_slab = 1
    for n in base.shape[:-2]:
      n_slab *= int(n)
    self.n_slab = n_slab
    if index.size and (index.max() >= n_slab or index.min() < -1):
      raise IndexError(f'slab index out of range [-1, {n_slab})')

  @property
  def shape(self):
    return tuple(self.index.shape) + self.slab_shape

  @property
  def ndim(self):
    return self.index.ndim + 2

  @property
  def dtype(self):
    return self.base.dtype

  @property
  def has_missing(self) -> bool:
    return bool(self.index.size) and bool((self.index < 0).any())

  def __repr__(self):
    return (f'<wb2hip.SlabGather shape={self.shape} dtype={self.dtype} of '
            f'{self.n_slab} slabs>')

  def _flat_base(self):
    return self.base.reshape((self.n_slab,) + self.slab_shape)

  def __getitem__(self, key):
    """Indexing of the outer dims (the two slab dims must be kept whole)."""
    if not isinstance(key, tuple):
      key = (key,)
    outer = key[:self.index.ndim]
    rest = key[self.index.ndim:]
    if any(not (isinstance(k, slice) and k == slice(None)) for k in rest):
      return np.asarray(self)[key]
    return SlabGather(self.base, self.index[outer])

  def permute_outer(self, perm):
    return SlabGather(self.base, np.transpose(self.index, perm))

  def materialize_host(self) -> np.ndarray:
    flat = self._flat_base()
    if _is_torch(flat):
      return self.materialize().cpu().numpy()
    idx = self.index.ravel()
    out = np.take(flat, np.maximum(idx, 0), axis=0)
    if (idx < 0).any():
      if out.dtype.kind != 'f':
        out = out.astype(np.float64)
      out[idx < 0] = np.nan
    return out.reshape(self.shape)

  def __array__(self, dtype=None, copy=None):
    out = self.materialize_host()
    return out if dtype is None else out.astype(dtype, copy=False)

  def materialize(self, device=None):
    """The gathered array as a torch tensor (one index_select where the base
    lives; host bases gather on the host first, so only the needed slabs cross
    PCIe)."""
    import torch
    flat = self._flat_base()
    if not _is_torch(flat):
      host = self.materialize_host()
      if device is None:
        return torch.from_numpy(host)
      from weatherbench2_amd import engine
      return engine.as_device_tensor(host, torch.device(device))
    idx = torch.as_tensor(np.maximum(self.index.ravel(), 0), device=flat.device)
    out = torch.index_select(flat, 0, idx)
    missing = self.index.ravel() < 0
    if missing.any():
      if not out.dtype.is_floating_point:
        out = out.to(torch.float64)
      out[torch.as_tensor(missing, device=flat.device)] = float('nan')
    out = out.reshape(self.shape)
    return out if device is None else out.to(device)

  def compact_host(self):
    """(array of the DISTINCT slabs referenced, index into it) for a host
    base: what has to cross PCIe."""
    idx = self.index.ravel()
    uniq, inv = np.unique(idx[idx >= 0], return_inverse=True)
    new_index = np.full(idx.shape, -1, dtype=np.int64)
    new_index[idx >= 0] = inv
    small = np.take(self._flat_base(), uniq, axis=0)
    return small, new_index.reshape(self.index.shape)


class SlabConcat:
  """A concatenation that has not happened: element [o..., r, c] of the array
  is slab `index[o...]` of the VIRTUAL concatenation of `bases` along their
  flattened leading dims.

  `bases` are C-contiguous arrays (all numpy or all torch tensors on one
  device, one dtype) whose last two dims are one 2-D slab (n_row, n_col);
  `index` is an int64 numpy array over the outer dims of the result.  This is
  how `evaluation.concat_chunks` hands k consecutive (init_time=1, lead_time=1)
  chunks of the Beam pipeline (evaluation.py:693-705) to ONE fused pass without
  moving a byte: the deterministic passes turn the index into one device
  ADDRESS per slab (`addresses`, wb2_stream_partials_addr); every other
  consumer materialises (one `cat` + `index_select`) and is merely correct."""

  def __init__(self, bases, index, uniform: bool = False):
    bases = list(bases)
    if not bases:
      raise ValueError('SlabConcat needs at least one base')
    index = np.asarray(index, dtype=np.int64)
    self.slab_shape = tuple(int(n) for n in bases[0].shape[-2:])
    if uniform:  # the caller vouches: equally shaped bases of one dtype
      n = 1
      for m in bases[0].shape[:-2]:
        n *= int(m)
      counts = [n] * len(bases)
    else:
      counts = []
      for b in bases:
        if b.ndim < 2 or tuple(int(n) for n in b.shape[-2:]) != self.slab_shape:
          raise ValueError('every base needs the same (n_row, n_col) last '
                           'dims')
        if b.dtype != bases[0].dtype:
          raise ValueError('bases differ in dtype')
        n = 1
        for m in b.shape[:-2]:
          n *= int(m)
        counts.append(n)
    self.bases = bases
    self.index = index
    self.offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    self.n_slab = int(self.offsets[-1])
    if index.size and (index.max() >= self.n_slab or index.min() < 0):
      raise IndexError(f'slab index out of range [0, {self.n_slab})')

  def with_bases(self, bases) -> 'SlabConcat':
    """The same concatenation over other arrays -- as many, shaped and typed
    like this one's (the caller vouches): the index, its range check and the
    offsets are shared, not recomputed (evaluation.concat_chunks builds the
    same layout for window after window)."""
    out = SlabConcat.__new__(SlabConcat)
    out.slab_shape, out.index = self.slab_shape, self.index
    out.offsets, out.n_slab = self.offsets, self.n_slab
    out.bases = list(bases)
    return out

  @property
  def shape(self):
    return tuple(self.index.shape) + self.slab_shape

  @property
  def ndim(self):
    return self.index.ndim + 2

  @property
  def dtype(self):
    return self.bases[0].dtype

  has_missing = False

  @property
  def on_device(self) -> bool:
    return _is_torch(self.bases[0])

  def __repr__(self):
    return (f'<wb2hip.SlabConcat shape={self.shape} dtype={self.dtype} of '
            f'{len(self.bases)} arrays>')

  def __getitem__(self, key):
    """Indexing of the outer dims (the two slab dims must be kept whole)."""
    if not isinstance(key, tuple):
      key = (key,)
    outer = key[:self.index.ndim]
    rest = key[self.index.ndim:]
    if any(not (isinstance(k, slice) and k == slice(None)) for k in rest):
      return np.asarray(self)[key]
    return SlabConcat(self.bases, self.index[outer])

  def permute_outer(self, perm):
    return SlabConcat(self.bases, np.transpose(self.index, perm))

  def addresses(self) -> np.ndarray:
    """int64 array of index.shape: the device address of every slab (bases
    must be torch tensors)."""
    item = self.bases[0].element_size()
    step = self.slab_shape[0] * self.slab_shape[1] * item
    first = np.array([b.data_ptr() for b in self.bases], dtype=np.int64)
    which = np.searchsorted(self.offsets, self.index, side='right') - 1
    return first[which] + (self.index - self.offsets[which]) * step

  def materialize(self, device=None):
    import torch
    if not self.on_device:
      host = self.materialize_host()
      if device is None:
        return torch.from_numpy(host)
      from weatherbench2_amd import engine
      return engine.as_device_tensor(host, torch.device(device))
    flat = torch.cat([b.reshape((-1,) + self.slab_shape) for b in self.bases])
    idx = torch.as_tensor(self.index.ravel(), device=flat.device)
    out = torch.index_select(flat, 0, idx).reshape(self.shape)
    return out if device is None else out.to(device)

  def materialize_host(self) -> np.ndarray:
    if self.on_device:
      return self.materialize().cpu().numpy()
    flat = np.concatenate([np.asarray(b).reshape((-1,) + self.slab_shape)
                           for b in self.bases])
    return np.take(flat, self.index.ravel(), axis=0).reshape(self.shape)

  def __array__(self, dtype=None, copy=None):
    out = self.materialize_host()
    return out if dtype is None else out.astype(dtype, copy=False)


_LAZY = (SlabGather, SlabConcat)


class DataArray:
  """N-d array with named dims and (1-D, per-dim) coordinates."""

  def __init__(self, data, dims: t.Sequence[str] = (), coords=None, name=None):
    if not _is_torch(data) and not isinstance(data, _LAZY):
      data = np.asarray(data)
    self.data = data
    self.dims = tuple(dims)
    if len(self.dims) != data.ndim:
      raise ValueError(f'dims {self.dims} do not match shape {tuple(data.shape)}')
    self.coords = dict(coords or {})
    self.name = name

  @property
  def shape(self):
    return tuple(self.data.shape)

  @property
  def ndim(self):
    return self.data.ndim

  @property
  def sizes(self):
    return dict(zip(self.dims, self.shape))

  @property
  def dtype(self):
    return self.data.dtype

  @property
  def values(self) -> np.ndarray:
    if _is_torch(self.data):
      # a result produced on another thread's stream has been published to the
      # default stream; a reader on a private stream waits for that first
      from weatherbench2_amd import engine
      engine.order_read(self.data)
      from weatherbench2_amd import feeder
      return feeder.download(self.data.detach())
    if isinstance(self.data, _LAZY):
      return self.data.materialize_host()
    return self.data

  def copy(self, data=None):
    return DataArray(self.data if data is None else data, self.dims,
                     self.coords, self.name)

  def __repr__(self):
    return (f'<wb2hip.DataArray {self.name or ""} dims={self.dims} '
            f'shape={self.shape} dtype={self.dtype}>')

  def _host(self) -> 'DataArray':
    return DataArray(self.values, self.dims, self.coords, self.name)

  def transpose(self, *dims):
    perm = [self.dims.index(d) for d in dims]
    if isinstance(self.data, _LAZY):
      n = self.data.index.ndim
      if perm[n:] == [n, n + 1]:  # the slab dims stay where they are
        data = self.data.permute_outer(perm[:n])
      else:
        data = np.transpose(self.data.materialize_host(), perm)
    elif _is_torch(self.data):
      data = self.data.permute(*perm)
    else:
      data = np.transpose(self.data, perm)
    return DataArray(data, dims, self.coords, self.name)

  def isel(self, **indexers):
    data, dims, coords = self.data, list(self.dims), dict(self.coords)
    for dim, idx in indexers.items():
      if dim not in dims:
        continue
      ax = dims.index(dim)
      sl = [slice(None)] * data.ndim
      sl[ax] = idx
      data = data[tuple(sl)]
      if np.ndim(idx) == 0 and not isinstance(idx, slice):
        dims.pop(ax)
        coords.pop(dim, None)
      elif dim in coords:
        coords[dim] = np.asarray(coords[dim])[idx]
    return DataArray(data, dims, coords, self.name)

  def expand_dims(self, dim: t.Union[str, dict], axis: int = 0):
    """`dim` is a name or {name: labels}; new dims go first (xarray)."""
    if isinstance(dim, str):
      dim = {dim: None}
    out = self
    for name, labels in reversed(list(dim.items())):
      n = 1 if labels is None else len(np.atleast_1d(labels))
      if _is_torch(out.data):  # device-resident maps stay on the device
        data = out.data.unsqueeze(0)
        if n != 1:
          data = data.expand((n,) + out.shape).contiguous()
      else:
        data = out.values[None] if n == 1 else np.broadcast_to(
            out.values[None], (n,) + out.shape).copy()
      coords = dict(out.coords)
      if labels is not None:
        coords[name] = np.atleast_1d(labels)
      out = DataArray(data, (name,) + out.dims, coords, out.name)
    return out

  def mean(self, dim=None, skipna: bool = False):
    a = self.values
    if dim is None:
      axes, keep = tuple(range(a.ndim)), ()
    else:
      dim = (dim,) if isinstance(dim, str) else tuple(dim)
      axes = tuple(self.dims.index(d) for d in dim if d in self.dims)
      keep = tuple(d for d in self.dims if d not in dim)
    if not axes:
      return self
    import warnings
    with warnings.catch_warnings(), np.errstate(all='ignore'):
      warnings.simplefilter('ignore')
      out = (np.nanmean if skipna else np.mean)(a, axis=axes)
    coords = {k: v for k, v in self.coords.items() if k in keep or
              (k not in self.dims)}
    return DataArray(out, keep, coords, self.name)

  # small host-side arithmetic (results are tiny)
  def _bin(self, other, fn):
    o = other.values if isinstance(other, DataArray) else other
    if isinstance(other, DataArray) and other.dims != self.dims:
      raise ValueError(f'dims differ: {self.dims} vs {other.dims}')
    with np.errstate(all='ignore'):
      return DataArray(fn(self.values, o), self.dims, self.coords, self.name)

  def __add__(self, o): return self._bin(o, np.add)
  def __sub__(self, o): return self._bin(o, np.subtract)
  def __mul__(self, o): return self._bin(o, np.multiply)
  def __rmul__(self, o): return self._bin(o, np.multiply)
  def __truediv__(self, o): return self._bin(o, np.true_divide)

  def sqrt(self):
    with np.errstate(all='ignore'):
      return DataArray(np.sqrt(self.values), self.dims, self.coords, self.name)


class Dataset:
  """Dict of DataArrays sharing coordinates."""

  def __init__(self, data_vars=None, coords=None, attrs=None):
    self.coords = dict(coords or {})
    self.data_vars: dict[str, DataArray] = {}
    self.attrs = dict(attrs or {})
    for k, v in (data_vars or {}).items():
      self[k] = v

  def __setitem__(self, name, value):
    if isinstance(value, tuple):
      value = DataArray(value[1], value[0])
    if is_xarray(value):  # an xarray.DataArray (a foreign metric's result)
      value = DataArray(value.data, tuple(value.dims))
    if not isinstance(value, DataArray):
      value = DataArray(value, ())
    for k, c in value.coords.items():
      self.coords.setdefault(k, c)
    value = DataArray(value.data, value.dims,
                      {k: v for k, v in self.coords.items()}, name)
    self.data_vars[name] = value

  def __getitem__(self, name):
    if name in self.data_vars:
      return self.data_vars[name]
    if name in self.coords:
      c = self.coords[name]
      return c if isinstance(c, DataArray) else DataArray(c, (name,),
                                                          {name: c}, name)
    raise KeyError(name)

  def __contains__(self, name): return name in self.data_vars
  def __iter__(self): return iter(self.data_vars)
  def __len__(self): return len(self.data_vars)
  def keys(self): return self.data_vars.keys()
  def items(self): return self.data_vars.items()

  @property
  def dims(self):
    out = {}
    for v in self.data_vars.values():
      out.update(v.sizes)
    return out

  sizes = dims

  def has_dim(self, name) -> bool:
    """`name in self.dims` without building the sizes of every variable."""
    for v in self.data_vars.values():
      if name in v.dims:
        return True
    return False

  def __repr__(self):
    return (f'<wb2hip.Dataset vars={list(self.data_vars)} dims={self.dims}>')

  def map(self, fn):
    return Dataset({k: fn(v) for k, v in self.data_vars.items()}, self.coords,
                   self.attrs)

  def copy(self, data=None):
    if data is None:
      return Dataset(dict(self.data_vars), self.coords, self.attrs)
    return Dataset({k: v.copy(data[k]) for k, v in self.data_vars.items()},
                   self.coords, self.attrs)

  def isel(self, **indexers):
    coords = dict(self.coords)
    for dim, idx in indexers.items():
      if dim in coords and not isinstance(coords[dim], DataArray):
        if np.ndim(idx) == 0 and not isinstance(idx, slice):
          coords.pop(dim)
        else:
          coords[dim] = np.asarray(coords[dim])[idx]
    # non-index coordinates over several dims (valid_time(init_time, lead), a
    # 2-D `time` of by-init truth) are indexed along the same dims as the data
    for name, c in list(coords.items()):
      if isinstance(c, DataArray) and any(d in c.dims for d in indexers):
        sub = c.isel(**{d: i for d, i in indexers.items() if d in c.dims})
        coords[name] = DataArray(sub.data, sub.dims, {}, name)
    out = Dataset(coords=coords, attrs=self.attrs)
    for k, v in self.data_vars.items():
      w = v.isel(**indexers)
      out.data_vars[k] = DataArray(w.data, w.dims, coords, k)
    return out

  def mean(self, dim=None, skipna: bool = False):
    dims = (dim,) if isinstance(dim, str) else tuple(dim or ())
    # like xarray: coordinates that vary along a reduced dim go with it
    coords = {k: v for k, v in self.coords.items()
              if k not in dims and not (isinstance(v, DataArray)
                                        and any(d in v.dims for d in dims))}
    out = Dataset(coords=coords, attrs=self.attrs)
    for k, v in self.data_vars.items():
      m = v.mean(dim, skipna=skipna)
      out.data_vars[k] = DataArray(m.data, m.dims, coords, k)
    return out

  def expand_dims(self, dim):
    if isinstance(dim, str):
      dim = {dim: None}
    coords = dict(self.coords)
    for name, labels in dim.items():
      if isinstance(labels, DataArray):
        labels = labels.values
        dim = {**dim, name: labels}
      if labels is not None:
        coords[name] = np.atleast_1d(labels)
    out = Dataset(coords=coords, attrs=self.attrs)
    for k, v in self.data_vars.items():
      w = v.expand_dims(dim)
      out.data_vars[k] = DataArray(w.data, w.dims, coords, k)
    return out

  def assign_attrs(self, **attrs):
    return Dataset(dict(self.data_vars), self.coords, {**self.attrs, **attrs})

  def _bin(self, other, fn):
    if isinstance(other, Dataset):
      names = [k for k in self.data_vars if k in other.data_vars]
      return Dataset({k: fn(self.data_vars[k], other.data_vars[k])
                      for k in names}, self.coords, self.attrs)
    return self.map(lambda v: fn(v, other))

  def __add__(self, o): return self._bin(o, lambda a, b: a + b)
  def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
  def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
  def __rmul__(self, o): return self._bin(o, lambda a, b: a * b)
  def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)

  def sqrt(self): return self.map(lambda v: v.sqrt())


def concat(datasets: t.Sequence[Dataset], dim: str) -> Dataset:
  """xr.concat along an existing length-1 dim (evaluation.py:430)."""
  first = datasets[0]
  coords = dict(first.coords)
  coords[dim] = np.concatenate([np.atleast_1d(d.coords[dim]) for d in datasets])
  out = Dataset(coords=coords, attrs=first.attrs)
  for k, v in first.data_vars.items():
    ax = v.dims.index(dim)
    parts = [d.data_vars[k] for d in datasets]
    if all(_is_torch(x.data) for x in parts):
      import torch
      data = torch.cat([x.data for x in parts], dim=ax)
    else:
      data = np.concatenate([x.values for x in parts], axis=ax)
    out.data_vars[k] = DataArray(data, v.dims, coords, k)
  return out


def merge_metrics(named: t.Sequence[tuple]) -> Dataset:
  """`merge([ds.expand_dims({'metric': [name]}) for name, ds in named])`
  (evaluation.py:424-437) in one step: every variable of the result is ONE
  stack of the per-metric arrays along a new leading `metric` dim (NaN-filled
  where a metric lacks the variable), labels in the order xarray's outer join
  gives them.  No per-metric unsqueeze / select views on the way."""
  names = [n for n, _ in named]
  if len(set(names)) != len(names):
    return merge([d.expand_dims({'metric': [n]}) for n, d in named])
  labels = list(names)
  if len(labels) > 1:
    # xarray's outer join is pandas' Index.union: sorted (see `merge`)
    try:
      labels = sorted(labels)
    except TypeError:
      pass
  by_label = dict(named)
  variables: list = []
  for _, d in named:
    variables += [k for k in d.data_vars if k not in variables]
  coords = {}
  for _, d in named:
    for k, v in d.coords.items():
      coords.setdefault(k, v)
  coords['metric'] = np.array(labels, dtype=object)
  out = Dataset(coords=coords)
  as_np = lambda dt: np.dtype(str(dt).replace('torch.', ''))
  for var in variables:
    holders = [by_label[m].data_vars.get(var) for m in labels]
    have = [h for h in holders if h is not None]
    # dim order: the first dataset IN THE LIST that holds the variable (xarray)
    ref = next(d.data_vars[var] for _, d in named if var in d.data_vars)
    dtype = np.result_type(*[as_np(h.dtype) for h in have])
    if dtype.kind != 'f':
      dtype = np.dtype(np.float64)
    on_device = all(_is_torch(h.data) for h in have)
    rows = []
    for h in holders:
      if h is None:
        rows.append(None)
        continue
      if h.dims != ref.dims:  # xarray aligns by name: same dims, other order
        if sorted(h.dims) != sorted(ref.dims):
          raise ValueError(f'{var}: cannot merge dims {h.dims} with {ref.dims}')
        h = h.transpose(*ref.dims)
      rows.append(h.data if on_device else h.values)
    if on_device:
      import torch
      tdtype = getattr(torch, dtype.name)
      if any(r is None for r in rows):
        hole = torch.full(ref.shape, float('nan'), dtype=tdtype,
                          device=ref.data.device)
        rows = [hole if r is None else r for r in rows]
      data = torch.stack([r if r.dtype == tdtype else r.to(tdtype)
                          for r in rows])
    else:
      data = np.full((len(labels),) + ref.shape, np.nan, dtype=dtype)
      for i, r in enumerate(rows):
        if r is not None:
          data[i] = r
    out.data_vars[var] = DataArray(data, ('metric',) + ref.dims, coords, var)
  return out


def merge(datasets: t.Sequence[Dataset]) -> Dataset:
  """xr.merge of results that differ along `metric` (evaluation.py:437).

  Variables missing from one operand are NaN-filled, like xarray's outer join.
  """
  labels: list = []
  per_dataset = [list(np.atleast_1d(d.coords['metric'])) for d in datasets]
  for ms in per_dataset:
    for m in ms:
      if m not in labels:
        labels.append(m)
  if any(ms != per_dataset[0] for ms in per_dataset):
    # xarray's outer join is pandas' Index.union, which SORTS its result unless
    # the indexes are equal: the merged `metric` coordinate of the reference
    # comes out alphabetical, not in the order of Eval.metrics
    try:
      labels = sorted(labels)
    except TypeError:  # labels that cannot be compared stay in order (pandas)
      pass
  names: list = []
  for d in datasets:
    names += [k for k in d.data_vars if k not in names]
  coords = {}
  for d in datasets:
    for k, v in d.coords.items():
      coords.setdefault(k, v)
  coords['metric'] = np.array(labels, dtype=object)
  out = Dataset(coords=coords)
  for name in names:
    ref = next(d.data_vars[name] for d in datasets if name in d.data_vars)
    ax = ref.dims.index('metric')
    shape = list(ref.shape)
    shape[ax] = len(labels)
    holders = [d.data_vars[name] for d in datasets if name in d.data_vars]
    on_device = all(_is_torch(v.data) for v in holders)
    # NumPy promotion over the operands (float32 results stay float32 when
    # every metric returned float32, like xarray's merge)
    as_np = lambda dt: np.dtype(str(dt).replace('torch.', ''))
    dtype = np.result_type(*[as_np(v.dtype) for v in holders])
    if dtype.kind != 'f':
      dtype = np.dtype(np.float64)  # the NaN fill needs a float
    if on_device:
      # results that live on the device are merged there, with ONE stack per
      # variable: the per-metric slices are views into the fused passes' output
      import torch
      tdtype = getattr(torch, dtype.name)
      rows: list = [None] * len(labels)
      for d in datasets:
        if name not in d.data_vars:
          continue
        v = d.data_vars[name]
        if v.dims != ref.dims:  # xarray aligns by name: same dims, other order
          if sorted(v.dims) != sorted(ref.dims):
            raise ValueError(
                f'{name}: cannot merge dims {v.dims} with {ref.dims}')
          v = v.transpose(*ref.dims)
        for i, m in enumerate(np.atleast_1d(d.coords['metric'])):
          rows[labels.index(m)] = v.data.select(ax, i).to(tdtype)
      if any(r is None for r in rows):
        shape.pop(ax)
        hole = torch.full(shape, float('nan'), dtype=tdtype,
                          device=ref.data.device)
        rows = [hole if r is None else r for r in rows]
      out.data_vars[name] = DataArray(torch.stack(rows, dim=ax), ref.dims,
                                      coords, name)
      continue
    data = np.full(shape, np.nan, dtype=dtype)
    for d in datasets:
      if name not in d.data_vars:
        continue
      v = d.data_vars[name]
      if v.dims != ref.dims:  # xarray aligns by name: same dims, other order
        if sorted(v.dims) != sorted(ref.dims):
          raise ValueError(f'{name}: cannot merge dims {v.dims} with {ref.dims}')
        v = v.transpose(*ref.dims)
      values = v.values
      for i, m in enumerate(np.atleast_1d(d.coords['metric'])):
        sl = [slice(None)] * len(shape)
        sl[ax] = labels.index(m)
        src = [slice(None)] * len(shape)
        src[ax] = i
        data[tuple(sl)] = values[tuple(src)]
    out.data_vars[name] = DataArray(data, ref.dims, coords, name)
  return out


def _index_or_slice(idx: np.ndarray):
  """A contiguous ascending run becomes a slice (a view, no copy)."""
  if len(idx) and np.array_equal(idx, np.arange(idx[0], idx[0] + len(idx))):
    return slice(int(idx[0]), int(idx[0]) + len(idx))
  return idx


def align_inner(a: Dataset, b: Dataset, exclude=()) -> tuple:
  """xarray's default arithmetic join for `a (op) b`: every dimension
  coordinate the two share keeps the labels present in both (left order).
  Returns the inputs themselves when nothing needs aligning."""
  sel_a, sel_b = {}, {}
  for d, ca in a.coords.items():
    cb = b.coords.get(d)
    if (cb is None or d in exclude or isinstance(ca, DataArray)
        or isinstance(cb, DataArray) or d not in a.dims or d not in b.dims):
      continue
    ca, cb = np.asarray(ca), np.asarray(cb)
    if ca.ndim != 1 or cb.ndim != 1:
      continue
    if ca.shape == cb.shape and np.array_equal(ca, cb):
      continue
    pos_b = {v: i for i, v in enumerate(label_list(cb))}
    keep = [(i, pos_b[v]) for i, v in enumerate(label_list(ca)) if v in pos_b]
    sel_a[d] = _index_or_slice(np.array([i for i, _ in keep], dtype=np.int64))
    sel_b[d] = _index_or_slice(np.array([j for _, j in keep], dtype=np.int64))
  if not sel_a:
    return a, b
  return a.isel(**sel_a), b.isel(**sel_b)


# -- optional bridges to real xarray -----------------------------------------
def from_xarray(ds) -> Dataset:
  coords = {}
  for k, c in ds.coords.items():
    coords[k] = (np.asarray(c.values) if c.ndim <= 1 and c.dims == (k,)
                 else DataArray(np.asarray(c.values), c.dims))
  return Dataset({k: DataArray(v.data, v.dims) for k, v in ds.data_vars.items()},
                 coords, dict(ds.attrs))


def to_xarray(ds: Dataset):
  if _xr is None:
    raise ImportError('xarray is not installed')
  coords = {}
  for k, c in ds.coords.items():
    if isinstance(c, DataArray):
      coords[k] = (c.dims, c.values)
    elif any(k in v.dims for v in ds.data_vars.values()):
      coords[k] = (k, np.asarray(c))
  return _xr.Dataset({k: (v.dims, v.values) for k, v in ds.data_vars.items()},
                     coords=coords, attrs=ds.attrs)


def is_xarray(obj) -> bool:
  """True for a real xarray.Dataset / DataArray (never without xarray)."""
  return _xr is not None and isinstance(obj, (_xr.Dataset, _xr.DataArray))


def as_dataset(obj) -> Dataset:
  """Accepts a lite Dataset or (when importable) a real xarray.Dataset."""
  if isinstance(obj, Dataset):
    return obj
  if _xr is not None and isinstance(obj, _xr.Dataset):
    return from_xarray(obj)
  raise TypeError(f'expected a Dataset, got {type(obj)}')


def like_input(result, *inputs):
  """The protocol's return convention: callers that hand in xarray objects
  (the reference's own `_metric_and_region_loop`, evaluation.py:408-435, goes
  on to call `.expand_dims` / `xr.concat` on what it gets back) receive xarray
  objects; lite in, lite out."""
  if not any(is_xarray(x) for x in inputs):
    return result
  if isinstance(result, Dataset):
    return to_xarray(result)
  if isinstance(result, DataArray):
    coords = {}
    for k, c in result.coords.items():
      if isinstance(c, DataArray):
        if all(d in result.dims for d in c.dims):
          coords[k] = (c.dims, c.values)
      elif k in result.dims:
        coords[k] = (k, np.asarray(c))
    return _xr.DataArray(result.values, dims=result.dims, coords=coords,
                         name=result.name)
  return result
