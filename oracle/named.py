"""Tiny named-dimension arrays: just enough xarray semantics for the oracle.

TEST INFRASTRUCTURE (see oracle/__init__.py).  xarray is not installed here,
so the oracle carries its own ~200-line stand-in.  Semantics restated
(SURVEY.md Appendix A.1/A.2):

* binary ops broadcast BY DIMENSION NAME; result dims = dims of the left
  operand followed by the new dims of the right one (xarray's
  first-appearance order);
* dtype follows NumPy promotion (float32 stays float32 under python scalars);
* ``mean/var/std(dim, skipna, ddof)`` reduce with ``np.mean``/``np.var`` when
  ``skipna`` is false and the ``nan*`` variants otherwise;
* a ``DS`` (dataset) is a dict of named arrays plus 1-D coordinate arrays;
  ``DS (op) DS`` keeps the variables present in both.
"""
from __future__ import annotations

import numpy as np


class NA:
  """A numpy array with named dimensions."""

  __array_priority__ = 100

  def __init__(self, data, dims):
    self.data = np.asarray(data)
    self.dims = tuple(dims)
    if self.data.ndim != len(self.dims):
      raise ValueError(f'{self.data.shape=} does not match {self.dims=}')

  # -- basics ---------------------------------------------------------------
  @property
  def shape(self):
    return self.data.shape

  @property
  def dtype(self):
    return self.data.dtype

  @property
  def sizes(self):
    return dict(zip(self.dims, self.data.shape))

  @property
  def values(self):
    return self.data

  def copy(self, data=None):
    return NA(self.data.copy() if data is None else data, self.dims)

  def __repr__(self):
    return f'NA(dims={self.dims}, shape={self.shape}, dtype={self.dtype})'

  def transpose(self, *dims):
    perm = [self.dims.index(d) for d in dims]
    return NA(np.transpose(self.data, perm), dims)

  def expand_dims(self, dim, size=1, axis=0):
    data = np.expand_dims(self.data, axis)
    if size != 1:
      data = np.repeat(data, size, axis=axis)
    dims = list(self.dims)
    dims.insert(axis, dim)
    return NA(data, dims)

  def isel(self, drop=True, **indexers):
    data, dims = self.data, list(self.dims)
    for dim, idx in indexers.items():
      if dim not in dims:
        continue
      ax = dims.index(dim)
      if np.ndim(idx) == 0 and not isinstance(idx, slice):
        data = np.take(data, idx, axis=ax)
        dims.pop(ax)
      elif isinstance(idx, slice):
        sl = [slice(None)] * data.ndim
        sl[ax] = idx
        data = data[tuple(sl)]
      else:
        data = np.take(data, np.asarray(idx), axis=ax)
    return NA(data, dims)

  # -- broadcasting by name ---------------------------------------------------
  @staticmethod
  def _align(a, b):
    """Returns (a_data, b_data, dims) broadcast-compatible by name."""
    if not isinstance(b, NA):
      return a.data, b, a.dims
    if not isinstance(a, NA):
      return a, b.data, b.dims
    dims = list(a.dims) + [d for d in b.dims if d not in a.dims]

    def lift(x):
      perm_dims = [d for d in dims if d in x.dims]
      data = np.transpose(x.data, [x.dims.index(d) for d in perm_dims])
      shape = [x.sizes[d] if d in x.dims else 1 for d in dims]
      return data.reshape(shape)

    return lift(a), lift(b), tuple(dims)

  def _binary(self, other, fn, reflected=False):
    if isinstance(other, DS):
      return NotImplemented
    a, b, dims = NA._align(self, other)
    with np.errstate(all='ignore'):
      out = fn(b, a) if reflected else fn(a, b)
    return NA(out, dims)

  def __add__(self, o): return self._binary(o, np.add)
  def __radd__(self, o): return self._binary(o, np.add, True)
  def __sub__(self, o): return self._binary(o, np.subtract)
  def __rsub__(self, o): return self._binary(o, np.subtract, True)
  def __mul__(self, o): return self._binary(o, np.multiply)
  def __rmul__(self, o): return self._binary(o, np.multiply, True)
  def __truediv__(self, o): return self._binary(o, np.true_divide)
  def __rtruediv__(self, o): return self._binary(o, np.true_divide, True)
  def __pow__(self, o): return self._binary(o, np.power)
  def __gt__(self, o): return self._binary(o, np.greater)
  def __ge__(self, o): return self._binary(o, np.greater_equal)
  def __lt__(self, o): return self._binary(o, np.less)
  def __le__(self, o): return self._binary(o, np.less_equal)
  def __neg__(self): return NA(-self.data, self.dims)
  def __abs__(self): return NA(np.abs(self.data), self.dims)

  def sqrt(self):
    with np.errstate(all='ignore'):
      return NA(np.sqrt(self.data), self.dims)

  def astype(self, dtype):
    return NA(self.data.astype(dtype), self.dims)

  def notnull(self):
    return NA(~np.isnan(self.data), self.dims)

  def fillna(self, value):
    return NA(np.where(np.isnan(self.data), value, self.data).astype(
        self.dtype, copy=False), self.dims)

  def where(self, cond, other):
    """Keep values where `cond`, else `other` (xarray .where semantics)."""
    a, c, dims = NA._align(self, cond)
    out = np.where(c, a, np.asarray(other, dtype=self.dtype))
    return NA(out, dims)

  # -- reductions --------------------------------------------------------------
  def _axes(self, dim):
    if dim is None:
      return tuple(range(self.data.ndim)), ()
    dim = (dim,) if isinstance(dim, str) else tuple(dim)
    axes = tuple(self.dims.index(d) for d in dim)
    keep = tuple(d for d in self.dims if d not in dim)
    return axes, keep

  def mean(self, dim=None, skipna=False):
    axes, keep = self._axes(dim)
    fn = np.nanmean if skipna else np.mean
    with np.errstate(all='ignore'):
      import warnings
      with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return NA(fn(self.data, axis=axes), keep)

  def var(self, dim=None, skipna=False, ddof=0):
    axes, keep = self._axes(dim)
    fn = np.nanvar if skipna else np.var
    with np.errstate(all='ignore'):
      import warnings
      with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return NA(fn(self.data, axis=axes, ddof=ddof), keep)

  def std(self, dim=None, skipna=False, ddof=0):
    axes, keep = self._axes(dim)
    fn = np.nanstd if skipna else np.std
    with np.errstate(all='ignore'):
      import warnings
      with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return NA(fn(self.data, axis=axes, ddof=ddof), keep)

  def sum(self, dim=None, skipna=None):
    """xarray's `sum`: with the default skipna=None missing values are SKIPPED
    for float dtypes (an all-NaN slice sums to 0) -- what the reference's
    `result.sum("quantile")` does (metrics.py:1158, 1868, 1891)."""
    axes, keep = self._axes(dim)
    skip = self.data.dtype.kind in 'cf' if skipna is None else skipna
    return NA((np.nansum if skip else np.sum)(self.data, axis=axes), keep)


class DS:
  """Dict of NA variables + 1-D coordinates keyed by dimension name."""

  def __init__(self, data_vars=None, coords=None):
    self.vars = {}
    for k, v in (data_vars or {}).items():
      self.vars[k] = v if isinstance(v, NA) else NA(v[1], v[0])
    # 1-D index coordinates are plain arrays; multi-dim ones (valid_time) stay NA
    self.coords = {k: (v if isinstance(v, NA) else np.asarray(v))
                   for k, v in (coords or {}).items()}

  # mapping-ish
  def keys(self): return self.vars.keys()
  def items(self): return self.vars.items()
  def __iter__(self): return iter(self.vars)
  def __getitem__(self, k): return self.vars[k]
  def __setitem__(self, k, v): self.vars[k] = v
  def __contains__(self, k): return k in self.vars
  def __len__(self): return len(self.vars)

  @property
  def dims(self):
    out = {}
    for v in self.vars.values():
      out.update(v.sizes)
    return out

  sizes = dims

  def coord(self, name):
    return self.coords[name]

  def map(self, fn):
    return DS({k: fn(v) for k, v in self.vars.items()}, self.coords)

  def copy(self, data=None):
    if data is None:
      return DS({k: v.copy() for k, v in self.vars.items()}, self.coords)
    return DS({k: NA(data[k], v.dims) for k, v in self.vars.items()},
              self.coords)

  def rename_vars(self, mapping):
    return DS({mapping.get(k, k): v for k, v in self.vars.items()},
              self.coords)

  def select_vars(self, names):
    return DS({k: self.vars[k] for k in names}, self.coords)

  def isel(self, **indexers):
    coords = dict(self.coords)
    for dim, idx in indexers.items():
      if dim in coords and not isinstance(coords[dim], NA):
        if np.ndim(idx) == 0 and not isinstance(idx, slice):
          coords.pop(dim)
        else:
          coords[dim] = coords[dim][idx]
    for k, c in list(coords.items()):
      if isinstance(c, NA):
        coords[k] = c.isel(**indexers)
    return DS({k: v.isel(**indexers) for k, v in self.vars.items()}, coords)

  def expand_dims(self, dim, size=1, coord=None):
    coords = dict(self.coords)
    if coord is not None:
      coords[dim] = np.asarray(coord)
      size = len(coords[dim])
    return DS({k: v.expand_dims(dim, size) for k, v in self.vars.items()},
              coords)

  def _binary(self, other, op):
    if isinstance(other, DS):
      a, b = align_inner(self, other)
      names = [k for k in a.vars if k in b.vars]
      coords = {**b.coords, **a.coords}
      return DS({k: op(a.vars[k], b.vars[k]) for k in names}, coords)
    return DS({k: op(v, other) for k, v in self.vars.items()}, self.coords)

  def __add__(self, o): return self._binary(o, lambda a, b: a + b)
  def __radd__(self, o): return self._binary(o, lambda a, b: b + a)
  def __sub__(self, o): return self._binary(o, lambda a, b: a - b)
  def __rsub__(self, o): return self._binary(o, lambda a, b: b - a)
  def __mul__(self, o): return self._binary(o, lambda a, b: a * b)
  def __rmul__(self, o): return self._binary(o, lambda a, b: b * a)
  def __truediv__(self, o): return self._binary(o, lambda a, b: a / b)
  def __pow__(self, o): return self._binary(o, lambda a, b: a ** b)
  def __abs__(self): return self.map(abs)
  def __neg__(self): return self.map(lambda v: -v)

  def sqrt(self): return self.map(lambda v: v.sqrt())

  def mean(self, dim=None, skipna=False):
    def f(v):
      d = [x for x in ((dim,) if isinstance(dim, str) else dim) if x in v.dims]
      return v.mean(d, skipna=skipna) if d else v
    coords = {k: c for k, c in self.coords.items()
              if k not in ((dim,) if isinstance(dim, str) else dim)}
    return DS({k: f(v) for k, v in self.vars.items()}, coords)

  def var(self, dim, skipna=False, ddof=0):
    coords = {k: c for k, c in self.coords.items() if k != dim}
    return DS({k: v.var(dim, skipna=skipna, ddof=ddof)
               for k, v in self.vars.items()}, coords)

  def std(self, dim, skipna=False, ddof=0):
    coords = {k: c for k, c in self.coords.items() if k != dim}
    return DS({k: v.std(dim, skipna=skipna, ddof=ddof)
               for k, v in self.vars.items()}, coords)

  def zeros_like(self):
    return self.map(lambda v: NA(np.zeros_like(v.data), v.dims))

  def __repr__(self):
    return f'DS(vars={self.vars}, coords={list(self.coords)})'


def align_inner(a: DS, b: DS):
  """xarray's default arithmetic join: for every dimension coordinate the two
  operands share, keep the labels present in both (left order)."""
  sel_a, sel_b = {}, {}
  for d, ca in a.coords.items():
    cb = b.coords.get(d)
    if cb is None or isinstance(ca, NA) or isinstance(cb, NA):
      continue
    ca, cb = np.asarray(ca), np.asarray(cb)
    if ca.ndim != 1 or cb.ndim != 1 or d not in a.dims or d not in b.dims:
      continue
    if ca.shape == cb.shape and np.array_equal(ca, cb):
      continue
    pos_b = {v: i for i, v in enumerate(cb.tolist())}
    keep = [(i, pos_b[v]) for i, v in enumerate(ca.tolist()) if v in pos_b]
    sel_a[d] = np.array([i for i, _ in keep], dtype=int)
    sel_b[d] = np.array([j for _, j in keep], dtype=int)
  if sel_a:
    a, b = a.isel(**sel_a), b.isel(**sel_b)
  return a, b
