"""xarray.testing subset (assert_equal / assert_identical / assert_allclose)."""
import numpy as np


def _fmt(a, b):
  return f'\nL: {a!r}\nR: {b!r}'


def _check_coords(a, b, rtol, atol, exact):
  assert set(a._coords) == set(b._coords), (
      f'coordinates differ: {sorted(map(str, a._coords))} vs '
      f'{sorted(map(str, b._coords))}')
  for k in a._coords:
    ca, cb = a._coords[k], b._coords[k]
    assert ca.dims == cb.dims, f'coordinate {k!r}: dims {ca.dims} vs {cb.dims}'
    if exact or ca.data.dtype.kind not in 'fc':
      assert ca.equals(cb), f'coordinate {k!r} differs'
    else:
      np.testing.assert_allclose(ca.data, cb.data, rtol=rtol, atol=atol,
                                 err_msg=f'coordinate {k!r}')


def _arrays(a, b, rtol, atol, exact, what):
  from . import _array_equiv
  assert a.shape == b.shape, f'{what}: shapes {a.shape} vs {b.shape}'
  if exact:
    assert _array_equiv(a, b), f'{what}: values differ{_fmt(a, b)}'
  elif a.dtype.kind in 'fciub' and b.dtype.kind in 'fciub':
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, equal_nan=True,
                               err_msg=what)
  else:
    assert _array_equiv(a, b), f'{what}: values differ'


def _compare(a, b, rtol, atol, exact, check_names=False):
  from . import DataArray, Dataset
  assert type(a) is type(b), f'{type(a).__name__} vs {type(b).__name__}'
  if isinstance(a, DataArray):
    assert a.dims == b.dims, f'dims {a.dims} vs {b.dims}'
    if check_names:
      assert a.name == b.name, f'names {a.name!r} vs {b.name!r}'
    _arrays(a.data, b.data, rtol, atol, exact, f'DataArray {a.name!r}')
    _check_coords(a, b, rtol, atol, exact)
  elif isinstance(a, Dataset):
    assert set(a._vars) == set(b._vars), (
        f'data variables differ: {sorted(a._vars)} vs {sorted(b._vars)}')
    for k in a._vars:
      (d1, x1, _), (d2, x2, _) = a._vars[k], b._vars[k]
      assert d1 == d2, f'variable {k!r}: dims {d1} vs {d2}'
      _arrays(x1, x2, rtol, atol, exact, f'variable {k!r}')
    _check_coords(a, b, rtol, atol, exact)
  else:
    raise TypeError(type(a))


def assert_equal(a, b, check_dim_order=True):
  _compare(a, b, 0, 0, True)


def assert_identical(a, b):
  _compare(a, b, 0, 0, True, check_names=True)
  assert a.attrs == b.attrs, f'attrs {a.attrs} vs {b.attrs}'


def assert_allclose(a, b, rtol=1e-05, atol=1e-08, decode_bytes=True,
                    check_dim_order=True):
  _compare(a, b, rtol, atol, False)
