"""The mini-xarray of oracle/refshim (test infrastructure): it must keep
passing the REFERENCE's own unit tests, which is what makes the vectors of
tests/golden/reference_vectors_v1.npz outputs of the reference's code.

Where /root/reference exists (the build container) the reference's
metrics_test.py, regions_test.py and the ZonalEnergySpectrum tests run on it
and the reference vectors are regenerated and compared with the committed file;
elsewhere (the GPU box) those two tests skip and only the stand-alone semantics
checks run.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, 'oracle', 'refshim')
REFERENCE = os.environ.get('WB2_REFERENCE', '/root/reference')
needs_reference = pytest.mark.skipif(
    not os.path.isdir(os.path.join(REFERENCE, 'weatherbench2')),
    reason='the reference checkout is only present in the build container')


def _run(script, *args):
  return subprocess.run([sys.executable, os.path.join(ROOT, script), *args],
                        capture_output=True, text=True, cwd=ROOT, timeout=900,
                        env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'))


@needs_reference
def test_reference_unit_tests_pass_on_the_mini_xarray():
  r = _run('oracle/refshim/run_reference_tests.py')
  tail = r.stdout.strip().splitlines()[-1]  # (unittest reports on stderr)
  assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
  assert 'failures 0, errors 0' in tail and 'ran 82' in tail, tail
  # the committed report lists the same tests
  committed = open(os.path.join(ROOT, 'tests', 'golden',
                                'reference_selftest.txt')).read()
  assert 'ran 82, failures 0, errors 0' in committed


@needs_reference
def test_committed_reference_vectors_are_what_the_reference_produces(tmp_path):
  """make_reference_vectors.py, re-run now, reproduces the committed .npz bit
  for bit (so the fixture cannot drift from the generator or the reference)."""
  env = dict(os.environ, WB2_VECTORS_OUT=str(tmp_path / 'v.npz'),
             PYTHONDONTWRITEBYTECODE='1')
  r = subprocess.run([sys.executable, os.path.join(
      ROOT, 'tests', 'golden', 'make_reference_vectors.py')],
                     capture_output=True, text=True, cwd=ROOT, env=env,
                     timeout=900)
  assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
  fresh = np.load(tmp_path / 'v.npz')
  have = np.load(os.path.join(ROOT, 'tests', 'golden',
                              'reference_vectors_v1.npz'))
  assert sorted(fresh.files) == sorted(have.files)
  for k in have.files:
    np.testing.assert_array_equal(fresh[k], have[k], err_msg=k)


# ---------------------------------------------------------------------------
# stand-alone semantics of the stand-in (no reference needed)
# ---------------------------------------------------------------------------
@pytest.fixture(scope='module')
def xr():
  sys.path.insert(0, SHIM)
  try:
    import importlib
    mod = importlib.import_module('xarray')
    assert 'wb2shim' in mod.__version__
    yield mod
  finally:
    sys.path.remove(SHIM)
    for name in [m for m in sys.modules if m == 'xarray' or
                 m.startswith('xarray.')]:
      del sys.modules[name]


def test_arithmetic_broadcasts_by_name_and_joins_indexes(xr):
  a = xr.DataArray(np.arange(6.0).reshape(2, 3), dims=('x', 'y'),
                   coords={'x': [10, 20], 'y': [1, 2, 3]})
  b = xr.DataArray(np.array([1.0, 2.0, 4.0]), dims=('y',),
                   coords={'y': [3, 2, 9]})
  c = a * b  # inner join on y: labels 2 and 3 survive, in a's order
  assert c.dims == ('x', 'y') and list(c.y.values) == [2, 3]
  np.testing.assert_array_equal(c.values, [[1 * 2.0, 2 * 1.0],
                                           [4 * 2.0, 5 * 1.0]])
  d = b + a  # left operand's dims first
  assert d.dims == ('y', 'x')
  ds = xr.Dataset({'p': a, 'q': a + 1}) - xr.Dataset({'q': a, 'r': a})
  assert list(ds.data_vars) == ['q']  # only common variables survive


def test_weighted_mean_follows_xarray_weighted(xr):
  x = np.array([[1.0, np.nan, 3.0], [4.0, 5.0, 6.0]], dtype=np.float32)
  da = xr.DataArray(x, dims=('t', 'lat'))
  w = xr.DataArray(np.array([1.0, 2.0, 0.5]), dims=('lat',))
  skip = da.weighted(w).mean(['lat'], skipna=True)
  np.testing.assert_allclose(skip.values, [(1 + 1.5) / 1.5, (4 + 10 + 3) / 3.5])
  assert skip.dtype == np.float64  # float32 data x float64 weights in einsum
  keep = da.weighted(w).mean(['lat'], skipna=False)
  assert np.isnan(keep.values[0]) and keep.values[1] == skip.values[1]
  zero = da.weighted(w * 0).mean(['lat'], skipna=True)
  assert np.isnan(zero.values).all()  # sum of weights 0 -> NaN
  with pytest.raises(ValueError):
    da.weighted(xr.DataArray(np.array([1.0, np.nan, 1.0]), dims=('lat',)))


def test_reductions_choose_plain_or_nan_functions(xr):
  x = np.array([1.0, np.nan, 3.0], dtype=np.float32)
  da = xr.DataArray(x, dims=('m',))
  assert np.isnan(da.mean('m', skipna=False).values)
  assert da.mean('m').values == 2.0            # skipna=None skips for floats
  assert da.var('m', ddof=1, skipna=True).values == 2.0
  assert xr.DataArray(np.array([1, 2]), dims='m').mean('m').dtype == np.float64


def test_label_selection(xr):
  lat = np.linspace(-90, 90, 19)
  da = xr.DataArray(np.arange(19.0), dims=('latitude',),
                    coords={'latitude': lat})
  assert list(da.sel(latitude=slice(-20, 20)).latitude.values) == [
      -20, -10, 0, 10, 20]                       # both ends inclusive
  assert list(da.sel(latitude=slice(-25, 25)).latitude.values) == [
      -20, -10, 0, 10, 20]
  assert da.sel(latitude=[30.0, -30.0]).values.tolist() == [12.0, 6.0]
  assert da.sel(latitude=10.0).dims == ()
  with pytest.raises(KeyError):
    da.sel(latitude=11.0)
  # pointwise (vectorised) selection with DataArray indexers sharing a dim
  clim = xr.DataArray(np.arange(24.0).reshape(2, 3, 4),
                      dims=('hour', 'dayofyear', 'x'),
                      coords={'hour': [0, 12], 'dayofyear': [58, 59, 60]})
  doy = xr.DataArray(np.array([60, 58, 59, 60]), dims=('time',))
  hour = xr.DataArray(np.array([0, 12, 0, 12]), dims=('time',))
  got = clim.sel(dayofyear=doy, hour=hour)
  assert got.dims == ('time', 'x')
  np.testing.assert_array_equal(got.values[:, 0], [8.0, 12.0, 4.0, 20.0])
  # a dimension without an index is addressed by position
  bare = xr.DataArray(np.arange(5.0), dims=('dayofyear',))
  assert bare.sel(dayofyear=xr.DataArray([1, 3], dims='t')).values.tolist() \
      == [1.0, 3.0]


def test_concat_and_merge_orders(xr):
  a = xr.DataArray(np.zeros((2, 3)), dims=('x', 'y'))
  b = xr.DataArray(np.ones((3, 2, 4)), dims=('y', 'x', 'm'))
  c = xr.concat([a, b], dim='new')   # new dim first, then first appearance
  assert c.dims == ('new', 'x', 'y', 'm') and c.shape == (2, 2, 3, 4)
  one = lambda name: xr.Dataset({'v': xr.DataArray(
      np.ones((1, 2)), dims=('metric', 't'), coords={'metric': [name]})})
  merged = xr.merge([one('mse'), one('acc'), one('bias')])
  assert list(merged.metric.values) == ['acc', 'bias', 'mse']  # sorted union
  same = xr.merge([one('mse'), one('mse').rename({'v': 'w'})])
  assert list(same.metric.values) == ['mse'] and set(same.data_vars) == {'v',
                                                                        'w'}


@needs_reference
def test_the_reference_checkout_is_left_untouched():
  """Importing the reference from these tests must not leave bytecode caches
  (or anything else) behind in the read-only checkout."""
  leftovers = []
  for base, dirs, files in os.walk(REFERENCE):
    leftovers += [os.path.join(base, d) for d in dirs
                  if d in ('__pycache__', '.pytest_cache')]
    leftovers += [os.path.join(base, f) for f in files if f.endswith('.pyc')]
  assert not leftovers, leftovers[:5]
