"""Host-side helpers of the loop against the REFERENCE's own functions (CPU,
build container only; reference code on the mini-xarray of oracle/refshim, its
beam imports resolved to import-only stand-ins):

  evaluation.make_latitude_increasing            evaluation.py:41-47
  truth.sel(time=forecast.valid_time)            evaluation.py:474-475
  metrics.get_lat_weights                        metrics.py:55-60

xarray objects go in, xarray objects must come out, equal to the reference's
(values, dims, coordinates)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get('WB2_REFERENCE', '/root/reference')
pytestmark = pytest.mark.skipif(
    not os.path.isdir(os.path.join(REFERENCE, 'weatherbench2')),
    reason='the reference checkout is only present in the build container')

SCRIPT = textwrap.dedent('''
    import numpy as np
    import xarray as xr
    assert 'wb2shim' in xr.__version__
    from weatherbench2 import evaluation as rev, metrics as rm
    from weatherbench2_amd import evaluation as gev, metrics as gm
    from weatherbench2_amd import xarray_lite as xl
    assert xl._xr is xr

    rs = np.random.RandomState(3)
    lat = np.linspace(90, -90, 19).astype(np.float32)      # ERA5: decreasing
    lon = np.linspace(0, 360, 36, endpoint=False).astype(np.float32)
    time = np.datetime64('2020-01-01T00', 'ns') + np.arange(7) * np.timedelta64(
        6, 'h')
    ds = xr.Dataset(
        {'z': (('time', 'level', 'latitude', 'longitude'),
               rs.standard_normal((7, 2, 19, 36)).astype(np.float32)),
         'lsm': (('latitude', 'longitude'), rs.rand(19, 36)),
         'scalar_per_time': (('time',), np.arange(7.0))},
        coords={'time': time, 'level': [500, 850], 'latitude': lat,
                'longitude': lon}, attrs={'source': 'test'})

    # --- make_latitude_increasing -----------------------------------------
    want = rev.make_latitude_increasing(ds)
    got = gev.make_latitude_increasing(ds)
    assert isinstance(got, xr.Dataset)
    xr.testing.assert_equal(got, want)
    assert got.latitude.dtype == np.float32
    inc = rev.make_latitude_increasing(want)               # already increasing
    xr.testing.assert_equal(gev.make_latitude_increasing(want), inc)

    # --- latitude weights ---------------------------------------------------
    for dtype in (np.float32, np.float64):
      d = want.assign_coords(latitude=want.latitude.astype(dtype))
      w_ref = rm.get_lat_weights(d)
      w_got = gm.get_lat_weights(d)
      assert isinstance(w_got, xr.DataArray) and w_got.dtype == w_ref.dtype
      np.testing.assert_array_equal(w_got.values, w_ref.values)   # bit-identical
      assert w_got.dims == w_ref.dims == ('latitude',)

    # --- truth.sel(time=forecast.valid_time) --------------------------------
    init = time[:3]
    lead = (np.arange(3) * np.timedelta64(12, 'h')).astype('timedelta64[ns]')
    forecast = xr.Dataset(
        {'z': (('init_time', 'lead_time', 'level', 'latitude', 'longitude'),
               rs.standard_normal((3, 3, 2, 19, 36)).astype(np.float32))},
        coords={'init_time': init, 'lead_time': lead, 'level': [500, 850],
                'latitude': want.latitude.values, 'longitude': lon,
                'valid_time': (('init_time', 'lead_time'),
                               init[:, None] + lead[None, :])})
    truth = want[['z']]
    sel_ref = truth.sel(time=forecast.valid_time)          # evaluation.py:474
    sel_got = gev.select_truth_at_valid_time(truth, forecast,
                                             init_dim='init_time',
                                             lead_dim='lead_time')
    assert isinstance(sel_got, xr.Dataset)
    assert sel_got['z'].dims == sel_ref['z'].dims
    np.testing.assert_array_equal(sel_got['z'].values, sel_ref['z'].values)
    for c in ('valid_time', 'time', 'init_time', 'lead_time', 'level',
              'latitude', 'longitude'):
      assert c in sel_ref.coords and c in sel_got.coords, c
      assert sel_got.coords[c].dims == sel_ref.coords[c].dims, c
      np.testing.assert_array_equal(sel_got.coords[c].values,
                                    sel_ref.coords[c].values)
    late = forecast.assign_coords(init_time=init + np.timedelta64(30, 'D'))
    late = late.assign_coords(valid_time=late.init_time + late.lead_time)
    for fn in (lambda: truth.sel(time=late.valid_time),
               lambda: gev.select_truth_at_valid_time(
                   truth, late, init_dim='init_time', lead_dim='lead_time')):
      try:
        fn()
      except KeyError:
        pass
      else:
        raise AssertionError('missing valid times must raise KeyError')
    print('HELPERS-OK')
''')


def test_host_helpers_equal_the_reference_functions():
  env = dict(os.environ)
  env['PYTHONPATH'] = os.pathsep.join(
      [os.path.join(ROOT, 'oracle', 'refshim'), REFERENCE, ROOT,
       env.get('PYTHONPATH', '')])
  res = subprocess.run([sys.executable, '-c', SCRIPT], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
  assert res.returncode == 0 and 'HELPERS-OK' in res.stdout, (
      res.stdout[-1500:] + res.stderr[-5000:])
