"""The xarray boundary without xarray: a stub package (tests/stubs/xarray) is
put on PYTHONPATH in a subprocess, so `import xarray` inside xarray_lite picks
it up.  Checks the conversion code paths -- xarray in => xarray out at every
public entry point, lite in => lite out -- that the build image cannot
exercise against the real library."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import numpy as np
    import xarray as xr                      # the stub
    from weatherbench2_amd import xarray_lite as xl
    from weatherbench2_amd import metrics as gm
    assert xl._xr is xr

    lat = np.linspace(-90, 90, 7); lon = np.linspace(0, 360, 12, endpoint=False)
    t = np.array(['2020-01-01', '2020-01-02'], dtype='datetime64[ns]')
    ds = xr.Dataset({'z': (('time', 'latitude', 'longitude'),
                           np.arange(2 * 7 * 12, dtype=np.float32).reshape(2, 7, 12))},
                    coords={'time': t, 'latitude': lat, 'longitude': lon,
                            'valid_time': (('time',), t)}, attrs={'a': 1})
    assert xl.is_xarray(ds) and not xl.is_xarray(xl.Dataset())
    lite = xl.as_dataset(ds)
    assert isinstance(lite, xl.Dataset) and lite['z'].dims == ('time', 'latitude', 'longitude')
    assert lite['z'].data is ds['z'].data          # zero copy, identity kept
    np.testing.assert_array_equal(lite.coords['latitude'], lat)
    assert lite.attrs == {'a': 1}
    back = xl.to_xarray(lite)
    assert isinstance(back, xr.Dataset) and back['z'].dims == lite['z'].dims
    np.testing.assert_array_equal(back['z'].values, ds['z'].values)

    # xarray in => xarray out at the metric entry points; lite in => lite out
    class Fake(gm.Metric):                      # no GPU needed
      def compute_chunk(self, forecast, truth, region=None, skipna=False):
        forecast, truth = xl.as_dataset(forecast), xl.as_dataset(truth)
        d = forecast['z'].values - truth['z'].values
        out = xl.Dataset(coords={'time': forecast.coords['time']})
        out.data_vars['z'] = xl.DataArray(d.mean((1, 2)), ('time',), out.coords, 'z')
        return out
    m = Fake()
    r = m.compute_chunk(ds, ds)
    assert isinstance(r, xr.Dataset), type(r)
    np.testing.assert_array_equal(r['z'].values, [0.0, 0.0])
    assert isinstance(m.compute_chunk(lite, lite), xl.Dataset)
    assert isinstance(m.compute(ds, ds), xr.Dataset)
    assert m.compute(ds, ds)['z'].dims == ()
    assert isinstance(m.compute(lite, lite), xl.Dataset)
    rr = m.compute_chunk_regions(ds, ds, {'a': None, 'b': None})
    assert isinstance(rr, xr.Dataset) and rr['z'].dims == ('region', 'time')
    assert isinstance(m.compute_regions(lite, lite, {'a': None}), xl.Dataset)

    # central_reliability on xarray objects (host-only function)
    hist = xr.Dataset({'temperature': (('bins',), np.array([0.2, 0.1, 0.7]))},
                      coords={'bins': np.arange(3)})
    rel = gm.central_reliability(hist)
    assert isinstance(rel, xr.Dataset)
    np.testing.assert_allclose(rel['temperature'].values, [0.1, 1.0])
    assert rel['temperature'].dims == ('desired_prob',)
    rel_da = gm.central_reliability(hist['temperature'])
    assert isinstance(rel_da, xr.DataArray)
    np.testing.assert_allclose(rel_da.values, [0.1, 1.0])
    print('BOUNDARY-OK')
''')


def test_xarray_in_gives_xarray_out_with_a_stub_xarray():
  env = dict(os.environ)
  env['PYTHONPATH'] = os.pathsep.join(
      [os.path.join(ROOT, 'tests', 'stubs'), ROOT, env.get('PYTHONPATH', '')])
  res = subprocess.run([sys.executable, '-c', SCRIPT], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
  assert res.returncode == 0 and 'BOUNDARY-OK' in res.stdout, (
      res.stdout[-2000:] + res.stderr[-4000:])
