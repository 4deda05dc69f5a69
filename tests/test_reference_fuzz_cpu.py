"""Seeded differential fuzzing of the ORACLE against the REFERENCE's own code
(CPU, build container only): random grids (irregular latitudes too), dim
orders, dtypes, NaN / inf patterns, skipna, random region sets -- the same
generators the GPU fuzz (tests/test_fuzz_gpu.py) uses against the HIP path, so
the chain  reference code == oracle == HIP path  is closed case by case.

The reference (/root/reference, unmodified) runs on the mini-xarray of
oracle/refshim in a subprocess (so that `xarray` never becomes importable in
the pytest process itself).  Skipped where the reference checkout is absent.
"""
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
    import sys
    import numpy as np
    import xarray as xr
    assert 'wb2shim' in xr.__version__
    from weatherbench2 import metrics as rm, regions as rr
    from weatherbench2 import derived_variables as rdv
    from oracle import metrics_np as om, regions_np as oreg, spectrum_np
    from oracle.named import DS, NA
    from tests import test_fuzz_gpu as fz      # the GPU fuzz's generators

    family, n_cases = sys.argv[1], int(sys.argv[2])

    def to_xr(ds):
      coords = {k: ((c.dims, c.data) if isinstance(c, NA) else c)
                for k, c in ds.coords.items()}
      return xr.Dataset({k: (v.dims, v.data) for k, v in ds.items()}, coords)

    def to_ref_region(region):
      if isinstance(region, oreg.SliceRegion):
        return rr.SliceRegion(lat_slice=region.lat_slice,
                              lon_slice=region.lon_slice)
      if isinstance(region, oreg.ExtraTropicalRegion):
        return rr.ExtraTropicalRegion()
      if isinstance(region, oreg.LandRegion):
        lsm = xr.DataArray(region.land_sea_mask.data,
                           dims=region.land_sea_mask.dims,
                           coords={'latitude': region.latitude,
                                   'longitude': region.longitude})
        return rr.LandRegion(land_sea_mask=lsm, threshold=region.threshold)
      if isinstance(region, oreg.CombinedRegion):
        return rr.CombinedRegion(regions=[to_ref_region(r)
                                          for r in region.regions])
      raise TypeError(region)

    def compare(want, got, what):
      assert tuple(got.dims) == tuple(want.dims), (what, got.dims, want.dims)
      w = np.asarray(want.data, dtype=np.float64)
      g = np.asarray(got.data, dtype=np.float64)
      np.testing.assert_allclose(g, w, rtol=1e-12, atol=1e-12, equal_nan=True,
                                 err_msg=what)

    n_checked = 0
    for seed in range(n_cases):
      if family == 'det':
        rs = np.random.RandomState(1000 + seed)
        lat, lon = fz._grid(rs)
        sizes = {'time': int(rs.randint(1, 4)), 'level': int(rs.randint(1, 4)),
                 'prediction_timedelta': int(rs.randint(1, 3)),
                 'latitude': len(lat), 'longitude': len(lon)}
        t0 = np.datetime64('2020-03-01T00', 'ns')
        coords = {'time': t0 + np.arange(sizes['time']) * np.timedelta64(6, 'h'),
                  'level': np.array([500, 700, 850])[:sizes['level']],
                  'prediction_timedelta': (
                      np.arange(sizes['prediction_timedelta'])
                      * np.timedelta64(6, 'h')).astype('timedelta64[ns]'),
                  'latitude': lat, 'longitude': lon}
        spatial = (['latitude', 'longitude'] if rs.rand() < 0.6
                   else ['longitude', 'latitude'])
        outer = ['prediction_timedelta', 'time', 'level']
        rs.shuffle(outer)
        fdims = tuple(outer) + tuple(spatial)
        touter = [d for d in outer if d != 'prediction_timedelta']
        rs.shuffle(touter)
        tdims = tuple(touter) + tuple(spatial)
        dtype = np.float32 if rs.rand() < 0.6 else np.float64
        skipna = bool(rs.rand() < 0.5)
        nan_frac = 0.08 if rs.rand() < 0.5 else 0.0
        inf_frac = 0.02 if rs.rand() < 0.2 else 0.0
        forecast = fz._dataset(rs, fdims, sizes, coords, dtype, nan_frac,
                               inf_frac)
        truth = fz._dataset(rs, tdims, sizes, coords, dtype, nan_frac / 2,
                            inf_frac / 2)
        regions = {f'r{i}': fz._random_region(rs, lat, lon) for i in range(3)}
        pairs = [(om.MSE(), rm.MSE()), (om.MAE(), rm.MAE()),
                 (om.Bias(), rm.Bias()),
                 (om.RMSESqrtBeforeTimeAvg(), rm.RMSESqrtBeforeTimeAvg())]
        if rs.rand() < 0.5:
          cdims = ('hour', 'dayofyear', 'level') + tuple(spatial)
          csizes = {'hour': 4, 'dayofyear': 3, **sizes}
          ccoords = {'hour': np.array([0, 6, 12, 18]),
                     'dayofyear': np.array([60, 61, 62]), **coords}
          clim = fz._dataset(rs, cdims, csizes, ccoords, dtype, 0.0)
          pairs.append((om.ACC(clim), rm.ACC(climatology=to_xr(clim))))
      else:
        rs = np.random.RandomState(2000 + seed)
        lat, lon = fz._grid(rs)
        m = int(rs.choice([1, 2, 3, 6, 17, 33]))
        sizes = {'realization': m, 'time': int(rs.randint(1, 3)),
                 'level': int(rs.randint(1, 3)), 'latitude': len(lat),
                 'longitude': len(lon)}
        coords = {'realization': np.arange(m), 'time': np.arange(sizes['time']),
                  'level': np.arange(sizes['level']), 'latitude': lat,
                  'longitude': lon}
        spatial = (['latitude', 'longitude'] if rs.rand() < 0.6
                   else ['longitude', 'latitude'])
        outer = ['realization', 'time', 'level']
        rs.shuffle(outer)
        fdims = tuple(outer) + tuple(spatial)
        tdims = tuple(d for d in outer if d != 'realization') + tuple(spatial)
        dtype = np.float32 if rs.rand() < 0.6 else np.float64
        skipna = bool(rs.rand() < 0.5)
        nan_frac = 0.05 if skipna else 0.0
        forecast = fz._dataset(rs, fdims, sizes, coords, dtype, nan_frac)
        truth = fz._dataset(rs, tdims, sizes, coords, dtype, nan_frac / 2)
        regions = {f'r{i}': fz._random_region(rs, lat, lon) for i in range(2)}
        names = ('CRPS', 'CRPSSpread', 'CRPSSkill', 'EnsembleMeanMSE',
                 'EnsembleVariance', 'DebiasedEnsembleMeanMSE',
                 'EnsembleMeanRMSESqrtBeforeTimeAvg',
                 'EnsembleStddevSqrtBeforeTimeAvg')
        pairs = [(getattr(om, n)(), getattr(rm, n)()) for n in names]
      xf, xt = to_xr(forecast), to_xr(truth)
      for oc, rc_ in pairs:
        for rname, region in regions.items():
          want = rc_.compute_chunk(xf, xt, region=to_ref_region(region),
                                   skipna=skipna)['z']
          got = oc.compute_chunk(forecast, truth, region=region,
                                 skipna=skipna)['z']
          compare(want, got, f'{family} seed={seed} {type(oc).__name__} '
                  f'{rname} {fdims} {dtype.__name__} skipna={skipna}')
          n_checked += 1
    print(f'FUZZ-OK {family} cases={n_cases} comparisons={n_checked}')
''')


def _run(family, n):
  env = dict(os.environ)
  env['PYTHONPATH'] = os.pathsep.join(
      [os.path.join(ROOT, 'oracle', 'refshim'), REFERENCE, ROOT,
       env.get('PYTHONPATH', '')])
  return subprocess.run([sys.executable, '-c', SCRIPT, family, str(n)],
                        env=env, cwd=ROOT, capture_output=True, text=True,
                        timeout=1200)


@pytest.mark.parametrize('family,n', [('det', 160), ('ens', 60)])
def test_oracle_equals_the_reference_on_random_cases(family, n):
  res = _run(family, n)
  assert res.returncode == 0 and f'FUZZ-OK {family}' in res.stdout, (
      res.stdout[-1500:] + res.stderr[-5000:])
