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

    def maps_case(seed):
      """tests/test_fuzz_gpu.py::test_maps_gaussian_and_rank_histogram_fuzz"""
      global n_checked
      rs = np.random.RandomState(3000 + seed)
      lat, lon = fz._grid(rs)
      m = int(rs.choice([2, 3, 5, 9, 20]))
      sizes = {'realization': m, 'time': int(rs.randint(1, 4)),
               'level': int(rs.randint(1, 3)), 'latitude': len(lat),
               'longitude': len(lon)}
      coords = {'realization': np.arange(m), 'time': np.arange(sizes['time']),
                'level': np.arange(sizes['level']), 'latitude': lat,
                'longitude': lon}
      spatial = (['latitude', 'longitude'] if rs.rand() < 0.6
                 else ['longitude', 'latitude'])
      outer = ['time', 'level']
      rs.shuffle(outer)
      ddims = tuple(outer) + tuple(spatial)
      eouter = ['realization', 'time', 'level']
      rs.shuffle(eouter)
      edims = tuple(eouter) + tuple(spatial)
      dtype = np.float32 if rs.rand() < 0.6 else np.float64
      skipna = bool(rs.rand() < 0.5)
      nan_frac = 0.05 if skipna else 0.0
      tag = f'maps seed={seed} {dtype.__name__} skipna={skipna} M={m}'
      forecast = fz._dataset(rs, ddims, sizes, coords, dtype, nan_frac)
      truth = fz._dataset(rs, ddims, sizes, coords, dtype, 0.0)
      for name in ('SpatialMSE', 'SpatialMAE', 'SpatialBias'):
        want = getattr(rm, name)().compute_chunk(to_xr(forecast),
                                                 to_xr(truth))['z']
        got = getattr(om, name)().compute_chunk(forecast, truth)['z']
        compare(want, got, f'{name} {tag}')
        n_checked += 1
      ens = fz._dataset(rs, edims, sizes, coords, dtype, nan_frac)
      for name in ('SpatialCRPS', 'SpatialEnsembleVariance'):
        want = getattr(rm, name)().compute_chunk(to_xr(ens), to_xr(truth),
                                                 skipna=skipna)['z']
        got = getattr(om, name)().compute_chunk(ens, truth, skipna=skipna)['z']
        compare(want, got, f'{name} {tag}')
        n_checked += 1
      clean = fz._dataset(rs, edims, sizes, coords, dtype, 0.0)
      want = rm.RankHistogram(break_ties_randomly=False).compute_chunk(
          to_xr(clean), to_xr(truth))['z']
      got = om.RankHistogram(break_ties_randomly=False).compute_chunk(
          clean, truth)['z']
      compare(want, got, f'RankHistogram {tag}')
      n_checked += 1
      mean = fz._dataset(rs, ddims, sizes, coords, dtype, nan_frac)
      std = fz._dataset(rs, ddims, sizes, coords, dtype, 0.0)
      gf = DS({'z': mean['z'],
               'z_std': NA(np.abs(std['z'].data) + dtype(0.1), std['z'].dims)},
              mean.coords)
      for name in ('GaussianCRPS', 'GaussianVariance'):
        want = getattr(rm, name)().compute_chunk(to_xr(gf), to_xr(truth),
                                                 skipna=skipna)['z']
        got = getattr(om, name)().compute_chunk(gf, truth, skipna=skipna)['z']
        compare(want, got, f'{name} {tag}')
        n_checked += 1

    def thr_case(seed):
      """tests/test_fuzz_gpu.py::test_threshold_family_fuzz"""
      global n_checked
      from oracle import thresholds_np as oth
      from weatherbench2 import thresholds as rth
      rs = np.random.RandomState(4000 + seed)
      lat, lon = fz._grid(rs)
      m = int(rs.choice([1, 2, 4, 7, 31]))
      n_time = int(rs.randint(1, 4))
      t0 = np.datetime64('2021-02-27T00', 'ns')
      sizes = {'realization': m, 'time': n_time,
               'level': int(rs.randint(1, 3)), 'latitude': len(lat),
               'longitude': len(lon), 'dayofyear': 6}
      coords = {'realization': np.arange(m),
                'time': t0 + np.arange(n_time) * np.timedelta64(24, 'h'),
                'level': np.array([500, 850])[:sizes['level']],
                'latitude': lat, 'longitude': lon,
                'dayofyear': 57 + np.arange(6)}
      spatial = (['latitude', 'longitude'] if rs.rand() < 0.6
                 else ['longitude', 'latitude'])
      outer = ['time', 'level']
      rs.shuffle(outer)
      ddims = tuple(outer) + tuple(spatial)
      eouter = ['realization', 'time', 'level']
      rs.shuffle(eouter)
      edims = tuple(eouter) + tuple(spatial)
      cdims = ('dayofyear', 'level') + tuple(spatial)
      dtype = np.float32 if rs.rand() < 0.5 else np.float64
      skipna = bool(rs.rand() < 0.5)
      nan_frac = 0.05 if skipna else 0.0
      truth = fz._dataset(rs, ddims, sizes, coords, dtype, nan_frac / 2)
      cm = fz._dataset(rs, cdims, sizes, coords, dtype, 0.0)['z']
      cs = fz._dataset(rs, cdims, sizes, coords, dtype, 0.0)['z']
      clim = DS({'z': NA(cm.data * dtype(0.3), cdims),
                 'z_std': NA(np.abs(cs.data) * dtype(0.5) + dtype(0.5), cdims)},
                {d: coords[d] for d in cdims})
      qs = (0.25, 0.7)
      oths = [oth.GaussianQuantileThreshold(clim, q) for q in qs]
      rths = [rth.GaussianQuantileThreshold(climatology=to_xr(clim), quantile=q)
              for q in qs]
      region = fz._random_region(rs, lat, lon)
      tag = f'thr seed={seed} {dtype.__name__} skipna={skipna} M={m}'
      ens = fz._dataset(rs, edims, sizes, coords, dtype, nan_frac)
      for name in ('EnsembleBrierScore', 'DebiasedEnsembleBrierScore',
                   'EnsembleIgnoranceScore', 'EnsembleRPS'):
        want = getattr(rm, name)(thresholds=rths).compute_chunk(
            to_xr(ens), to_xr(truth), region=to_ref_region(region),
            skipna=skipna)['z']
        got = getattr(om, name)(thresholds=oths).compute_chunk(
            ens, truth, region=region, skipna=skipna)['z']
        compare(want, got, f'{name} {tag}')
        n_checked += 1
      mean = fz._dataset(rs, ddims, sizes, coords, dtype, nan_frac)
      std = fz._dataset(rs, ddims, sizes, coords, dtype, 0.0)
      gf = DS({'z': mean['z'],
               'z_std': NA(np.abs(std['z'].data) + dtype(0.2), std['z'].dims)},
              mean.coords)
      for name in ('GaussianBrierScore', 'GaussianIgnoranceScore',
                   'GaussianRPS'):
        want = getattr(rm, name)(thresholds=rths).compute_chunk(
            to_xr(gf), to_xr(truth), region=to_ref_region(region),
            skipna=skipna)['z']
        got = getattr(om, name)(thresholds=oths).compute_chunk(
            gf, truth, region=region, skipna=skipna)['z']
        compare(want, got, f'{name} {tag}')
        n_checked += 1

    def spectrum_case(seed):
      """tests/test_fuzz_gpu.py::test_spectrum_fuzz (the materialised part)"""
      global n_checked
      rs = np.random.RandomState(5000 + seed)
      n_lon = int(rs.choice([6, 9, 15, 16, 30, 45, 64, 72, 100, 128, 240, 360]))
      n_lat = int(rs.randint(1, 9))
      lat = np.sort(rs.uniform(-85, 85, n_lat))
      lon = np.linspace(0, 360, n_lon, endpoint=False)
      n_time, n_lev = int(rs.randint(1, 4)), int(rs.randint(1, 3))
      dtype = np.float32 if rs.rand() < 0.6 else np.float64
      x = rs.standard_normal((n_time, n_lev, n_lat, n_lon)).astype(dtype)
      canon = ('time', 'level', 'latitude', 'longitude')
      perm = list(range(4))
      rs.shuffle(perm)
      dims = tuple(canon[i] for i in perm)
      ds = xr.Dataset({'z': (dims, np.ascontiguousarray(np.transpose(x, perm)))},
                      {'time': np.arange(n_time), 'level': np.arange(n_lev),
                       'latitude': lat, 'longitude': lon})
      want = rdv.ZonalEnergySpectrum('z').compute(ds)
      got, freq, _ = spectrum_np.zonal_energy_spectrum(x, lat, lon, lat_axis=2,
                                                       lon_axis=3)
      assert want.dims == tuple(d for d in dims if d != 'longitude') + (
          'zonal_wavenumber',), (want.dims, dims)
      w = want.transpose('time', 'level', 'latitude', 'zonal_wavenumber').data
      np.testing.assert_allclose(got, w, rtol=1e-12, atol=0,
                                 err_msg=f'spectrum seed={seed} {dims}')
      np.testing.assert_allclose(
          freq, want.coords['frequency'].transpose('zonal_wavenumber',
                                                   'latitude').data, rtol=1e-14)
      n_checked += 1

    def seeps_case(seed):
      """tests/test_fuzz_gpu.py::test_seeps_fuzz"""
      global n_checked
      rs = np.random.RandomState(7000 + seed)
      lat, lon = fz._grid(rs)
      name = 'total_precipitation_24hr'
      n_time = int(rs.randint(1, 4))
      t0 = np.datetime64('2022-03-01T00', 'ns')
      time = t0 + np.arange(n_time) * np.timedelta64(24, 'h')
      spatial = (('latitude', 'longitude') if rs.rand() < 0.6
                 else ('longitude', 'latitude'))
      sshape = tuple(len(lat) if d == 'latitude' else len(lon) for d in spatial)
      dtype = np.float32
      dry = np.float32(0.25 / 1000.0)
      wet = (rs.uniform(0.002, 0.02, size=(4, 3) + sshape)).astype(dtype)
      frac = rs.uniform(0.0, 1.0, size=(4, 3) + sshape).astype(dtype)

      def precip():
        x = (rs.gamma(0.3, 2.0, size=(n_time,) + sshape) * 1e-2).astype(dtype)
        pick = rs.rand(*x.shape)
        x = np.where(pick < 0.1, dry, x)
        x = np.where((pick >= 0.1) & (pick < 0.2), wet[0, 0][None], x)
        x = np.where(pick > 0.97, np.nan, x)
        return x.astype(dtype)
      dims = ('time',) + spatial
      coords = {'time': time, 'latitude': lat, 'longitude': lon,
                'valid_time': NA(time, ('time',))}
      forecast = DS({name: NA(precip(), dims)}, coords)
      truth = DS({name: NA(precip(), dims)}, coords)
      cdims = ('hour', 'dayofyear') + spatial
      clim = DS({name + '_seeps_threshold': NA(wet, cdims),
                 name + '_seeps_dry_fraction': NA(frac, cdims)},
                {'hour': np.array([0, 6, 12, 18]),
                 'dayofyear': np.array([60, 61, 62]), 'latitude': lat,
                 'longitude': lon})
      region = fz._random_region(rs, lat, lon)
      want = rm.SEEPS(climatology=to_xr(clim)).compute_chunk(
          to_xr(forecast), to_xr(truth), region=to_ref_region(region))[name]
      got = om.SEEPS(climatology=clim).compute_chunk(forecast, truth,
                                                     region=region)[name]
      compare(want, got, f'SEEPS seed={seed}')
      wmap = rm.SpatialSEEPS(climatology=to_xr(clim)).compute_chunk(
          to_xr(forecast), to_xr(truth))[name]
      gmap = om.SpatialSEEPS(climatology=clim).compute_chunk(forecast,
                                                             truth)[name]
      compare(wmap, gmap, f'SpatialSEEPS seed={seed}')
      n_checked += 2

    def rank_case(seed):
      """Seeded rank histograms on heavily tied data in random layouts: the
      reference's perturbation stream (np.random.default_rng(seed).uniform over
      concat([truth, forecast]), metrics.py:1955-2010) decides the bins, so the
      oracle must consume NumPy's stream in exactly the reference's element
      order; and Metric.compute (time mean) of the deterministic metrics."""
      global n_checked
      rs = np.random.RandomState(8000 + seed)
      lat, lon = fz._grid(rs)
      m = int(rs.choice([2, 3, 5, 8]))
      sizes = {'realization': m, 'time': int(rs.randint(1, 4)),
               'level': int(rs.randint(1, 3)), 'latitude': len(lat),
               'longitude': len(lon)}
      coords = {'realization': np.arange(m), 'time': np.arange(sizes['time']),
                'level': np.arange(sizes['level']), 'latitude': lat,
                'longitude': lon}
      spatial = (['latitude', 'longitude'] if rs.rand() < 0.6
                 else ['longitude', 'latitude'])
      eouter = ['realization', 'time', 'level']
      rs.shuffle(eouter)
      edims = tuple(eouter) + tuple(spatial)
      touter = ['time', 'level']
      rs.shuffle(touter)
      tdims = tuple(touter) + tuple(spatial)
      dtype = np.float32 if rs.rand() < 0.5 else np.float64
      ens = fz._dataset(rs, edims, sizes, coords, dtype, 0.0)
      truth = fz._dataset(rs, tdims, sizes, coords, dtype, 0.0)
      round1 = lambda ds: DS({'z': NA(np.round(ds['z'].data, 1).astype(dtype),
                                      ds['z'].dims)}, ds.coords)
      ens, truth = round1(ens), round1(truth)             # many exact ties
      rng_seed = int(rs.randint(0, 2**31 - 1))
      nb = None if (m + 1) % 2 or rs.rand() < 0.5 else (m + 1) // 2
      want = rm.RankHistogram(num_bins=nb, seed=rng_seed).compute_chunk(
          to_xr(ens), to_xr(truth))['z']
      got = om.RankHistogram(num_bins=nb, seed=rng_seed).compute_chunk(
          ens, truth)['z']
      assert tuple(got.dims) == tuple(want.dims), (seed, got.dims, want.dims)
      np.testing.assert_array_equal(np.asarray(got.data), np.asarray(want.data),
                                    err_msg=f'rank seed={seed} M={m} {edims}')
      n_checked += 1
      # Metric.compute: the time mean with and without skipna
      f = fz._dataset(rs, tdims, sizes, coords, dtype, 0.1)
      for skipna in (False, True):
        for name in ('MSE', 'Bias', 'RMSESqrtBeforeTimeAvg'):
          w = getattr(rm, name)().compute(to_xr(f), to_xr(truth),
                                          skipna=skipna)['z']
          g = getattr(om, name)().compute(f, truth, skipna=skipna)['z']
          compare(w, g, f'compute {name} seed={seed} skipna={skipna}')
          n_checked += 1

    special = {'maps': maps_case, 'thr': thr_case, 'spectrum': spectrum_case,
               'seeps': seeps_case, 'rank': rank_case}
    for seed in range(n_cases):
      if family in special:
        special[family](seed)
        continue
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
  env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')  # nothing into /root/reference
  env['PYTHONPATH'] = os.pathsep.join(
      [os.path.join(ROOT, 'oracle', 'refshim'), REFERENCE, ROOT,
       env.get('PYTHONPATH', '')])
  return subprocess.run([sys.executable, '-c', SCRIPT, family, str(n)],
                        env=env, cwd=ROOT, capture_output=True, text=True,
                        timeout=1200)


@pytest.mark.parametrize('family,n', [('det', 160), ('ens', 60), ('maps', 40),
                                      ('thr', 30), ('spectrum', 24),
                                      ('seeps', 20), ('rank', 30)])
def test_oracle_equals_the_reference_on_random_cases(family, n):
  res = _run(family, n)
  assert res.returncode == 0 and f'FUZZ-OK {family}' in res.stdout, (
      res.stdout[-1500:] + res.stderr[-5000:])


def test_oracle_equals_the_reference_at_full_size():
  """One BASELINE-size unit (13 x 721 x 1440 float32): the oracle and the
  reference's own metric code agree on the first (metric, region) evaluations
  -- every metric on the global region, then MSE / RMSE on the tropics -- to the
  printed digits of the summed results (tools/cpu_reference_vs_port.py)."""
  env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
  res = subprocess.run([sys.executable, os.path.join(
      ROOT, 'tools', 'cpu_reference_vs_port.py'), '7'], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
  assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
  sums = [line.split('checksum')[1].strip(' )\n')
          for line in res.stdout.splitlines() if 'checksum' in line]
  assert len(sums) == 2 and sums[0] == sums[1], res.stdout
