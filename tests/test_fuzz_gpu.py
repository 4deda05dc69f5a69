"""Seeded differential fuzzing of the fused passes against the oracle (-m gpu):
random grid sizes (odd widths take the scalar-load path), dim orders, dtypes,
NaN patterns, skipna, region sets (label slices, lists of slices,
extra-tropics, land masks with thresholds, combinations), truth broadcast
over extra forecast dims -- 160 deterministic + 60 ensemble cases here, then maps / Gaussian / rank
histogram (40), threshold family (30), spectrum (24), reductions (30) and
SEEPS with values exactly on the category boundaries (20)."""
import numpy as np
import pytest

from oracle import metrics_np as om
from oracle import regions_np as oreg
from oracle.named import DS, NA
from tests import helpers

pytestmark = pytest.mark.gpu


def _grid(rs):
  n_lat = int(rs.randint(2, 24))
  n_lon = int(rs.randint(2, 48))
  lat = np.sort(rs.uniform(-89, 89, n_lat)) if rs.rand() < 0.3 else (
      np.linspace(-90, 90, n_lat))
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  return lat, lon


def _random_region(rs, lat, lon):
  kind = rs.randint(0, 6)
  def lat_slice():
    a, b = np.sort(rs.uniform(-95, 95, 2))
    return slice(float(a), float(b))
  def lon_slice():
    a, b = np.sort(rs.uniform(-10, 370, 2))
    return slice(float(a), float(b))
  def land():
    frac = np.clip(rs.rand(len(lat), len(lon)) * 1.6 - 0.3, 0, 1)
    thr = None if rs.rand() < 0.5 else 0.5
    return oreg.LandRegion(NA(frac, ('latitude', 'longitude')), lat, lon, thr)
  if kind == 0:
    return oreg.SliceRegion()
  if kind == 1:
    return oreg.SliceRegion(lat_slice=lat_slice(), lon_slice=lon_slice())
  if kind == 2:
    return oreg.SliceRegion(lat_slice=[lat_slice(), lat_slice()],
                            lon_slice=[lon_slice(), lon_slice()])
  if kind == 3:
    return oreg.ExtraTropicalRegion()
  if kind == 4:
    return land()
  return oreg.CombinedRegion([oreg.SliceRegion(lat_slice=lat_slice()), land()])


def _dataset(rs, dims, sizes, coords, dtype, nan_frac, inf_frac=0.0):
  shape = tuple(sizes[d] for d in dims)
  x = rs.standard_normal(shape).astype(dtype)
  if nan_frac:
    x[rs.rand(*shape) < nan_frac] = np.nan
  if inf_frac:  # a few +-inf: inf - inf = NaN, inf ** 2 = inf, like NumPy
    hit = rs.rand(*shape) < inf_frac
    x[hit] = np.where(rs.rand(int(hit.sum())) < 0.5, np.inf, -np.inf)
  return DS({'z': NA(x, dims)}, {d: coords[d] for d in dims})


@pytest.mark.parametrize('seed', range(160))
def test_deterministic_family_fuzz(seed):
  from weatherbench2_amd import metrics as gm
  rs = np.random.RandomState(1000 + seed)
  lat, lon = _grid(rs)
  sizes = {'time': int(rs.randint(1, 4)), 'level': int(rs.randint(1, 4)),
           'prediction_timedelta': int(rs.randint(1, 3)),
           'latitude': len(lat), 'longitude': len(lon)}
  t0 = np.datetime64('2020-03-01T00', 'ns')
  coords = {'time': t0 + np.arange(sizes['time']) * np.timedelta64(6, 'h'),
            'level': np.array([500, 700, 850])[:sizes['level']],
            'prediction_timedelta': (np.arange(sizes['prediction_timedelta'])
                                     * np.timedelta64(6, 'h')),
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
  forecast = _dataset(rs, fdims, sizes, coords, dtype, nan_frac, inf_frac)
  truth = _dataset(rs, tdims, sizes, coords, dtype, nan_frac / 2, inf_frac / 2)
  regions = {f'r{i}': _random_region(rs, lat, lon) for i in range(3)}
  # several distinct masks in one set are grouped by the product itself
  g = helpers.to_gpu_dataset
  g_regions = {k: helpers.to_gpu_region(v) for k, v in regions.items()}
  pairs = [(om.MSE(), gm.MSE()), (om.MAE(), gm.MAE()), (om.Bias(), gm.Bias()),
           (om.RMSESqrtBeforeTimeAvg(), gm.RMSESqrtBeforeTimeAvg())]
  if rs.rand() < 0.5:
    cdims = ('hour', 'dayofyear', 'level') + tuple(spatial)
    csizes = {'hour': 4, 'dayofyear': 3, **sizes}
    ccoords = {'hour': np.array([0, 6, 12, 18]),
               'dayofyear': np.array([60, 61, 62]), **coords}
    clim = _dataset(rs, cdims, csizes, ccoords, dtype, 0.0)
    pairs.append((om.ACC(clim), gm.ACC(g(clim))))
  # float32 inputs: the elementwise values are float32 on both sides and every
  # spatial sum is float64 on both sides (the latitude weights are float64), so
  # only the summation order differs: far inside 1e-9, with an absolute floor
  # of a few ulp(float64) of sum|w x| / sum w for the means that cancel (Bias)
  rtol = 1e-9
  with gm.fused_regions(g_regions):
    for oc, gc in pairs:
      for rname, region in regions.items():
        want = oc.compute_chunk(forecast, truth, region=region,
                                skipna=skipna)['z']
        got = gc.compute_chunk(g(forecast), g(truth), region=g_regions[rname],
                               skipna=skipna)['z']
        assert got.dims == want.dims, (seed, type(oc).__name__, rname)
        helpers.assert_close(
            got.values, want.data, rtol=rtol, atol=1e-12,
            err_msg=f'seed={seed} {type(oc).__name__} {rname} '
            f'{fdims} {dtype.__name__} skipna={skipna}')


@pytest.mark.parametrize('seed', range(60))
def test_ensemble_family_fuzz(seed):
  from weatherbench2_amd import metrics as gm
  rs = np.random.RandomState(2000 + seed)
  lat, lon = _grid(rs)
  m = int(rs.choice([1, 2, 3, 6, 17, 33, 70, 140]))
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
  forecast = _dataset(rs, fdims, sizes, coords, dtype, nan_frac)
  truth = _dataset(rs, tdims, sizes, coords, dtype, nan_frac / 2)
  regions = {f'r{i}': _random_region(rs, lat, lon) for i in range(2)}
  g = helpers.to_gpu_dataset
  g_regions = {k: helpers.to_gpu_region(v) for k, v in regions.items()}
  names = ('CRPS', 'CRPSSpread', 'CRPSSkill', 'EnsembleMeanMSE',
           'EnsembleVariance', 'DebiasedEnsembleMeanMSE')
  # float32 members: the pointwise statistics follow numpy operation by
  # operation in float32, the spatial sums are float64 on both sides
  rtol = 1e-6 if dtype == np.float32 else 1e-9
  with gm.fused_regions(g_regions):
    for name in names:
      for rname, region in regions.items():
        want = getattr(om, name)().compute_chunk(forecast, truth,
                                                 region=region,
                                                 skipna=skipna)['z']
        got = getattr(gm, name)().compute_chunk(
            g(forecast), g(truth), region=g_regions[rname], skipna=skipna)['z']
        assert got.dims == want.dims, (seed, name, rname)
        helpers.assert_close(
            got.values, want.data, rtol=rtol,
            atol=1e-9 if dtype == np.float32 else 1e-12,
            err_msg=f'seed={seed} {name} {rname} M={m} {fdims} '
            f'{dtype.__name__} skipna={skipna}')


def _values(da):
  v = da.data
  return v.cpu().numpy() if hasattr(v, 'cpu') else np.asarray(v)


@pytest.mark.parametrize('seed', range(40))
def test_maps_gaussian_and_rank_histogram_fuzz(seed):
  """The map-valued and tier-2 families on random layouts: Spatial{MSE,MAE,
  Bias}, SpatialCRPS / SpatialEnsembleVariance, GaussianCRPS / Variance and
  the (tie-free) rank histogram."""
  from weatherbench2_amd import metrics as gm
  rs = np.random.RandomState(3000 + seed)
  lat, lon = _grid(rs)
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
  tol = dict(rtol=2e-5, atol=1e-6) if dtype == np.float32 else dict(
      rtol=1e-9, atol=1e-12)
  g = helpers.to_gpu_dataset
  tag = f'seed={seed} {dtype.__name__} skipna={skipna}'
  # deterministic maps
  forecast = _dataset(rs, ddims, sizes, coords, dtype, nan_frac)
  truth = _dataset(rs, ddims, sizes, coords, dtype, 0.0)
  for name in ('SpatialMSE', 'SpatialMAE', 'SpatialBias'):
    want = getattr(om, name)().compute_chunk(forecast, truth)['z']
    got = getattr(gm, name)().compute_chunk(g(forecast), g(truth))['z']
    assert set(got.dims) == set(want.dims)
    helpers.assert_close(_values(got), want.transpose(*got.dims).data,
                         err_msg=f'{name} {tag}', **tol)
  # ensemble maps + rank histogram
  ens = _dataset(rs, edims, sizes, coords, dtype, nan_frac)
  for name in ('SpatialCRPS', 'SpatialEnsembleVariance'):
    want = getattr(om, name)().compute_chunk(ens, truth, skipna=skipna)['z']
    got = getattr(gm, name)().compute_chunk(g(ens), g(truth),
                                            skipna=skipna)['z']
    helpers.assert_close(_values(got), want.transpose(*got.dims).data,
                         err_msg=f'{name} {tag} M={m}', **tol)
  clean = _dataset(rs, edims, sizes, coords, dtype, 0.0)
  want = om.RankHistogram(break_ties_randomly=False).compute_chunk(
      clean, truth)['z']
  got = gm.RankHistogram(break_ties_randomly=False).compute_chunk(
      g(clean), g(truth))['z']
  np.testing.assert_array_equal(_values(got), want.transpose(*got.dims).data,
                                err_msg=f'RankHistogram {tag} M={m}')
  # Gaussian family: mean / std variables
  mean = _dataset(rs, ddims, sizes, coords, dtype, nan_frac)
  std = _dataset(rs, ddims, sizes, coords, dtype, 0.0)
  gf = DS({'z': mean['z'], 'z_std': NA(np.abs(std['z'].data) + dtype(0.1),
                                       std['z'].dims)}, mean.coords)
  for name in ('GaussianCRPS', 'GaussianVariance'):
    want = getattr(om, name)().compute_chunk(gf, truth, skipna=skipna)['z']
    got = getattr(gm, name)().compute_chunk(g(gf), g(truth), skipna=skipna)['z']
    assert got.dims == want.dims
    helpers.assert_close(got.values, want.data, err_msg=f'{name} {tag}', **tol)


@pytest.mark.parametrize('seed', range(30))
def test_threshold_family_fuzz(seed):
  """Gaussian and ensemble threshold metrics with a climatological Gaussian
  quantile threshold on random layouts (thresholds.py:151-187; metrics.py:
  975-1158, 1524-1891)."""
  from oracle import thresholds_np as oth
  from weatherbench2_amd import metrics as gm
  from weatherbench2_amd import thresholds as gth
  rs = np.random.RandomState(4000 + seed)
  lat, lon = _grid(rs)
  m = int(rs.choice([1, 2, 4, 7, 31]))
  n_time = int(rs.randint(1, 4))
  t0 = np.datetime64('2021-02-27T00', 'ns')
  sizes = {'realization': m, 'time': n_time, 'level': int(rs.randint(1, 3)),
           'latitude': len(lat), 'longitude': len(lon), 'dayofyear': 6}
  coords = {'realization': np.arange(m),
            'time': t0 + np.arange(n_time) * np.timedelta64(24, 'h'),
            'level': np.array([500, 850])[:sizes['level']], 'latitude': lat,
            'longitude': lon, 'dayofyear': 57 + np.arange(6)}
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
  truth = _dataset(rs, ddims, sizes, coords, dtype, nan_frac / 2)
  cm = _dataset(rs, cdims, sizes, coords, dtype, 0.0)['z']
  cs = _dataset(rs, cdims, sizes, coords, dtype, 0.0)['z']
  clim = DS({'z': NA(cm.data * dtype(0.3), cdims),
             'z_std': NA(np.abs(cs.data) * dtype(0.5) + dtype(0.5), cdims)},
            {d: coords[d] for d in cdims})
  g = helpers.to_gpu_dataset
  qs = (0.25, 0.7)
  oths = [oth.GaussianQuantileThreshold(clim, q) for q in qs]
  gths = [gth.GaussianQuantileThreshold(climatology=g(clim), quantile=q)
          for q in qs]
  region = _random_region(rs, lat, lon)
  g_region = helpers.to_gpu_region(region)
  tol = dict(rtol=3e-5, atol=2e-6) if dtype == np.float32 else dict(
      rtol=1e-9, atol=1e-12)
  tag = f'seed={seed} {dtype.__name__} skipna={skipna} M={m}'
  ens = _dataset(rs, edims, sizes, coords, dtype, nan_frac)
  for name in ('EnsembleBrierScore', 'DebiasedEnsembleBrierScore',
               'EnsembleIgnoranceScore', 'EnsembleRPS'):
    want = getattr(om, name)(thresholds=oths).compute_chunk(
        ens, truth, region=region, skipna=skipna)['z']
    got = getattr(gm, name)(thresholds=gths).compute_chunk(
        g(ens), g(truth), region=g_region, skipna=skipna)['z']
    assert got.dims == want.dims, (name, tag)
    helpers.assert_close(got.values, want.data, err_msg=f'{name} {tag}', **tol)
  mean = _dataset(rs, ddims, sizes, coords, dtype, nan_frac)
  std = _dataset(rs, ddims, sizes, coords, dtype, 0.0)
  gf = DS({'z': mean['z'], 'z_std': NA(np.abs(std['z'].data) + dtype(0.2),
                                       std['z'].dims)}, mean.coords)
  for name in ('GaussianBrierScore', 'GaussianIgnoranceScore', 'GaussianRPS'):
    want = getattr(om, name)(thresholds=oths).compute_chunk(
        gf, truth, region=region, skipna=skipna)['z']
    got = getattr(gm, name)(thresholds=gths).compute_chunk(
        g(gf), g(truth), region=g_region, skipna=skipna)['z']
    assert got.dims == want.dims, (name, tag)
    helpers.assert_close(got.values, want.data, err_msg=f'{name} {tag}', **tol)


@pytest.mark.parametrize('seed', range(24))
def test_spectrum_fuzz(seed):
  """ZonalEnergySpectrum on random row lengths (even / odd, fused and rocFFT
  paths), dim orders, dtypes and NaN rows, plain and with the fused time
  mean."""
  from oracle import spectrum_np
  from weatherbench2_amd import derived_variables as dv
  from weatherbench2_amd import xarray_lite as xl
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
  ds = xl.Dataset({'z': xl.DataArray(np.ascontiguousarray(
      np.transpose(x, perm)), dims)},
                  {'time': np.arange(n_time), 'level': np.arange(n_lev),
                   'latitude': lat, 'longitude': lon})
  want, freq, _ = spectrum_np.zonal_energy_spectrum(x, lat, lon, lat_axis=2,
                                                    lon_axis=3)
  got = dv.ZonalEnergySpectrum('z').compute(ds)
  assert got.dims == tuple(d for d in dims if d != 'longitude') + (
      'zonal_wavenumber',)
  g = np.asarray(got.transpose('time', 'level', 'latitude',
                               'zonal_wavenumber').values)
  scale = np.abs(want).sum(-1, keepdims=True)
  tol = 3e-6 if dtype == np.float32 else 1e-12
  assert np.max(np.abs(g - want) / np.where(scale > 0, scale, 1)) < tol, (
      seed, n_lon, dims, dtype)
  np.testing.assert_allclose(got.coords['frequency'].values, freq, rtol=1e-12)
  # time mean with a NaN row
  if n_time > 1:
    xn = x.copy()
    xn[0, 0, 0, 0] = np.nan
    dsn = xl.Dataset({'z': xl.DataArray(np.ascontiguousarray(
        np.transpose(xn, perm)), dims)}, ds.coords)
    wn, _, _ = spectrum_np.zonal_energy_spectrum(xn, lat, lon, lat_axis=2,
                                                 lon_axis=3)
    for skipna in (True, False):
      fused = dv.ZonalEnergySpectrum('z').compute(dsn, time_mean_dim='time',
                                                  skipna=skipna)
      f = np.asarray(fused.transpose('level', 'latitude',
                                     'zonal_wavenumber').values)
      import warnings
      with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        w = np.nanmean(wn, 0) if skipna else wn.mean(0)
      sc = np.nansum(np.abs(w), -1, keepdims=True)
      err = np.abs(f - w) / np.where(sc > 0, sc, 1)
      assert np.array_equal(np.isnan(f), np.isnan(w)), (seed, skipna)
      if np.isfinite(err).any():  # (a single all-NaN row has nothing to compare)
        assert np.nanmax(err) < tol, (seed, n_lon, skipna)


@pytest.mark.parametrize('seed', range(30))
def test_reductions_fuzz(seed):
  """reductions.averages / ensemble_mean / statistical_moments on random dim
  orders and reduction sets (merged, non-adjacent, weighted, NaNs)."""
  from oracle import reductions_np as ored
  from weatherbench2_amd import reductions
  rs = np.random.RandomState(6000 + seed)
  lat, lon = _grid(rs)
  sizes = {'realization': int(rs.randint(1, 5)), 'time': int(rs.randint(1, 5)),
           'level': int(rs.randint(1, 4)), 'latitude': len(lat),
           'longitude': len(lon)}
  coords = {'realization': np.arange(sizes['realization']),
            'time': np.arange(sizes['time']),
            'level': np.arange(sizes['level']), 'latitude': lat,
            'longitude': lon}
  dims = list(sizes)
  rs.shuffle(dims)
  dims = tuple(dims)
  dtype = np.float32 if rs.rand() < 0.6 else np.float64
  skipna = bool(rs.rand() < 0.5)
  ds = _dataset(rs, dims, sizes, coords, dtype, 0.06 if skipna else 0.0)
  g = helpers.to_gpu_dataset
  k = int(rs.randint(1, 4))
  red = list(rs.choice(list(sizes), size=k, replace=False))
  tol = dict(rtol=3e-6, atol=1e-6) if dtype == np.float32 else dict(
      rtol=1e-12, atol=1e-12)
  want = ored.averages(ds, red, skipna=skipna)['z']
  got = reductions.averages(g(ds), red, skipna=skipna)['z']
  assert set(got.dims) == set(want.dims)
  helpers.assert_close(_values(got), want.transpose(*got.dims).data,
                       err_msg=f'seed={seed} averages over {red} of {dims}',
                       **tol)
  want = ored.ensemble_mean(ds, skipna=skipna)['z']
  got = reductions.ensemble_mean(g(ds), skipna=skipna)['z']
  helpers.assert_close(_values(got), want.transpose(*got.dims).data,
                       err_msg=f'seed={seed} ensemble_mean {dims}', **tol)
  want = ored.statistical_moments(ds)
  got = reductions.statistical_moments(g(ds))
  for key in ('z_zeroth', 'z_first', 'z_second'):
    helpers.assert_close(_values(got[key]),
                         want[key].transpose(*got[key].dims).data,
                         err_msg=f'seed={seed} {key} {dims}', **tol)


@pytest.mark.parametrize('seed', range(20))
def test_seeps_fuzz(seed):
  """SEEPS / SpatialSEEPS on random precipitation with values sitting EXACTLY
  on the dry and wet thresholds (the category boundaries of metrics.py:
  444-460), NaNs, masked dry fractions and random regions."""
  from weatherbench2_amd import metrics as gm
  rs = np.random.RandomState(7000 + seed)
  lat, lon = _grid(rs)
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
    x = np.where(pick < 0.1, dry, x)                      # exactly dry
    x = np.where((pick >= 0.1) & (pick < 0.2), wet[0, 0][None], x)  # exactly wet
    x = np.where(pick > 0.97, np.nan, x)
    return x.astype(dtype)
  dims = ('time',) + spatial
  coords = {'time': time, 'latitude': lat, 'longitude': lon,
            'valid_time': NA(time, ('time',))}  # SEEPS reads da.valid_time
  forecast = DS({name: NA(precip(), dims)}, coords)
  truth = DS({name: NA(precip(), dims)}, coords)
  cdims = ('hour', 'dayofyear') + spatial
  clim = DS({name + '_seeps_threshold': NA(wet, cdims),
             name + '_seeps_dry_fraction': NA(frac, cdims)},
            {'hour': np.array([0, 6, 12, 18]),
             'dayofyear': np.array([60, 61, 62]), 'latitude': lat,
             'longitude': lon})
  g = helpers.to_gpu_dataset
  region = _random_region(rs, lat, lon)
  want = om.SEEPS(climatology=clim).compute_chunk(forecast, truth,
                                                  region=region)[name]
  got = gm.SEEPS(climatology=g(clim)).compute_chunk(
      g(forecast), g(truth), region=helpers.to_gpu_region(region))[name]
  assert got.dims == want.dims
  helpers.assert_close(got.values, want.data, rtol=2e-6, atol=1e-7,
                       err_msg=f'seed={seed} SEEPS')
  wmap = om.SpatialSEEPS(climatology=clim).compute_chunk(forecast, truth)[name]
  gmap = gm.SpatialSEEPS(climatology=g(clim)).compute_chunk(g(forecast),
                                                            g(truth))[name]
  helpers.assert_close(_values(gmap), wmap.transpose(*gmap.dims).data,
                       rtol=2e-6, atol=1e-7, err_msg=f'seed={seed} SpatialSEEPS')
