"""GPU parity of EXACTLY the launches bench.py times, against the NumPy oracle.

bench.py (BASELINE configs[1]) calls engine.stream_reduce with 13-level units of
721 x 1440 float32, slab tables on forecast / truth / climatology, 32 rows per
chunk, the 13 predefined regions and MODE_DET_ACC; `--workload ensemble`
(configs[2]) calls engine.ensemble_reduce with 50 members, 8-row chunks and slab
tables.  The other full-size tests use different chunk geometries or check
properties; here every (metric, region, level) number of those two launches is
compared with the oracle's own functions (oracle/metrics_np.py restating
/root/reference/weatherbench2/metrics.py:141-163, 175-414, 781-846).

Also: float32 latitude coordinates (the real 0.25-degree ERA5 case,
metrics.py:41,57 -- the weights inherit the coordinate dtype).
"""
import numpy as np
import pytest

from oracle import metrics_np as om
from oracle.named import DS, NA
from tests import helpers

pytestmark = pytest.mark.gpu

N_LEV, N_LAT, N_LON = 13, 721, 1440
LAT = np.linspace(-90, 90, N_LAT)
LON = np.linspace(0, 360, N_LON, endpoint=False)
LEVELS = np.array([50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925,
                   1000])


@pytest.fixture(scope='module')
def torch_dev():
  import torch
  if not torch.cuda.is_available():
    pytest.fail('-m gpu tests need a HIP device')
  return torch.device('cuda')


def test_deterministic_bench_launch_matches_oracle(torch_dev):
  """2 units x 13 levels from 3-unit pools, gathered through slab tables the way
  bench.py does (forecast consecutive, truth / climatology permuted), 32 rows
  per chunk, 13 regions: MSE, RMSE, MAE, Bias, ACC for every region and level."""
  import torch
  from weatherbench2_amd import _lib, engine, plan as plan_lib
  dev = torch_dev
  units, pool = 2, 3
  rs = np.random.RandomState(2024)
  mk = lambda: rs.normal(size=(pool, N_LEV, N_LAT, N_LON)).astype(np.float32)
  fpool, tpool, cpool = mk(), mk(), mk()
  f_unit = np.array([0, 1])
  t_unit = np.array([2, 0])   # truth gathered from other pool entries
  c_unit = np.array([1, 2])   # climatology rows ((dayofyear, hour) gather)
  lev = np.arange(N_LEV)
  tab = lambda u: torch.as_tensor((u[:, None] * N_LEV + lev[None]).reshape(-1),
                                  dtype=torch.int64, device=dev)
  regions = helpers.predefined_regions(oracle=False)
  pl = plan_lib.build_plan(LAT, LON, plan_lib.LATLON, regions, dev,
                           rows_per_chunk=32)  # bench.py's chunk geometry
  to_dev = lambda a: torch.as_tensor(a, device=dev).reshape(-1, N_LAT, N_LON)
  n_outer = units * N_LEV
  metrics, _ = engine.stream_reduce(
      pl, _lib.MODE_DET_ACC, [to_dev(fpool), to_dev(tpool), to_dev(cpool)],
      [tab(f_unit), tab(t_unit), tab(c_unit)], n_outer, skipna=False)
  got = metrics.cpu().numpy()  # [metric, region, outer]

  oregions = helpers.predefined_regions(oracle=True)
  dims = ('time', 'level', 'latitude', 'longitude')
  # the climatology gather as the reference does it: sel(dayofyear, hour) of
  # the forecast's valid time (metrics.py:394-404)
  ccoords = {'hour': np.array([0]), 'dayofyear': np.arange(1, pool + 1),
             'level': LEVELS, 'latitude': LAT, 'longitude': LON}
  clim = DS({'z': NA(cpool[None], ('hour', 'dayofyear') + dims[1:])}, ccoords)
  osuite = {'mse': om.MSE(), 'rmse': om.RMSESqrtBeforeTimeAvg(),
            'mae': om.MAE(), 'bias': om.Bias(), 'acc': om.ACC(clim)}
  for u in range(units):
    day = np.datetime64('2020-01-01T00', 'ns') + np.timedelta64(
        int(c_unit[u]), 'D')
    coords = {'time': np.array([day]), 'level': LEVELS, 'latitude': LAT,
              'longitude': LON}
    f = DS({'z': NA(fpool[f_unit[u]][None], dims)}, coords)
    t = DS({'z': NA(tpool[t_unit[u]][None], dims)}, coords)
    sl = slice(u * N_LEV, (u + 1) * N_LEV)
    for ri, rname in enumerate(pl.region_names):
      for mname, metric in osuite.items():
        want = metric.compute_chunk(f, t, region=oregions[rname])['z'].data[0]
        helpers.assert_close(got[_lib.METRIC_INDEX[mname], ri, sl], want,
                             rtol=1e-9, atol=1e-12,
                             err_msg=f'unit {u} {mname}/{rname}')


def test_ensemble_bench_launch_matches_oracle(torch_dev):
  """50 members x 2 slabs (tables pick them out of a 3-slab pool), 8-row chunks,
  13 regions: the eight ensemble metrics of K3 for every region."""
  import torch
  from weatherbench2_amd import _lib, engine, plan as plan_lib
  dev = torch_dev
  m, pool, n_slab = 50, 3, 2
  rs = np.random.RandomState(77)
  ens = rs.normal(size=(m, pool, N_LAT, N_LON)).astype(np.float32)
  truth = rs.normal(size=(pool, N_LAT, N_LON)).astype(np.float32)
  e_tab, t_tab = np.array([2, 0]), np.array([1, 2])
  regions = helpers.predefined_regions(oracle=False)
  pl = plan_lib.build_plan(LAT, LON, plan_lib.LATLON, regions, dev,
                           rows_per_chunk=plan_lib.ENSEMBLE_ROWS_PER_CHUNK)
  ti = lambda a: torch.as_tensor(a, dtype=torch.int64, device=dev)
  metrics, _ = engine.ensemble_reduce(
      pl, torch.as_tensor(ens, device=dev), pool * N_LAT * N_LON, m,
      ti(e_tab), torch.as_tensor(truth, device=dev), ti(t_tab), n_slab, False)
  got = metrics.cpu().numpy()  # [metric, region, slab]

  oregions = helpers.predefined_regions(oracle=True)
  dims = ('realization', 'latitude', 'longitude')
  coords = {'latitude': LAT, 'longitude': LON}
  idx = _lib.ENS_METRIC_INDEX
  for s in range(n_slab):
    f = DS({'z': NA(ens[:, e_tab[s]], dims)}, coords)
    t = DS({'z': NA(truth[t_tab[s]], dims[1:])}, coords)
    # the pointwise fields once (metrics.py:532-565, 781-824), then the
    # reference's spatial average per region (metrics.py:141-163)
    mean = f.mean('realization', skipna=False)
    var = f.var('realization', skipna=False, ddof=1)
    fields = {
        'crps_spread': om.pointwise_crps_spread(f, 'realization', False),
        'crps_skill': om.pointwise_crps_skill(f, t, 'realization', False),
        'ensemble_mean_mse': (t - mean) ** 2,
        'ensemble_variance': var,
        'debiased_ensemble_mean_mse': om.debiased_ensemble_mean_mse(
            f, t, 'realization', False),
    }
    std2 = f.std('realization', skipna=False, ddof=1) ** 2
    for ri, rname in enumerate(pl.region_names):
      sa = lambda ds: float(np.asarray(
          om.spatial_average(ds, oregions[rname], False)['z'].data))
      want = {k: sa(v) for k, v in fields.items()}
      want['crps'] = want['crps_skill'] - 0.5 * want['crps_spread']
      want['ensemble_mean_rmse'] = np.sqrt(want['ensemble_mean_mse'])
      want['ensemble_stddev'] = np.sqrt(sa(std2))
      for name, w in want.items():
        # float32 member statistics: both sides round identically per point;
        # the spatial sums are fp64 on both sides
        helpers.assert_close(got[idx[name], ri, s], w, rtol=2e-6, atol=1e-7,
                             err_msg=f'slab {s} {name}/{rname}')


@pytest.mark.parametrize('n_lat,n_lon', [(721, 1440), (181, 360)])
def test_float32_latitude_era5_style(torch_dev, n_lat, n_lon):
  """ERA5 ships float32 coordinates with latitude DEcreasing: after the
  evaluation-time flip (evaluation.py:41-47) the weights are computed in
  float32 (metrics.py:41, 57).  The reference's einsum then accumulates in
  float32 (noise ~1e-5 at a million points, result float32); the product
  applies the SAME float32-valued weights, sums in float64 and returns the
  reference's dtype (float32 here: the float64 sum rounded once).  Checked (a)
  against a float64 evaluation with those weights to one float32 ulp and (b)
  against the oracle's float32 path at the float32-summation noise."""
  from weatherbench2_amd import evaluation, metrics as gm
  from weatherbench2_amd import plan as plan_lib
  rs = np.random.RandomState(5)
  lat32 = np.linspace(90, -90, n_lat).astype(np.float32)   # decreasing
  lon32 = np.linspace(0, 360, n_lon, endpoint=False).astype(np.float32)
  dims = ('time', 'level', 'latitude', 'longitude')
  f = rs.normal(size=(2, 2, n_lat, n_lon)).astype(np.float32)
  t = rs.normal(size=(2, 2, n_lat, n_lon)).astype(np.float32)
  coords = {'time': np.array(['2020-01-01T00', '2020-01-01T06'],
                             dtype='datetime64[ns]'),
            'level': np.array([500, 850]), 'latitude': lat32,
            'longitude': lon32}
  g = helpers.to_gpu_dataset
  gf = evaluation.make_latitude_increasing(g(DS({'z': NA(f, dims)}, coords)))
  gt = evaluation.make_latitude_increasing(g(DS({'z': NA(t, dims)}, coords)))
  assert np.asarray(gf.coords['latitude']).dtype == np.float32
  assert (np.diff(np.asarray(gf.coords['latitude'])) > 0).all()
  # the oracle sees the flipped arrays (what evaluation.py hands the metrics)
  ocoords = dict(coords, latitude=lat32[::-1].copy())
  of = DS({'z': NA(f[:, :, ::-1].copy(), dims)}, ocoords)
  ot = DS({'z': NA(t[:, :, ::-1].copy(), dims)}, ocoords)
  w32 = plan_lib.get_lat_weights(lat32[::-1])
  assert w32.dtype == np.float32
  np.testing.assert_array_equal(w32, om.get_lat_weights(lat32[::-1]).data)
  w64 = w32.astype(np.float64)[None, None, :, None]
  d = (of['z'].data - ot['z'].data)  # float32 elementwise, like numpy
  oregions = {'global': None,
              'tropics': helpers.predefined_regions(True)['tropics']}
  gregions = {'global': None,
              'tropics': helpers.predefined_regions(False)['tropics']}
  lat_sel = {'global': np.ones(n_lat, bool),
             'tropics': (lat32[::-1] >= -20) & (lat32[::-1] <= 20)}
  exact = {
      'mse': lambda r: ((d * d).astype(np.float64) * w64)[:, :, r].sum((2, 3))
      / (w64[:, :, r].sum() * n_lon),
      'mae': lambda r: (np.abs(d).astype(np.float64) * w64)[:, :, r].sum((2, 3))
      / (w64[:, :, r].sum() * n_lon),
      'bias': lambda r: (d.astype(np.float64) * w64)[:, :, r].sum((2, 3))
      / (w64[:, :, r].sum() * n_lon),
  }
  pairs = {'mse': (gm.MSE(), om.MSE()), 'mae': (gm.MAE(), om.MAE()),
           'bias': (gm.Bias(), om.Bias())}
  for rname in oregions:
    for mname, (gmet, omet) in pairs.items():
      got = gmet.compute_chunk(gf, gt, region=gregions[rname])['z'].values
      # the reference's dtype for float32 data on float32 coordinates and a
      # slice region (metrics._reference_result_dtype); the VALUE is the
      # float64 sum rounded once: within one float32 ulp of the exact one
      assert got.dtype == np.float32
      helpers.assert_close(got, exact[mname](lat_sel[rname]), rtol=1.2e-7,
                           atol=1e-9, err_msg=f'{mname}/{rname} (fp64 sums)')
      want = omet.compute_chunk(of, ot, region=oregions[rname])['z'].data
      assert want.dtype == np.float32  # the reference's result dtype
      # float32 einsum noise of the reference path; Bias cancels to ~1e-3, so
      # its absolute floor is eps32 * mean|w d| ~ 1e-7 * sqrt(n)
      helpers.assert_close(got, want, rtol=5e-5, atol=2e-6,
                           err_msg=f'{mname}/{rname} (oracle float32 path)')
