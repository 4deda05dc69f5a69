"""wb2_det_suite_step (one chunk of the deterministic suite per C-ABI call:
K1 -> K2 -> running temporal mean) against the three separate entry points:
the same kernels, so the SAME BITS -- by slab numbers, by slab addresses, with
a 2-D weight field, through `bind`, and against the NumPy oracle
(oracle/metrics_np.py restating /root/reference/weatherbench2/metrics.py:
141-163, 236-414 and evaluation.py:735-744's mean)."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

N_LEV, N_LAT, N_LON = 3, 181, 360
LAT = np.linspace(-90, 90, N_LAT)
LON = np.linspace(0, 360, N_LON, endpoint=False)


@pytest.fixture(scope='module')
def dev():
  import torch
  if not torch.cuda.is_available():
    pytest.fail('-m gpu tests need a HIP device')
  return torch.device('cuda')


def _pools(dev, pool, seed=5):
  import torch
  g = torch.Generator(device=dev).manual_seed(seed)
  return [torch.randn((pool * N_LEV, N_LAT, N_LON), generator=g, device=dev)
          for _ in range(3)]


def _tables(dev, units, pool, shift):
  import torch
  lev = torch.arange(N_LEV, device=dev)
  out = []
  for j in range(3):
    u = (shift + (2 * j + 1) * torch.arange(units, device=dev)) % pool
    out.append((u[:, None] * N_LEV + lev[None]).reshape(-1).contiguous())
  return out


def _regions(with_field: bool):
  from weatherbench2_amd import regions as R
  from weatherbench2_amd import xarray_lite as xl
  regions = helpers.predefined_regions(oracle=False)
  if with_field:
    rs = np.random.RandomState(3)
    lsm = (rs.uniform(size=(N_LAT, N_LON)) > 0.6).astype(np.float32)
    regions['land'] = R.LandRegion(land_sea_mask=xl.DataArray(
        lsm, ('latitude', 'longitude'), {'latitude': LAT, 'longitude': LON}))
  return regions


@pytest.mark.parametrize('with_field', [False, True])
@pytest.mark.parametrize('skipna', [False, True])
def test_suite_step_equals_the_three_calls(dev, with_field, skipna):
  import torch
  from weatherbench2_amd import _lib, engine, plan as plan_lib
  units, pool = 4, 6
  pl = plan_lib.build_plan(LAT, LON, plan_lib.LATLON, _regions(with_field),
                           dev, rows_per_chunk=16)
  inputs = _pools(dev, pool)
  if skipna:
    inputs[0][3, 10:20, 5:50] = float('nan')
  nr, n_outer = pl.n_region, units * N_LEV
  want_t = torch.zeros((_lib.NMETRIC * nr, N_LEV), dtype=torch.float64,
                       device=dev)
  want_c = torch.zeros_like(want_t)
  got_t, got_c = torch.zeros_like(want_t), torch.zeros_like(want_t)
  step = engine.SuiteStep(pl, _lib.MODE_DET_ACC, torch.float32, skipna,
                          n_outer)
  step.accumulate_into(got_t, got_c, (_lib.NMETRIC * nr, units, N_LEV),
                       skipna=skipna)
  for s in range(3):
    tabs = _tables(dev, units, pool, s)
    m, _ = engine.stream_reduce(pl, _lib.MODE_DET_ACC, inputs, tabs, n_outer,
                                skipna)
    engine.time_accumulate(m.view(_lib.NMETRIC * nr, units, N_LEV), 1, skipna,
                           want_t, want_c)
    got = step.run(inputs, tabs)
    assert torch.equal(torch.nan_to_num(got, nan=-7.0),
                       torch.nan_to_num(m, nan=-7.0))
  assert torch.equal(torch.nan_to_num(got_t, nan=-7.0),
                     torch.nan_to_num(want_t, nan=-7.0))
  assert torch.equal(got_c, want_c)
  assert got_c.max().item() == 3 * units


def test_bound_calls_and_address_form(dev):
  """bind() marshals once; the by-address form reads the same slabs through
  byte addresses -- both bit-identical to the table form."""
  import torch
  from weatherbench2_amd import _lib, engine, plan as plan_lib
  units, pool = 3, 5
  pl = plan_lib.build_plan(LAT, LON, plan_lib.LATLON, _regions(False), dev,
                           rows_per_chunk=32)
  inputs = _pools(dev, pool, seed=9)
  nr, n_outer = pl.n_region, units * N_LEV
  view = (_lib.NMETRIC * nr, units, N_LEV)
  acc = [torch.zeros((view[0], N_LEV), dtype=torch.float64, device=dev)
         for _ in range(6)]
  by_table = engine.SuiteStep(pl, _lib.MODE_DET_ACC, torch.float32, False,
                              n_outer)
  by_table.accumulate_into(acc[0], acc[1], view)
  bound = engine.SuiteStep(pl, _lib.MODE_DET_ACC, torch.float32, False, n_outer)
  bound.accumulate_into(acc[2], acc[3], view)
  by_addr = engine.SuiteStep(pl, _lib.MODE_DET_ACC, torch.float32, False,
                             n_outer, by_address=True)
  by_addr.accumulate_into(acc[4], acc[5], view)
  slab_bytes = N_LAT * N_LON * 4
  calls = []
  for s in range(4):
    tabs = _tables(dev, units, pool, s)
    by_table.run(inputs, tabs)
    calls.append(bound.bind(inputs, tabs))
    addr = [x.data_ptr() + tb * slab_bytes for x, tb in zip(inputs, tabs)]
    by_addr.run(None, addr)
  for call in calls:
    call()
  torch.cuda.synchronize()
  for k in (2, 4):
    assert torch.equal(acc[0], acc[k]) and torch.equal(acc[1], acc[k + 1])


def test_suite_step_matches_oracle(dev):
  """One step's metrics against the NumPy oracle for every region: MSE, RMSE,
  MAE, Bias and -- through a climatology with its (hour, dayofyear) gather --
  ACC (metrics.py:236-301, 319-330, 348-359, 377-414)."""
  import torch
  from oracle import metrics_np as om
  from oracle.named import DS, NA
  from weatherbench2_amd import _lib, engine, plan as plan_lib
  regions = helpers.predefined_regions(oracle=False)
  pl = plan_lib.build_plan(LAT, LON, plan_lib.LATLON, regions, dev,
                           rows_per_chunk=16)
  rs = np.random.RandomState(11)
  f, t, c = (rs.normal(size=(N_LEV, N_LAT, N_LON)).astype(np.float32)
             for _ in range(3))
  step = engine.SuiteStep(pl, _lib.MODE_DET_ACC, torch.float32, False, N_LEV)
  got = step.run([torch.as_tensor(x, device=dev) for x in (f, t, c)],
                 [None, None, None]).cpu().numpy()
  coords = {'level': np.arange(N_LEV), 'latitude': LAT, 'longitude': LON}
  dims = ('level', 'latitude', 'longitude')
  fd, td = (DS({'z': NA(x, dims)}, coords) for x in (f, t))
  oregions = helpers.predefined_regions(oracle=True)
  # ACC as the reference computes it: the climatology selected at the valid
  # time's (dayofyear, hour), anomalies in float32, three spatial averages
  day = np.datetime64('2020-01-03T00', 'ns')
  tcoords = dict(coords, time=np.array([day]))
  tdims = ('time',) + dims
  ft, tt = (DS({'z': NA(x[None], tdims)}, tcoords) for x in (f, t))
  clim = DS({'z': NA(np.stack([c + 1, c + 2, c, c - 1])[None],
                     ('hour', 'dayofyear') + dims)},
            dict(coords, hour=np.array([0]), dayofyear=np.arange(1, 5)))
  suite = {'mse': om.MSE(), 'rmse': om.RMSESqrtBeforeTimeAvg(),
           'mae': om.MAE(), 'bias': om.Bias()}
  for ri, (name, region) in enumerate(oregions.items()):
    for mname, metric in suite.items():
      want = metric.compute_chunk(fd, td, region=region)['z'].data
      np.testing.assert_allclose(got[_lib.METRIC_INDEX[mname], ri], want,
                                 rtol=1e-9, atol=1e-12,
                                 err_msg=f'{mname}/{name}')
    want = om.ACC(clim).compute_chunk(ft, tt, region=region)['z'].data[0]
    np.testing.assert_allclose(got[_lib.METRIC_INDEX['acc'], ri], want,
                               rtol=1e-9, atol=1e-12, err_msg=f'acc/{name}')


def test_bad_accumulate_view_is_an_error(dev):
  import torch
  from weatherbench2_amd import _lib, engine, plan as plan_lib
  pl = plan_lib.build_plan(LAT, LON, plan_lib.LATLON, None, dev)
  step = engine.SuiteStep(pl, _lib.MODE_DET, torch.float32, False, N_LEV)
  acc = torch.zeros((7,), dtype=torch.float64, device=dev)
  with pytest.raises(ValueError):
    step.accumulate_into(acc, acc.clone(), (7, 1, 1))
