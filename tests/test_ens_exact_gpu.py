"""GPU parity of K3's compile-time member counts (-m gpu): the kernels of
csrc/ensemble_exact.hip (one object per member count of WB2_SORT3_SIZES,
csrc/sort3_networks.inc: 4 ... 100 float32 members) and the
NaN-free fast path of the exact skipna kernels, against the NumPy oracle
(oracle/metrics_np.py restating /root/reference/weatherbench2/metrics.py:
532-565, 585-607, 775-846, 1161-1363)."""
import numpy as np
import pytest

from oracle import metrics_np as om
from oracle import regions_np as oreg
from oracle.named import DS, NA
from tests import helpers

pytestmark = pytest.mark.gpu

from weatherbench2_amd import build as _build

# every member count with a kernel of its own (WB2_SORT3_SIZES)
EXACT = tuple(m for m, _ in _build.exact_sizes())
# counts WITHOUT one: run inside the next larger program with the dead slots
# at +inf (ens_point_hosted; without skipna) -- below, between and next to the
# instantiated sizes, incl. the smallest (2, 3) and the last before 100
HOSTED = (2, 3, 6, 7, 9, 13, 24, 33, 44, 47, 49, 63, 77, 99)
N_LAT, N_LON = 721, 1440
LAT = np.linspace(-90, 90, N_LAT)
LON = np.linspace(0, 360, N_LON, endpoint=False)


@pytest.fixture(scope='module')
def dev():
  import torch
  if not torch.cuda.is_available():
    pytest.fail('-m gpu tests need a HIP device')
  return torch.device('cuda', 0)


def _oracle_fields(f, t, skipna):
  """The six pointwise fields of K3 (slot order) as DS."""
  mean = f.mean('realization', skipna=skipna)
  return [
      om.pointwise_crps_skill(f, t, 'realization', skipna),
      om.pointwise_crps_spread(f, 'realization', skipna),
      (t - mean) ** 2,
      f.var('realization', skipna=skipna, ddof=1),
      f.std('realization', skipna=skipna, ddof=1) ** 2,
      om.debiased_ensemble_mean_mse(f, t, 'realization', skipna),
  ]


@pytest.mark.parametrize('skipna', [False, True])
@pytest.mark.parametrize('m', EXACT + HOSTED)
def test_exact_member_counts_match_oracle(dev, m, skipna):
  """Small grid (3 column tiles per row), ties, an infinite member, a NaN
  patch: with skipna some waves take the fast path and some the general one;
  slice regions + a land-sea mask (the WF instantiation)."""
  import torch
  from weatherbench2_amd import _lib, engine, plan as plan_lib
  n_lat, n_lon, n_slab = 23, 180, 3
  lat = np.linspace(-90, 90, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  rs = np.random.RandomState(100 + m)
  ens = rs.normal(size=(m, n_slab, n_lat, n_lon)).astype(np.float32)
  ens[:, 0, :, 100:140] = np.round(ens[:, 0, :, 100:140] * 2) / 2   # ties
  truth = rs.normal(size=(n_slab, n_lat, n_lon)).astype(np.float32)
  ens[min(3, m - 1), 1, 5, 70] = np.inf
  # NaN patches: columns 0..19 of some rows (the first wave of those rows)
  ens[rs.randint(0, m, size=40), 2, rs.randint(0, n_lat, size=40),
      rs.randint(0, 20, size=40)] = np.nan
  truth[2, 7, 3] = np.nan
  lsm = np.clip(rs.uniform(-0.5, 1.3, size=(n_lat, n_lon)), 0, 1)
  oregions = {
      'global': oreg.SliceRegion(),
      'tropics': oreg.SliceRegion(lat_slice=slice(-20, 20)),
      'box': oreg.SliceRegion(lat_slice=slice(-30, 60),
                              lon_slice=slice(30, 200)),
      'land': oreg.LandRegion(NA(lsm, ('latitude', 'longitude')), lat, lon),
  }
  gregions = {k: helpers.to_gpu_region(v) for k, v in oregions.items()}
  pl = plan_lib.build_plan(lat, lon, plan_lib.LATLON, gregions, dev,
                           rows_per_chunk=plan_lib.ENSEMBLE_ROWS_PER_CHUNK)
  metrics, _ = engine.ensemble_reduce(
      pl, torch.as_tensor(ens, device=dev), n_slab * n_lat * n_lon, m, None,
      torch.as_tensor(truth, device=dev), None, n_slab, skipna)
  got = metrics.cpu().numpy()
  idx = _lib.ENS_METRIC_INDEX
  dims = ('realization', 'latitude', 'longitude')
  coords = {'latitude': lat, 'longitude': lon}
  names = ('crps_skill', 'crps_spread', 'ensemble_mean_mse',
           'ensemble_variance', None, 'debiased_ensemble_mean_mse')
  for s in range(n_slab):
    if not skipna and s == 2:
      continue  # NaNs without skipna: every number NaN, checked below
    f = DS({'z': NA(ens[:, s], dims)}, coords)
    t = DS({'z': NA(truth[s], dims[1:])}, coords)
    with np.errstate(all='ignore'):
      fields = _oracle_fields(f, t, skipna)
      for ri, rname in enumerate(pl.region_names):
        sa = [float(np.asarray(om.spatial_average(
            x, oregions[rname], skipna)['z'].data)) for x in fields]
        want = {n: v for n, v in zip(names, sa) if n}
        want['crps'] = want['crps_skill'] - 0.5 * want['crps_spread']
        want['ensemble_mean_rmse'] = np.sqrt(want['ensemble_mean_mse'])
        want['ensemble_stddev'] = np.sqrt(sa[4])
        for name, w in want.items():
          helpers.assert_close(got[idx[name], ri, s], w, rtol=2e-6, atol=1e-7,
                               err_msg=f'M={m} slab {s} {name}/{rname}')
  if not skipna:
    assert np.isnan(got[idx['crps'], 0, 2])


@pytest.mark.parametrize('m', [10, 30, 50, 51, 3, 13, 44, 77])  # + hosted counts
def test_skipna_fast_path_gives_the_general_paths_bits(dev, m):
  """Pointwise maps: the same ensemble twice, once clean (every wave takes the
  NaN-free fast path) and once with one NaN per 64-column tile and row (every
  wave takes the general path): all other points must agree bit for bit."""
  import torch
  from weatherbench2_amd import engine, plan as plan_lib
  n_lat, n_lon = 11, 256
  lat = np.linspace(-90, 90, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  rs = np.random.RandomState(m)
  ens = rs.normal(size=(m, 1, n_lat, n_lon)).astype(np.float32)
  ens[:, 0, :, 10:30] = np.round(ens[:, 0, :, 10:30] * 2) / 2
  ens[1, 0, 4, 200] = np.inf
  truth = rs.normal(size=(1, n_lat, n_lon)).astype(np.float32)
  dirty = ens.copy()
  dirty[0, 0, :, 5::64] = np.nan
  pl = plan_lib.build_plan(lat, lon, plan_lib.LATLON, None, dev,
                           rows_per_chunk=plan_lib.ENSEMBLE_ROWS_PER_CHUNK)
  out = []
  for x in (ens, dirty):
    maps = torch.empty((6, 1, n_lat * n_lon), dtype=torch.float64, device=dev)
    engine.ensemble_reduce(pl, torch.as_tensor(x, device=dev), n_lat * n_lon,
                           m, None, torch.as_tensor(truth, device=dev), None,
                           1, True, maps=maps)
    out.append(maps.cpu().numpy().reshape(6, n_lat, n_lon))
  keep = np.ones(n_lon, bool)
  keep[5::64] = False
  a, b = out[0][:, :, keep], out[1][:, :, keep]
  assert np.array_equal(a, b, equal_nan=True)
  assert np.isfinite(a[:, :, :100]).all()


@pytest.mark.parametrize('m', EXACT + (33, 44, 47, 63, 77))
def test_exact_member_counts_full_size(dev, m):
  """One 721 x 1440 slab per member count at the benched geometry (5-row
  chunks, the 13 predefined regions): all eight metrics vs the oracle."""
  import torch
  from weatherbench2_amd import _lib, engine, plan as plan_lib
  rs = np.random.RandomState(m)
  ens = rs.normal(size=(m, 1, N_LAT, N_LON)).astype(np.float32)
  truth = rs.normal(size=(1, N_LAT, N_LON)).astype(np.float32)
  regions = helpers.predefined_regions(oracle=False)
  oregions = helpers.predefined_regions(oracle=True)
  pl = plan_lib.build_plan(LAT, LON, plan_lib.LATLON, regions, dev,
                           rows_per_chunk=plan_lib.ENSEMBLE_ROWS_PER_CHUNK)
  metrics, _ = engine.ensemble_reduce(
      pl, torch.as_tensor(ens, device=dev), N_LAT * N_LON, m, None,
      torch.as_tensor(truth, device=dev), None, 1, False)
  got = metrics.cpu().numpy()
  idx = _lib.ENS_METRIC_INDEX
  dims = ('realization', 'latitude', 'longitude')
  coords = {'latitude': LAT, 'longitude': LON}
  f = DS({'z': NA(ens[:, 0], dims)}, coords)
  t = DS({'z': NA(truth[0], dims[1:])}, coords)
  fields = _oracle_fields(f, t, False)
  names = ('crps_skill', 'crps_spread', 'ensemble_mean_mse',
           'ensemble_variance', None, 'debiased_ensemble_mean_mse')
  for ri, rname in enumerate(pl.region_names):
    sa = [float(np.asarray(om.spatial_average(
        x, oregions[rname], False)['z'].data)) for x in fields]
    want = {n: v for n, v in zip(names, sa) if n}
    want['crps'] = want['crps_skill'] - 0.5 * want['crps_spread']
    want['ensemble_mean_rmse'] = np.sqrt(want['ensemble_mean_mse'])
    want['ensemble_stddev'] = np.sqrt(sa[4])
    for name, w in want.items():
      helpers.assert_close(got[idx[name], ri, 0], w, rtol=2e-6, atol=1e-7,
                           err_msg=f'M={m} {name}/{rname}')


@pytest.mark.parametrize('m', [3, 7, 21, 44, 77, 100])
def test_hosted_counts_give_the_padded_networks_values(dev, m, monkeypatch,
                                                       tmp_path):
  """The hosted program and the padded power-of-two network (WB2HIP_ENS_HOSTED
  is read once per process, so the second form comes from a GATHERED launch
  with the variable unset vs. a child process) do the same operations on the
  live members in the same order: pointwise maps agree bit for bit, strided and
  gathered members alike -- including an infinite member, ties and a NaN
  member (every value NaN there)."""
  import subprocess
  import sys
  import torch
  from weatherbench2_amd import engine, plan as plan_lib
  n_lat, n_lon = 9, 200
  lat = np.linspace(-90, 90, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  rs = np.random.RandomState(7 * m)
  ens = rs.normal(size=(m, 2, n_lat, n_lon)).astype(np.float32)
  ens[:, 0, :, 10:30] = np.round(ens[:, 0, :, 10:30] * 2) / 2
  ens[min(1, m - 1), 0, 4, 150] = np.inf
  ens[0, 1, 3, 77] = np.nan
  ens[m - 1, 1, 6, 12] = np.nan
  truth = rs.normal(size=(2, n_lat, n_lon)).astype(np.float32)
  truth[1, 2, 2] = np.nan
  pl = plan_lib.build_plan(lat, lon, plan_lib.LATLON, None, dev,
                           rows_per_chunk=plan_lib.ENSEMBLE_ROWS_PER_CHUNK)
  d_ens = torch.as_tensor(ens, device=dev)
  d_truth = torch.as_tensor(truth, device=dev)
  slab = n_lat * n_lon

  def run(gather):
    maps = torch.empty((6, 2, slab), dtype=torch.float64, device=dev)
    ptrs = None
    if gather:
      index = np.arange(m)[None, :] * 2 + np.arange(2)[:, None]  # [outer, m]
      ptrs = torch.as_tensor(engine.gather_pointers(d_ens, index, slab),
                             device=dev)
    engine.ensemble_reduce(pl, d_ens, 2 * slab, m, None, d_truth, None, 2,
                           False, maps=maps, member_ptrs=ptrs)
    return maps.cpu().numpy()
  strided, gathered = run(False), run(True)
  if m in EXACT:
    # a strided exact size has a kernel of its own (compile-time divisions,
    # 2 / (M (M - 1)) as one constant): equal to float32 rounding
    helpers.assert_close(strided, gathered, rtol=2e-6, atol=1e-7)
  else:
    assert np.array_equal(strided, gathered, equal_nan=True)
  for maps in (strided, gathered):
    assert np.isnan(maps[:, 1, 3 * n_lon + 77]).all()
    assert np.isnan(maps[:, 1, 6 * n_lon + 12]).all()
    assert np.isfinite(maps[:, 0, :100]).all()
  # the padded runtime network in a process of its own
  f_ens, f_truth, f_out = (str(tmp_path / f'{k}.npy')
                           for k in ('ens', 'truth', 'padded'))
  np.save(f_ens, ens)
  np.save(f_truth, truth)
  code = f'''
import numpy as np, torch
from weatherbench2_amd import engine, plan as plan_lib
ens = np.load({f_ens!r}); truth = np.load({f_truth!r})
dev = torch.device('cuda', 0)
lat = np.linspace(-90, 90, {n_lat}); lon = np.linspace(0, 360, {n_lon}, endpoint=False)
pl = plan_lib.build_plan(lat, lon, plan_lib.LATLON, None, dev, rows_per_chunk=plan_lib.ENSEMBLE_ROWS_PER_CHUNK)
maps = torch.empty((6, 2, {slab}), dtype=torch.float64, device=dev)
engine.ensemble_reduce(pl, torch.as_tensor(ens, device=dev), 2 * {slab}, {m}, None, torch.as_tensor(truth, device=dev), None, 2, False, maps=maps)
np.save({f_out!r}, maps.cpu().numpy())
'''
  import os
  env = dict(os.environ, WB2HIP_ENS_HOSTED='0')
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  res = subprocess.run([sys.executable, '-c', code], env=env, cwd=root,
                       capture_output=True, text=True, timeout=600)
  assert res.returncode == 0, res.stderr[-2000:]
  padded = np.load(f_out)
  if m in EXACT:  # (the child's strided launch took the exact kernel too)
    assert np.array_equal(strided, padded, equal_nan=True)
  else:
    assert np.array_equal(gathered, padded, equal_nan=True)
