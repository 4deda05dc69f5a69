"""GPU parity of the tier-2 metrics built so far (-m gpu): Gaussian CRPS /
variance (one extra kernel mode) and the energy score (host composition of the
fused deterministic pass), incl. xarray's inner join on mismatched times."""
import numpy as np
import pytest

from oracle import fixtures
from oracle import metrics_np as om
from oracle import regions_np as oreg
from tests import helpers

pytestmark = pytest.mark.gpu


def _gaussian_fixture():
  kw = dict(variables_3d=[], time_start='2022-01-01')
  forecast = fixtures.mock_forecast_data(
      variables_2d=['2m_temperature', '2m_temperature_std'],
      time_stop='2022-01-02', lead_stop='1 day', **kw)
  truth = fixtures.mock_truth_data(variables_2d=['2m_temperature'],
                                   time_stop='2022-01-20', **kw)
  return forecast, truth


def test_gaussian_known_answers():
  # metrics_test.py:286-304, 340-362 (forecast has 1 time, truth 19: inner join)
  from weatherbench2_amd import metrics as gm
  forecast, truth = _gaussian_fixture()
  forecast = forecast + 1.0
  truth = truth + 1.02
  g = helpers.to_gpu_dataset
  result = gm.GaussianCRPS().compute(g(forecast), g(truth))
  np.testing.assert_allclose(result['2m_temperature'].values,
                             np.array([0.23385455, 0.23385455]), rtol=1e-6)
  result = gm.GaussianVariance().compute(g(forecast), g(truth))
  np.testing.assert_allclose(result['2m_temperature'].values, [1.0, 1.0])


@pytest.mark.parametrize('skipna', [False, True])
def test_gaussian_random_matches_oracle(skipna):
  from weatherbench2_amd import metrics as gm
  truth, mean = fixtures.get_random_truth_and_forecast(
      variables=('geopotential',), spatial_resolution_in_degrees=10)
  _, std = fixtures.get_random_truth_and_forecast(
      variables=('geopotential',), spatial_resolution_in_degrees=10, seed=77)
  data = {'geopotential': mean['geopotential'].data.astype(np.float32),
          'geopotential_std': (np.abs(std['geopotential'].data) + 0.1
                               ).astype(np.float32)}
  from oracle.named import DS, NA
  dims = mean['geopotential'].dims
  forecast = DS({k: NA(v, dims) for k, v in data.items()}, mean.coords)
  truth = truth.copy(data={'geopotential':
                           truth['geopotential'].data.astype(np.float32)})
  if skipna:
    forecast = fixtures.insert_nan(forecast, 0.05, seed=1)
    forecast = forecast.copy(data={k: v.data.astype(np.float32)
                                   for k, v in forecast.items()})
  g = helpers.to_gpu_dataset
  regions = {'global': None,
             'tropics': oreg.SliceRegion(lat_slice=slice(-20, 20))}
  g_regions = {k: helpers.to_gpu_region(v) for k, v in regions.items()}
  with gm.fused_regions(g_regions):
    for oc, gc in ((om.GaussianCRPS(), gm.GaussianCRPS()),
                   (om.GaussianVariance(), gm.GaussianVariance())):
      for rname, region in regions.items():
        want = oc.compute_chunk(forecast, truth, region=region, skipna=skipna)
        got = gc.compute_chunk(g(forecast), g(truth), region=g_regions[rname],
                               skipna=skipna)
        helpers.assert_close(got['geopotential'].values,
                             want['geopotential'].data, rtol=1e-9, atol=1e-12,
                             err_msg=f'{type(oc).__name__}/{rname}')


@pytest.mark.parametrize('ensemble_size', [1, 2, 3, 10])
def test_energy_score_matches_oracle(ensemble_size):
  from weatherbench2_amd import metrics as gm
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=ensemble_size, lead_stop='3 day')
  g = helpers.to_gpu_dataset
  gf, gt = g(forecast), g(truth)
  region = oreg.ExtraTropicalRegion()
  for oname in ('EnergyScore', 'EnergyScoreSpread', 'EnergyScoreSkill'):
    for reg, greg in ((None, None), (region, helpers.to_gpu_region(region))):
      want = getattr(om, oname)().compute_chunk(forecast, truth, region=reg)
      got = getattr(gm, oname)().compute_chunk(gf, gt, region=greg)
      assert got['geopotential'].dims == want['geopotential'].dims
      helpers.assert_close(got['geopotential'].values,
                           want['geopotential'].data, rtol=1e-9, atol=1e-12,
                           err_msg=oname)


def test_gaussian_scores_pointwise_against_scipy():
  """The Gaussian kernels' normal cdf / pdf (csrc/gauss_math.hpp: one exp, an
  erfcx table in LDS, the asymptotic series beyond |z| = 17) point by point:
  every outer slab is a 2 x 2 grid holding ONE sample four times, so the
  spatial mean is the sample's score.  z sweeps [-37, 37] (both tails, the
  table's interval edges, the switch to the series; beyond, the tail is a
  float64 denormal and nobody's digits mean anything) and |z| > 39 (the tail
  underflows to 0 on both sides), plus NaN / inf / zero std; against
  scipy.stats.norm in float64."""
  import torch
  from scipy import stats
  from weatherbench2_amd import _lib, engine, plan as plan_lib
  dev = torch.device('cuda', 0)
  rng = np.random.default_rng(5)
  edges = np.arange(0, 26 * 16 + 1) / 16.0 * np.sqrt(2.0)
  z = np.concatenate([rng.uniform(-37, 37, 20000), rng.normal(0, 1.5, 20000),
                      edges, -edges, np.nextafter(edges, 0), [0.0, -0.0],
                      rng.uniform(39, 60, 50), -rng.uniform(39, 60, 50)])
  sd = rng.uniform(0.25, 4.0, z.size).astype(np.float32)
  y = rng.normal(0, 3, z.size).astype(np.float32)
  mean = (y + z.astype(np.float32) * sd).astype(np.float32)
  special = np.array([[np.nan, 1, 0], [0, np.nan, 0], [0, 1, np.nan],
                      [1, 0, 0], [0, 0, 0], [np.inf, 1, 0], [0, np.inf, 1]],
                     dtype=np.float32)
  mean = np.concatenate([mean, special[:, 0]])
  sd = np.concatenate([sd, special[:, 1]])
  y = np.concatenate([y, special[:, 2]])
  n = mean.size
  plan = plan_lib.build_plan(np.array([-30.0, 30.0]), np.array([0.0, 180.0]),
                             plan_lib.LATLON, {'global': None}, dev)

  def slabs(v):
    return torch.from_numpy(np.repeat(v[:, None], 4, 1).reshape(n, 2, 2)).to(dev)
  with np.errstate(all='ignore'):
    nd = ((mean - y) / sd).astype(np.float64)  # formed in float32, promoted
    want_crps = sd.astype(np.float64) * (
        nd * (2 * stats.norm.cdf(nd) - 1) + 2 * stats.norm.pdf(nd)
        - 1 / np.sqrt(np.pi))
  got, _ = engine.stream_reduce(plan, _lib.MODE_GAUSS,
                                [slabs(mean), slabs(sd), slabs(y)],
                                [None] * 3, n, False)
  got = got.cpu().numpy()[:, 0, :]
  np.testing.assert_allclose(got[0], want_crps, rtol=1e-13, atol=1e-14,
                             equal_nan=True)
  np.testing.assert_allclose(got[1], (sd * sd).astype(np.float64), rtol=1e-15,
                             equal_nan=True)
  # thresholds: Brier, ignorance, RPS part (metrics.py:975-1121)
  thr = rng.normal(0, 3, n).astype(np.float32)
  with np.errstate(all='ignore'):
    nt = ((thr - mean) / sd).astype(np.float64)
    cdf = stats.norm.cdf(nt)
    above, below = y > thr, y < thr
    want = [((1 - cdf) - above) ** 2,
            -np.where(above, np.log(1 - cdf), np.log(cdf)),
            (cdf - below) ** 2]
  got, _ = engine.stream_reduce(plan, _lib.MODE_GAUSS_THR,
                                [slabs(mean), slabs(sd), slabs(y), slabs(thr)],
                                [None] * 4, n, False)
  got = got.cpu().numpy()[:, 0, :]
  for k in range(3):
    np.testing.assert_allclose(got[k], want[k], rtol=2e-13, atol=1e-15,
                               equal_nan=True, err_msg=str(k))


@pytest.mark.parametrize('skipna', [False, True])
@pytest.mark.parametrize('m', [1, 2, 8, 9, 17, 50])
def test_fused_energy_score_blocks_regions_and_nans(m, skipna):
  """wb2_energy_score across its member blocks (8 members per wave, 4 with a
  mask AND skipna: 9 = one full block + one member whose block has no pair of
  its own; 17 and 50 straddle several blocks), slice regions + a land-sea mask
  in one pass, NaN patches in members and truth, against the oracle's
  EnergyScore{,Spread,Skill} (oracle/metrics_np.py restating
  /root/reference/weatherbench2/metrics.py:1403-1517) region by region."""
  from oracle.named import DS, NA
  from weatherbench2_amd import metrics as gm
  n_lat, n_lon, n_time = 19, 150, 2
  lat = np.linspace(-90, 90, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  rs = np.random.RandomState(31 + m)
  ens = rs.normal(size=(m, n_time, n_lat, n_lon)).astype(np.float32)
  tru = rs.normal(size=(n_time, n_lat, n_lon)).astype(np.float32)
  if skipna:
    ens[rs.randint(0, m, 30), rs.randint(0, n_time, 30),
        rs.randint(0, n_lat, 30), rs.randint(0, n_lon, 30)] = np.nan
    tru[1, 3, 7] = np.nan
    if m > 2:
      ens[2, 0] = np.nan       # a member that is NaN everywhere (time 0)
  lsm = np.clip(rs.uniform(-0.5, 1.3, size=(n_lat, n_lon)), 0, 1)
  coords = {'time': np.arange(n_time), 'latitude': lat, 'longitude': lon}
  forecast = DS({'z': NA(ens, ('realization', 'time', 'latitude', 'longitude'))},
                dict(coords, realization=np.arange(m)))
  truth = DS({'z': NA(tru, ('time', 'latitude', 'longitude'))}, coords)
  oregions = {
      'global': oreg.SliceRegion(),
      'tropics': oreg.SliceRegion(lat_slice=slice(-20, 20)),
      'box': oreg.SliceRegion(lat_slice=slice(-30, 60),
                              lon_slice=slice(30, 200)),
      'land': oreg.LandRegion(NA(lsm, ('latitude', 'longitude')), lat, lon),
  }
  gregions = {k: helpers.to_gpu_region(v) for k, v in oregions.items()}
  gf, gt = helpers.to_gpu_dataset(forecast), helpers.to_gpu_dataset(truth)
  for oname in ('EnergyScore', 'EnergyScoreSpread', 'EnergyScoreSkill'):
    got = getattr(gm, oname)().compute_chunk_regions(gf, gt, gregions, skipna)
    assert list(got.coords['region']) == list(gregions)
    for ri, (rname, region) in enumerate(oregions.items()):
      with np.errstate(all='ignore'):
        want = getattr(om, oname)().compute_chunk(forecast, truth,
                                                  region=region, skipna=skipna)
      helpers.assert_close(got['z'].values[ri], want['z'].data, rtol=1e-9,
                           atol=1e-12, err_msg=f'{oname}/{rname}/M={m}')


def test_fused_energy_score_full_size_and_float64():
  """One 721 x 1440 slab set at the benched geometry (50 members, 5-row
  chunks, 13 regions) through the engine call, float32 and float64 members,
  gathered through slab tables: every region against the oracle."""
  import torch
  from oracle.named import DS, NA
  from weatherbench2_amd import engine, plan as plan_lib
  dev = torch.device('cuda', 0)
  n_lat, n_lon, m = 721, 1440, 50
  lat = np.linspace(-90, 90, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  regions = helpers.predefined_regions(oracle=False)
  oregions = helpers.predefined_regions(oracle=True)
  pl = plan_lib.build_plan(lat, lon, plan_lib.LATLON, regions, dev,
                           rows_per_chunk=plan_lib.ENSEMBLE_ROWS_PER_CHUNK)
  for dtype, tdtype in ((np.float32, torch.float32), (np.float64,
                                                      torch.float64)):
    rs = np.random.RandomState(3)
    ens = rs.normal(size=(m, 2, n_lat, n_lon)).astype(dtype)
    tru = rs.normal(size=(2, n_lat, n_lon)).astype(dtype)
    tab = torch.as_tensor([1], device=dev)        # outer 0 = slab 1
    out = engine.energy_score(pl, torch.as_tensor(ens, device=dev),
                              2 * n_lat * n_lon, m, tab,
                              torch.as_tensor(tru, device=dev), tab, 1,
                              False).cpu().numpy()
    coords = {'latitude': lat, 'longitude': lon}
    f = DS({'z': NA(ens[:, 1], ('realization', 'latitude', 'longitude'))},
           dict(coords, realization=np.arange(m)))
    t = DS({'z': NA(tru[1], ('latitude', 'longitude'))}, coords)
    for ri, rname in enumerate(pl.region_names):
      for row, oname in enumerate(('EnergyScore', 'EnergyScoreSpread',
                                   'EnergyScoreSkill')):
        want = getattr(om, oname)().compute_chunk(f, t,
                                                  region=oregions[rname])
        helpers.assert_close(out[row, ri, 0], want['z'].data, rtol=1e-9,
                             atol=1e-12, err_msg=f'{oname}/{rname}/{dtype}')


@pytest.mark.parametrize('with_land', [False, True])
def test_fused_energy_score_skipna_with_the_official_region_sets(with_land):
  """skipna doubles the sums per member block (32 slots) and the 13 predefined
  regions cut the grid into 14 bands x 13 segs: the combine kernel then needs
  more than the default 64 KiB of LDS (gfx950 lets a workgroup take 160 KiB);
  with a land-sea mask on top the blocks shrink to 4 members."""
  from oracle.named import DS, NA
  from weatherbench2_amd import metrics as gm
  n_lat, n_lon, m = 37, 144, 10
  lat = np.linspace(-90, 90, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  rs = np.random.RandomState(77)
  ens = rs.normal(size=(m, n_lat, n_lon)).astype(np.float32)
  tru = rs.normal(size=(n_lat, n_lon)).astype(np.float32)
  ens[rs.randint(0, m, 40), rs.randint(0, n_lat, 40),
      rs.randint(0, n_lon, 40)] = np.nan
  coords = {'latitude': lat, 'longitude': lon}
  forecast = DS({'z': NA(ens, ('realization', 'latitude', 'longitude'))},
                dict(coords, realization=np.arange(m)))
  truth = DS({'z': NA(tru, ('latitude', 'longitude'))}, coords)
  oregions = helpers.predefined_regions(oracle=True)
  if with_land:
    lsm = np.clip(rs.uniform(-0.5, 1.3, size=(n_lat, n_lon)), 0, 1)
    oregions['land'] = oreg.LandRegion(NA(lsm, ('latitude', 'longitude')),
                                       lat, lon)
  gregions = {k: helpers.to_gpu_region(v) for k, v in oregions.items()}
  got = gm.EnergyScore().compute_chunk_regions(
      helpers.to_gpu_dataset(forecast), helpers.to_gpu_dataset(truth),
      gregions, True)
  for ri, (rname, region) in enumerate(oregions.items()):
    with np.errstate(all='ignore'):
      want = om.EnergyScore().compute_chunk(forecast, truth, region=region,
                                            skipna=True)
    helpers.assert_close(got['z'].values[ri], want['z'].data, rtol=1e-9,
                         atol=1e-12, err_msg=rname)
