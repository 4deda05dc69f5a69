"""GPU parity of RankHistogram (-m gpu) and host parity of central_reliability.

Tie-free inputs: the counting kernel must equal the reference's
argsort-of-perturbed-values rank exactly (the oracle restates that with NumPy's
RNG).  Tied inputs: the reference's own statistical tests
(metrics_test.py:538-650), since tie breaking is random by definition.
"""
import numpy as np
import pytest

from oracle import fixtures
from oracle import metrics_np as om
from oracle.named import DS, NA
from tests import helpers


def _values(da):
  v = da.data
  return v.cpu().numpy() if hasattr(v, 'cpu') else np.asarray(v)


def _as(ds, dtype):
  return ds.copy(data={k: v.data.astype(dtype) for k, v in ds.items()})


@pytest.mark.gpu
@pytest.mark.parametrize('ensemble_size,num_bins,dtype', [
    (1, None, np.float32), (2, None, np.float64), (10, None, np.float32),
    (9, 5, np.float32), (50, None, np.float32), (50, 17, np.float64),
    (127, 64, np.float32)])
def test_one_hot_matches_oracle_without_ties(ensemble_size, num_bins, dtype):
  from weatherbench2_amd import metrics as gm
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=ensemble_size, spatial_resolution_in_degrees=15,
      time_start='2019-12-01', time_stop='2019-12-04', levels=(0, 1, 2))
  truth, forecast = _as(truth, dtype), _as(forecast, dtype)
  want = om.RankHistogram(num_bins=num_bins, seed=3).compute_chunk(
      forecast, truth)
  g = helpers.to_gpu_dataset
  got = gm.RankHistogram(num_bins=num_bins, seed=3).compute_chunk(
      g(forecast), g(truth))
  for name in want.keys():
    w = want[name].transpose(*got[name].dims)
    a = _values(got[name])
    assert a.dtype == np.float64
    np.testing.assert_array_equal(a, w.data)
    np.testing.assert_array_equal(a.sum(-1), 1.0)
  np.testing.assert_array_equal(got.coords['bins'],
                                np.arange(num_bins or ensemble_size + 1))


@pytest.mark.gpu
def test_nan_members_and_truth_rank_highest():
  from weatherbench2_amd import metrics as gm
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=6, spatial_resolution_in_degrees=20,
      time_start='2019-12-01', time_stop='2019-12-03', levels=(0,))
  forecast = fixtures.insert_nan(forecast, 0.2, seed=4)
  g = helpers.to_gpu_dataset
  # without perturbation NumPy's sort puts NaNs last and the rest is tie-free
  want = om.RankHistogram(break_ties_randomly=False).compute_chunk(
      forecast, truth)
  got = gm.RankHistogram(break_ties_randomly=False).compute_chunk(
      g(forecast), g(truth), skipna=True)
  w = want['geopotential'].transpose(*got['geopotential'].dims)
  np.testing.assert_array_equal(_values(got['geopotential']), w.data)
  # NaN truth: above every non-NaN member
  tdata = truth['geopotential'].data.copy()
  tdata[...] = np.nan
  nan_truth = truth.copy(data={'geopotential': tdata})
  got = gm.RankHistogram().compute_chunk(g(forecast), g(nan_truth))
  fd = forecast['geopotential']
  ax = fd.dims.index('realization')
  n_ok = (~np.isnan(fd.data)).sum(ax)
  rest = tuple(d for d in fd.dims if d != 'realization')
  ranks = _values(got['geopotential']).argmax(-1)
  order = [rest.index(d) for d in got['geopotential'].dims[:-1]]
  np.testing.assert_array_equal(ranks, np.transpose(n_ok, order))


@pytest.mark.gpu
def test_compute_is_temporal_mean_of_chunks():
  from weatherbench2_amd import metrics as gm
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=5, spatial_resolution_in_degrees=15,
      time_start='2019-12-01', time_stop='2019-12-08', levels=(0, 1))
  g = helpers.to_gpu_dataset
  metric = gm.RankHistogram(num_bins=3, seed=0)
  chunk = metric.compute_chunk(g(forecast), g(truth))['geopotential']
  mean = metric.compute(g(forecast), g(truth))
  assert mean.attrs['ensemble_size'] == 5
  da = mean['geopotential']
  assert 'time' not in da.dims and da.dims[-1] == 'bins'
  want = _values(chunk).mean(chunk.dims.index('time'))
  np.testing.assert_allclose(_values(da), want, rtol=0, atol=1e-15)


@pytest.mark.gpu
def test_ties_are_broken_uniformly_and_reproducibly():
  # metrics_test.py:603-650: forecast == truth == 0 everywhere
  from weatherbench2_amd import metrics as gm
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=4, spatial_resolution_in_degrees=10,
      time_start='2019-12-01', time_stop='2019-12-21', levels=(0, 1))
  truth, forecast = truth * 0.0, forecast * 0.0
  g = helpers.to_gpu_dataset
  a = gm.RankHistogram(seed=11).compute_chunk(g(forecast), g(truth))
  b = gm.RankHistogram(seed=11).compute_chunk(g(forecast), g(truth))
  c = gm.RankHistogram(seed=12).compute_chunk(g(forecast), g(truth))
  va, vb, vc = (_values(x['geopotential']) for x in (a, b, c))
  np.testing.assert_array_equal(va, vb)
  assert (va != vc).any()
  flat = va.reshape(-1, 5)
  hist = flat.mean(0)
  rtol = 5 * np.sqrt(4 / flat.shape[0])
  np.testing.assert_allclose(hist, 0.2, rtol=rtol)
  # neighbouring samples must not share their draw
  ranks = flat.argmax(-1)
  assert abs(np.corrcoef(ranks[:-1], ranks[1:])[0, 1]) < 0.02
  # partially tied: two of four members equal the truth -> rank in {lo..lo+2}
  first = gm.RankHistogram(break_ties_randomly=False).compute_chunk(
      g(forecast), g(truth))
  np.testing.assert_array_equal(
      _values(first['geopotential']).reshape(-1, 5).argmax(-1), 0)


@pytest.mark.gpu
@pytest.mark.parametrize('ensemble_size,num_bins', [(1, None), (10, None),
                                                    (9, 5)])
def test_well_and_mis_calibrated_levels(ensemble_size, num_bins):
  # metrics_test.py:540-600 through compute() (device-side temporal mean)
  from weatherbench2_amd import metrics as gm
  num_bins = ensemble_size + 1 if num_bins is None else num_bins
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=ensemble_size, time_start='2019-12-01',
      time_stop='2019-12-10', levels=(0, 1, 2, 3, 4))
  data = forecast['geopotential'].data.copy()
  lev = forecast['geopotential'].dims.index('level')
  sl = lambda i: tuple(i if a == lev else slice(None)
                       for a in range(data.ndim))
  data[sl(1)] *= 0.1
  data[sl(2)] *= 10
  data[sl(3)] -= 1
  data[sl(4)] += 1
  forecast = forecast.copy(data={'geopotential': data})
  g = helpers.to_gpu_dataset
  mean = gm.RankHistogram(num_bins=num_bins).compute(g(forecast), g(truth))
  da = mean['geopotential']
  v = _values(da)
  keep = (da.dims.index('level'), da.dims.index('bins'))
  others = tuple(i for i in range(v.ndim) if i not in keep)
  hist = v.mean(others)
  if keep[0] > keep[1]:
    hist = hist.T
  n_time = forecast.sizes['time']
  sample_size = v.size / v.shape[keep[0]] / num_bins * n_time
  rtol = 5 * np.sqrt((num_bins - 1) / sample_size)
  np.testing.assert_allclose(hist[0], 1 / num_bins, rtol=rtol)
  if num_bins > 2:
    convex, concave = hist[1], hist[2]
    assert (np.diff(convex[:len(convex) // 2 + 1]) < 0).all()
    assert (np.diff(convex[len(convex) // 2:]) > 0).all()
    assert (np.diff(concave[:len(concave) // 2 + 1]) > 0).all()
    assert (np.diff(concave[len(concave) // 2:]) < 0).all()
  assert (np.diff(hist[3]) > 0).all()
  assert (np.diff(hist[4]) < 0).all()
  # perfectly calibrated level: the reliability curve is the diagonal
  if num_bins >= 3:
    from weatherbench2_amd import xarray_lite as xl
    rel = gm.central_reliability(xl.DataArray(
        hist[0], ('bins',), {'bins': np.arange(num_bins)}, 'z'))
    np.testing.assert_allclose(rel.values, rel.coords['desired_prob'],
                               atol=3 * rtol)


@pytest.mark.gpu
def test_bad_bin_count_raises():
  from weatherbench2_amd import metrics as gm
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=4, spatial_resolution_in_degrees=30)
  g = helpers.to_gpu_dataset
  with pytest.raises(ValueError, match='Cannot bin data'):
    gm.RankHistogram(num_bins=3).compute_chunk(g(forecast), g(truth))


# ---- host logic, no GPU needed ---------------------------------------------
@pytest.mark.parametrize('n_bins', [3, 4, 10, 11])
def test_central_reliability_matches_oracle(n_bins):
  from weatherbench2_amd import metrics as gm
  from weatherbench2_amd import xarray_lite as xl
  rng = np.random.default_rng(n_bins)
  hist = rng.random((2, n_bins, 3))
  hist /= hist.sum(1, keepdims=True)
  da = xl.DataArray(hist, ('level', 'bins', 'region'),
                    {'level': np.arange(2), 'bins': np.arange(n_bins)}, 'z')
  got = gm.central_reliability(da)
  probs, desired = om.central_reliability(np.moveaxis(hist, 1, -1))
  assert got.dims == (('desired_prob', 'level', 'region') if n_bins % 2
                      else ('level', 'desired_prob', 'region'))
  np.testing.assert_allclose(
      np.moveaxis(np.asarray(got.values), got.dims.index('desired_prob'), -1),
      probs, rtol=1e-15)
  np.testing.assert_allclose(got.coords['desired_prob'], desired, rtol=1e-15)
  assert np.asarray(got.values).max() <= 1 + 1e-12
  # Dataset in, Dataset out
  ds = gm.central_reliability(xl.Dataset({'z': da}, da.coords))
  np.testing.assert_array_equal(np.asarray(ds['z'].values),
                                np.asarray(got.values))


@pytest.mark.parametrize('hist,expected,desired', [
    # metrics_test.py:700-779, through the product's host function
    ([0.2, 0.1, 0.7], [0.1, 1.0], [1 / 3, 1.0]),
    ([0.2, 0.0, 0.1, 0.1, 0.6], [0.1, 0.2, 1.0], [1 / 5, 3 / 5, 1.0]),
    ([0.1, 0.1, 0.5, 0.3], [0.6, 1.0], [1 / 2, 1.0]),
    ([0.1, 0.1, 0.3, 0.2, 0.0, 0.3], [0.5, 0.6, 1.0], [1 / 3, 2 / 3, 1.0]),
])
def test_central_reliability_reference_vectors(hist, expected, desired):
  from weatherbench2_amd import metrics as gm
  from weatherbench2_amd import xarray_lite as xl
  ds = xl.Dataset({'temperature': xl.DataArray(np.array(hist), ('bins',))},
                  {'bins': np.arange(len(hist))})
  got = gm.central_reliability(ds)['temperature']
  assert got.dims == ('desired_prob',)
  np.testing.assert_allclose(np.asarray(got.values), expected, rtol=1e-12)
  np.testing.assert_allclose(got.coords['desired_prob'], desired, rtol=1e-12)


def test_central_reliability_too_few_bins():
  from weatherbench2_amd import metrics as gm
  from weatherbench2_amd import xarray_lite as xl
  with pytest.raises(ValueError, match='Too few bins'):
    gm.central_reliability(xl.DataArray(np.ones(2) / 2, ('bins',),
                                        {'bins': np.arange(2)}, 'z'))


@pytest.mark.gpu
@pytest.mark.parametrize('cutoff_below', [True, False])
@pytest.mark.parametrize('ensemble_size', [1, 2, 3, 10])
def test_repeated_entries_get_random_bin(ensemble_size, cutoff_below):
  # metrics_test.py:603-650: truth and members share a point mass at 0, so
  # most samples are PARTIALLY tied; the histogram must still be flat.
  from tests.test_oracle_golden import _censored_case
  from weatherbench2_amd import metrics as gm
  num_bins = ensemble_size + 1
  truth, forecast = _censored_case(ensemble_size, cutoff_below)
  g = helpers.to_gpu_dataset
  da = gm.RankHistogram(num_bins=num_bins, seed=802701).compute_chunk(
      g(forecast), g(truth))['geopotential']
  v = _values(da)
  sample_size = v.size / num_bins
  rtol = 5 * (num_bins - 1) / np.sqrt(sample_size)
  np.testing.assert_allclose(v.reshape(-1, num_bins).mean(0), 1 / num_bins,
                             rtol=rtol)


@pytest.mark.gpu
def test_rank_histogram_c_abi_edge_shapes():
  """n_point not a multiple of the wave size, more bins than lanes, one
  member, float64 -- straight through engine.rank_histogram."""
  import torch
  from weatherbench2_amd import engine
  dev = torch.device('cuda', 0)
  gen = torch.Generator(device=dev).manual_seed(2)
  for dtype in (torch.float32, torch.float64):
    for n_member, n_bins, n_outer, n_point in [(1, 2, 1, 1), (3, 4, 2, 65),
                                               (7, 2, 3, 130), (99, 100, 1, 77),
                                               (127, 64, 2, 64)]:
      ens = torch.randn((n_member, n_outer, n_point), device=dev, dtype=dtype,
                        generator=gen)
      truth = torch.randn((n_outer, n_point), device=dev, dtype=dtype,
                          generator=gen)
      out = engine.rank_histogram(ens, n_outer * n_point, n_member, None,
                                  truth, None, n_outer, n_point, n_bins, True, 5)
      rank = (ens < truth[None]).sum(0) // ((n_member + 1) // n_bins)
      want = torch.nn.functional.one_hot(rank, n_bins).double()
      torch.testing.assert_close(out, want, rtol=0, atol=0)
      # accumulate mode over the outer axis
      rows = torch.zeros((n_outer,), dtype=torch.int64, device=dev)
      acc = engine.rank_histogram(ens, n_outer * n_point, n_member, None,
                                  truth, None, n_outer, n_point, n_bins, True,
                                  5, rows, 1)
      torch.testing.assert_close(acc[0], want.sum(0), rtol=0, atol=0)


def _tied(ds, step, seed):
  """Quantises the values so that ties between truth and members are common."""
  return ds.copy(data={k: (np.round(v.data / step) * step).astype(v.data.dtype)
                       for k, v in ds.items()})


@pytest.mark.gpu
@pytest.mark.parametrize('ensemble_size,num_bins,dtype,step', [
    (1, None, np.float32, 1.0), (2, None, np.float64, 0.5),
    (5, 3, np.float32, 0.5), (10, None, np.float32, 0.25),
    (50, None, np.float32, 0.5), (50, 17, np.float64, 1.0)])
def test_seeded_ties_reproduce_numpy_stream(ensemble_size, num_bins, dtype, step):
  """With a seed the reference breaks ties with
  np.random.default_rng(seed).uniform over the concatenated [truth, members]
  array (metrics.py:1955-1980).  The kernel jumps PCG64 to each element's place
  in that stream: one-hot outputs EQUAL the oracle's (which runs NumPy's RNG
  itself) on heavily tied data -- the mock layout has a lead-time dim only the
  forecast carries, so the member stride in the stream is not 1."""
  from weatherbench2_amd import metrics as gm
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=ensemble_size, spatial_resolution_in_degrees=20,
      time_start='2019-12-01', time_stop='2019-12-03', levels=(0, 1))
  truth = _tied(_as(truth, dtype), step, 0)
  forecast = _tied(_as(forecast, dtype), step, 1)
  for seed in (0, 802701):
    want = om.RankHistogram(num_bins=num_bins, seed=seed).compute_chunk(
        forecast, truth)
    g = helpers.to_gpu_dataset
    got = gm.RankHistogram(num_bins=num_bins, seed=seed).compute_chunk(
        g(forecast), g(truth))
    for name in want.keys():
      assert got[name].dims == want[name].dims
      np.testing.assert_array_equal(_values(got[name]), want[name].data,
                                    err_msg=f'seed {seed}')
  # the ties are really there (otherwise this test would prove nothing)
  f = forecast['geopotential']
  t = NA._align(f, truth['geopotential'])[1]
  assert (f.data == t).any(axis=f.dims.index('realization')).mean() > 0.2


@pytest.mark.gpu
@pytest.mark.parametrize('layout', ['ens_last_dims_match', 'ens_middle'])
def test_seeded_ties_other_layouts_nan_and_inf(layout):
  """Truth with the forecast's dims (stride 1 between members in the stream) and
  an ensemble dim in the middle; NaN members / truth and repeated infinities
  switch the reference to its +-1/4 perturbation (min_diff is NaN), which can
  reorder DISTINCT values: reproduced too."""
  from weatherbench2_amd import metrics as gm
  rs = np.random.RandomState(4)
  n_t, n_m, n_lat, n_lon = 3, 7, 6, 9
  lat = np.linspace(-75, 75, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  if layout == 'ens_last_dims_match':
    fdims = ('time', 'latitude', 'longitude', 'realization')
    tdims = ('time', 'latitude', 'longitude')
  else:
    fdims = ('time', 'realization', 'longitude', 'latitude')
    tdims = ('time', 'longitude', 'latitude')
  sizes = {'time': n_t, 'realization': n_m, 'latitude': n_lat,
           'longitude': n_lon}
  f = np.round(rs.standard_normal([sizes[d] for d in fdims]) * 2) / 4
  t = np.round(rs.standard_normal([sizes[d] for d in tdims]) * 2) / 4
  f = f.astype(np.float32)
  t = t.astype(np.float32)
  f[rs.rand(*f.shape) < 0.05] = np.nan
  t[rs.rand(*t.shape) < 0.05] = np.nan
  f[rs.rand(*f.shape) < 0.05] = np.inf
  f[rs.rand(*f.shape) < 0.03] = -np.inf
  t[rs.rand(*t.shape) < 0.03] = np.inf
  coords = {'time': np.arange(n_t), 'realization': np.arange(n_m),
            'latitude': lat, 'longitude': lon}
  forecast = DS({'z': NA(f, fdims)}, coords)
  truth = DS({'z': NA(t, tdims)}, {k: v for k, v in coords.items()
                                   if k != 'realization'})
  want = om.RankHistogram(seed=5).compute_chunk(forecast, truth)['z']
  g = helpers.to_gpu_dataset
  got = gm.RankHistogram(seed=5).compute_chunk(g(forecast), g(truth))['z']
  assert got.dims == want.dims
  a, w = _values(got), want.data
  # Where the perturbed values of truth and a member are EXACTLY equal --
  # a NaN truth next to NaN members, an infinite truth next to an equal
  # infinity (inf + perturbation == inf) -- np.argsort's order among the equal
  # elements is unspecified (introsort); the product puts the truth first.
  # Everything else must match bit for bit.
  ens_ax = fdims.index('realization')
  t_b = np.expand_dims(t, ens_ax)
  clash = (np.isinf(t_b) & (f == t_b)).any(axis=ens_ax)
  ok = ~np.isnan(t) & ~clash
  assert ok.mean() > 0.8 and clash.sum() >= 1
  np.testing.assert_array_equal(a[ok], w[ok])
  np.testing.assert_array_equal(a.sum(-1), 1.0)


@pytest.mark.gpu
def test_seeded_temporal_mean_matches_oracle():
  from weatherbench2_amd import metrics as gm
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=4, spatial_resolution_in_degrees=20,
      time_start='2019-12-01', time_stop='2019-12-05', levels=(0,))
  truth, forecast = _tied(truth, 0.5, 0), _tied(forecast, 0.5, 1)
  want = om.RankHistogram(seed=9).compute(forecast, truth)['geopotential']
  g = helpers.to_gpu_dataset
  got = gm.RankHistogram(seed=9).compute(g(forecast), g(truth))['geopotential']
  assert got.dims == want.dims
  np.testing.assert_allclose(_values(got), want.data, rtol=0, atol=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype_name', ['float32', 'float64'])
def test_mean_kernel_equals_the_atomic_counts(dtype_name):
  """wb2_rank_histogram_mean (a wave owns 64 points of a result row, counts in
  LDS) against the atomic-add form over the same rows, for the averaged axis
  first, in the middle and last, ragged point counts, ties (quantised values:
  the same counter-based draws per sample) -- the same bits."""
  import torch
  from weatherbench2_amd import engine
  dev = torch.device('cuda', 0)
  dtype = getattr(torch, dtype_name)
  gen = torch.Generator(device=dev).manual_seed(4)
  for n_member, n_bins, shape, axis, n_point in [
      (5, 6, (4, 3), 0, 130), (5, 3, (2, 5, 3), 1, 64), (9, 10, (3, 7), 1, 77),
      (50, 51, (6, 2), 0, 257), (1, 2, (1, 1, 1), 1, 1), (255, 256, (3,), 0, 70)]:
    n_outer = int(np.prod(shape))
    ens = (torch.randn((n_member, n_outer, n_point), device=dev, dtype=dtype,
                       generator=gen) * 4).round() / 4
    truth = (torch.randn((n_outer, n_point), device=dev, dtype=dtype,
                         generator=gen) * 4).round() / 4
    ens[0, 0, 0] = float('nan')
    kept = tuple(n for i, n in enumerate(shape) if i != axis)
    n_acc = int(np.prod(kept)) if kept else 1
    rows = np.arange(n_acc, dtype=np.int64).reshape(kept)
    rows = np.broadcast_to(np.expand_dims(rows, axis), shape)
    acc_row = torch.from_numpy(np.array(rows).ravel()).to(dev)
    counts = engine.rank_histogram(ens, n_outer * n_point, n_member, None,
                                   truth, None, n_outer, n_point, n_bins, True,
                                   77, acc_row, n_acc)
    mean_over = (int(np.prod(shape[:axis])), shape[axis],
                 int(np.prod(shape[axis + 1:])))
    mean = engine.rank_histogram(ens, n_outer * n_point, n_member, None, truth,
                                 None, n_outer, n_point, n_bins, True, 77,
                                 mean_over=mean_over)
    assert mean.shape == counts.shape
    # true division (NumPy's mean): torch turns `/ python scalar` into a
    # multiplication by the reciprocal, tensor / tensor divides
    want = counts / torch.full_like(counts, float(shape[axis]))
    assert torch.equal(mean, want), (n_member, shape, axis)
    assert float(mean.sum()) == pytest.approx(n_acc * n_point, rel=1e-12)
