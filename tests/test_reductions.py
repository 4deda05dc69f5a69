"""Averaging pipelines: the oracle against plain NumPy (CPU) and
the HIP path against the oracle (-m gpu).  Expectations follow the reference's
own script tests (scripts/compute_averages_test.py:31-104,
scripts/compute_ensemble_mean_test.py:28-82)."""
import numpy as np
import pytest

from oracle import fixtures
from oracle import metrics_np as om
from oracle import reductions_np as ored
from tests import helpers


def _values(da):
  v = da.data
  return v.cpu().numpy() if hasattr(v, 'cpu') else np.asarray(v)


def _case(nan_frac=0.0, dtype=np.float32):
  ds = fixtures.random_like(fixtures.mock_forecast_data(
      ensemble_size=3, variables_3d=['geopotential', 'temperature'],
      variables_2d=['2m_temperature'], time_start='2020-06-01',
      time_stop='2020-06-09', lead_stop='2 day',
      spatial_resolution_in_degrees=15), seed=11)
  if nan_frac:
    ds = fixtures.insert_nan(ds, nan_frac, seed=12)
  return ds.copy(data={k: v.data.astype(dtype) for k, v in ds.items()})


# ---- oracle vs the expressions the reference tests compare against ---------
def test_oracle_ensemble_mean_is_mean_over_realization():
  ds = _case()
  got = ored.ensemble_mean(ds)
  for k, v in ds.items():
    want = v.data.mean(axis=v.dims.index('realization'))
    np.testing.assert_array_equal(got[k].data, want)
    assert 'realization' not in got[k].dims


def test_oracle_averages_time_longitude_and_time_latitude():
  ds = _case()
  got = ored.averages(ds, ['time', 'longitude'])
  v = ds['geopotential']
  want = v.data.mean(axis=(v.dims.index('time'), v.dims.index('longitude')))
  np.testing.assert_allclose(got['geopotential'].data, want, rtol=1e-6)
  # compute_averages_test.py:93-102: .mean(time).weighted(w).mean(latitude)
  got = ored.averages(ds, ['time', 'latitude'])
  w = om.get_lat_weights(ds.coord('latitude')).data
  ax_t, ax_l = v.dims.index('time'), v.dims.index('latitude')
  tm = v.data.astype(np.float64).mean(axis=ax_t)
  ax_l2 = ax_l - (1 if ax_t < ax_l else 0)
  want = np.tensordot(tm, w, axes=([ax_l2], [0])) / w.sum()
  np.testing.assert_allclose(got['geopotential'].data, want, rtol=1e-6)
  assert got['geopotential'].data.dtype == np.float64


def test_oracle_statistical_moments():
  ds = _case(nan_frac=0.05)
  got = ored.statistical_moments(ds)
  v = ds['temperature']
  axes = (v.dims.index('latitude'), v.dims.index('longitude'))
  np.testing.assert_allclose(got['temperature_zeroth'].data,
                             (~np.isnan(v.data)).mean(axis=axes))
  np.testing.assert_allclose(got['temperature_first'].data,
                             np.nanmean(v.data, axis=axes), rtol=1e-6)
  np.testing.assert_allclose(got['temperature_second'].data,
                             np.nanmean(v.data ** 2, axis=axes), rtol=1e-6)


# ---- HIP path vs the oracle --------------------------------------------------
def _close(got, want, rtol, err_msg=''):
  g = _values(got)
  w = want.transpose(*got.dims).data
  assert g.dtype == w.dtype, (g.dtype, w.dtype, err_msg)
  helpers.assert_close(g, w, rtol=rtol, atol=1e-12 if w.dtype == np.float64
                       else 1e-6, err_msg=err_msg)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype,rtol', [(np.float32, 2e-6), (np.float64, 1e-12)])
@pytest.mark.parametrize('skipna', [False, True])
def test_ensemble_mean_matches_oracle(dtype, rtol, skipna):
  from weatherbench2_amd import reductions
  ds = _case(nan_frac=0.02 if skipna else 0.0, dtype=dtype)
  want = ored.ensemble_mean(ds, skipna=skipna)
  got = reductions.ensemble_mean(helpers.to_gpu_dataset(ds), skipna=skipna)
  assert set(got.keys()) == set(want.keys())
  for k in want.keys():
    assert 'realization' not in got[k].dims
    _close(got[k], want[k], rtol, k)
  with pytest.raises(ValueError):
    reductions.ensemble_mean(helpers.to_gpu_dataset(ds), 'number')


@pytest.mark.gpu
@pytest.mark.parametrize('dims', [['time', 'longitude'], ['time', 'latitude'],
                                  ['latitude'], ['latitude', 'longitude'],
                                  ['realization', 'level'],
                                  ['prediction_timedelta', 'time', 'longitude',
                                   'latitude']])
@pytest.mark.parametrize('skipna', [False, True])
def test_averages_match_oracle(dims, skipna):
  from weatherbench2_amd import reductions
  ds = _case(nan_frac=0.02 if skipna else 0.0)
  want = ored.averages(ds, dims, skipna=skipna)
  got = reductions.averages(helpers.to_gpu_dataset(ds), dims, skipna=skipna)
  for k in want.keys():
    assert not set(dims) & set(got[k].dims)
    _close(got[k], want[k], 2e-6, f'{k} over {dims}')
  for d in ('level', 'longitude'):
    if d in dims:
      assert d not in got.coords
    else:
      np.testing.assert_array_equal(np.asarray(got.coords[d]),
                                    np.asarray(ds.coords[d]))


@pytest.mark.gpu
def test_statistical_moments_match_oracle_and_are_deterministic():
  from weatherbench2_amd import reductions
  ds = _case(nan_frac=0.05)
  want = ored.statistical_moments(ds)
  g = helpers.to_gpu_dataset(ds)
  got = reductions.statistical_moments(g)
  again = reductions.statistical_moments(g)
  for k in want.keys():
    a = _values(got[k])
    helpers.assert_close(a, want[k].transpose(*got[k].dims).data, rtol=2e-6,
                         atol=1e-7, err_msg=k)
    np.testing.assert_array_equal(a, _values(again[k]))
  # time mean of the spatial moments (compute_statistical_moments.py:98-110)
  tm = reductions.mean(got, 'time', skipna=True)
  w = want['temperature_first']
  np.testing.assert_allclose(
      _values(tm['temperature_first']),
      np.nanmean(w.transpose(*got['temperature_first'].dims).data,
                 axis=got['temperature_first'].dims.index('time')), rtol=2e-6,
      atol=1e-7)


@pytest.mark.gpu
def test_quarter_degree_field_means_split_the_reduced_axis():
  """A 721 x 1440 global mean per level: the contiguous kernel with the reduced
  axis cut into slices, against an fp64 torch reduction."""
  import torch
  from weatherbench2_amd import reductions, xarray_lite as xl
  dev = torch.device('cuda', 0)
  x = torch.randn((13, 721, 1440), device=dev,
                  generator=torch.Generator(device=dev).manual_seed(5))
  lat = np.linspace(-90, 90, 721)
  lon = np.linspace(0, 360, 1440, endpoint=False)
  ds = xl.Dataset({'z': xl.DataArray(x, ('level', 'latitude', 'longitude'))},
                  {'level': np.arange(13), 'latitude': lat, 'longitude': lon})
  got = reductions.averages(ds, ['latitude', 'longitude'])['z']
  from weatherbench2_amd import plan
  w = torch.as_tensor(plan.get_lat_weights(lat), device=dev)
  want = (x.double() * w[None, :, None]).mean((1, 2))
  torch.testing.assert_close(got.data, want, rtol=1e-12, atol=1e-14)
  # members leading, huge tail: the strided kernel
  ens = torch.randn((7, 4, 721, 1440), device=dev)
  e = xl.Dataset({'z': xl.DataArray(ens, ('realization', 'level', 'latitude',
                                          'longitude'))},
                 {'level': np.arange(4), 'latitude': lat, 'longitude': lon})
  m = reductions.ensemble_mean(e)['z']
  assert m.data.dtype == torch.float32
  torch.testing.assert_close(m.data, ens.double().mean(0).float(), rtol=1e-6,
                             atol=1e-7)


@pytest.mark.gpu
def test_axis_moments_edge_shapes():
  """Ragged sizes through the C ABI: unaligned views, tails that are not a
  multiple of the vector width, single-element axes, float64, empty reduce."""
  import torch
  from weatherbench2_amd import engine
  dev = torch.device('cuda', 0)
  gen = torch.Generator(device=dev).manual_seed(3)
  for dtype in (torch.float32, torch.float64):
    for n_lead, n_red, n_tail in [(1, 1, 1), (3, 7, 1), (2, 5, 3), (1, 129, 67),
                                  (5, 4099, 1), (2, 33, 1026), (1, 1, 4096)]:
      big = torch.randn((n_lead * n_red * n_tail + 1,), device=dev, dtype=dtype,
                        generator=gen)
      for off in (0, 1):  # off = 1: not 16-byte aligned -> scalar loads
        x = big[off:off + n_lead * n_red * n_tail]
        x = x.clone() if off == 0 else x
        xs = x.reshape(n_lead, n_red, n_tail)
        xs_nan = xs.clone()
        xs_nan[..., ::3, :] = float('nan') if n_red > 2 else xs_nan[..., ::3, :]
        w = torch.rand((n_red,), device=dev, dtype=torch.float64,
                       generator=gen)
        for data, skipna in ((xs, False), (xs_nan, True)):
          flat = data.reshape(-1)
          if off == 1:  # keep the misaligned storage
            big[1:1 + flat.numel()] = flat
            flat = big[1:1 + flat.numel()]
          s, q, c = engine.axis_moments(flat, n_lead, n_red, n_tail, w, skipna,
                                        True)
          d = data.double()
          ok = ~torch.isnan(d) if skipna else torch.ones_like(d, dtype=torch.bool)
          z = torch.where(ok, d, torch.zeros_like(d))
          ws = w[None, :, None]
          torch.testing.assert_close(s.reshape(n_lead, n_tail), (z * ws).sum(1),
                                     rtol=1e-12, atol=1e-12)
          z2 = torch.where(ok, (data * data).double(), torch.zeros_like(d))
          torch.testing.assert_close(q.reshape(n_lead, n_tail), (z2 * ws).sum(1),
                                     rtol=1e-12, atol=1e-12)
          torch.testing.assert_close(c.reshape(n_lead, n_tail),
                                     ok.double().sum(1), rtol=0, atol=0)
  # nothing to reduce: sums 0, counts 0 (mean = NaN like an empty xarray mean)
  empty = torch.empty((0,), device=dev)
  s, _, c = engine.axis_moments(empty, 4, 0, 1, None, True)
  assert float(s.abs().sum()) == 0.0 and float(c.sum()) == 0.0
