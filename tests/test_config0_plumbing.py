"""BASELINE configs[0]: synthetic 64 x 32 single-level single-init WeightedRMSE
(the reference's own CPU-runnable case; SURVEY 8d "Config 1").  Latitudes are
the 32 equiangular points WITHOUT poles of the 64x32 datasets
(docs/source/official-evaluation.md:13-15), truth = RandomState(0).normal,
forecast = RandomState(1).normal in float64 (evaluation_test.py:46,55).

CPU: the oracle runs it (the reference's xarray path cannot run here) and the
result has the value the construction implies.  GPU: same number from the HIP
path, through compute_chunk and through the loop."""
import numpy as np
import pytest

from oracle import metrics_np as om
from oracle.named import DS, NA
from tests import helpers


def config0():
  n_lon, n_lat = 64, 32
  lat = -90 + 180 / n_lat * (np.arange(n_lat) + 0.5)  # -87.1875 ... 87.1875
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  dims = ('time', 'prediction_timedelta', 'level', 'longitude', 'latitude')
  shape = (1, 1, 1, n_lon, n_lat)
  coords = {'time': np.array(['2020-01-01'], dtype='datetime64[ns]'),
            'prediction_timedelta': np.array([0], dtype='timedelta64[ns]'),
            'level': np.array([500]), 'longitude': lon, 'latitude': lat}
  truth = DS({'geopotential': NA(
      np.random.RandomState(0).normal(size=shape), dims)}, coords)
  forecast = DS({'geopotential': NA(
      np.random.RandomState(1).normal(size=shape), dims)}, coords)
  return forecast, truth


def test_config0_runs_on_the_oracle():
  forecast, truth = config0()
  assert abs(truth.coord('latitude')[0] + 87.1875) < 1e-12
  rmse = om.RMSESqrtBeforeTimeAvg().compute_chunk(forecast, truth)
  mse = om.MSE().compute_chunk(forecast, truth)
  v = rmse['geopotential'].data
  assert v.shape == (1, 1, 1) and np.isfinite(v).all()
  np.testing.assert_allclose(v, np.sqrt(mse['geopotential'].data), rtol=1e-15)
  # independent N(0,1) fields: the weighted MSE of the difference is ~2
  assert 1.7 < float(mse['geopotential'].data.ravel()[0]) < 2.3
  # the closed form of the weighted average, written out
  d = forecast['geopotential'].data - truth['geopotential'].data
  w = om.get_lat_weights(truth.coord('latitude')).data
  want = np.sqrt((d[0, 0, 0] ** 2 * w[None, :]).sum() / (w.sum() * 64))
  np.testing.assert_allclose(v.ravel()[0], want, rtol=1e-13)


@pytest.mark.gpu
def test_config0_on_the_gpu_matches_the_oracle():
  from weatherbench2_amd import config, evaluation, metrics as gm
  forecast, truth = config0()
  g = helpers.to_gpu_dataset
  want = om.RMSESqrtBeforeTimeAvg().compute_chunk(forecast, truth)
  got = gm.RMSESqrtBeforeTimeAvg().compute_chunk(g(forecast), g(truth))
  np.testing.assert_allclose(got['geopotential'].values,
                             want['geopotential'].data, rtol=1e-13)
  cfg = config.Eval(metrics={'rmse': gm.RMSESqrtBeforeTimeAvg()})
  loop = evaluation._metric_and_region_loop(g(forecast), g(truth), cfg, False)
  np.testing.assert_allclose(np.asarray(loop['geopotential'].values).ravel(),
                             want['geopotential'].data.ravel(), rtol=1e-13)
