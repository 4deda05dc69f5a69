"""Regression tests for the round-5 advisor findings, on the GPU.

* chunk programs + `derived_variables` over more than one chunk
  (evaluation.py:402-405: the loop assigns them into every chunk);
* a dict of configs must not leak one config's derived variables into the
  next (every pipeline branch of the reference reads the chunk itself,
  evaluation.py:757-828);
* an ensemble chunk that is a strided view AND needs a cast;
* the fetch thread's uploader goes away with the thread;
* engine.energy_score's contiguity error.
"""
import gc

import numpy as np
import pytest

from oracle import metrics_np as om
from tests import helpers, official_chunks as oc

pytestmark = pytest.mark.gpu


class _Doubled:  # a user-defined DerivedVariable (duck-typed protocol)
  base_variables = ['geopotential']

  def compute(self, dataset):
    return dataset['geopotential'] * 2.0


def _same(a, b):
  assert sorted(a.data_vars) == sorted(b.data_vars)
  for name in a.data_vars:
    x, y = np.asarray(a[name].values), np.asarray(b[name].values)
    assert a[name].dims == b[name].dims and x.dtype == y.dtype
    assert np.array_equal(x, y, equal_nan=True), name


def _chunks(resident: bool):
  from weatherbench2_amd import evaluation
  forecast, truth, _ = oc.make(n_init=4, n_lead=2)
  gf, gt = helpers.to_gpu_dataset(forecast), helpers.to_gpu_dataset(truth)
  if resident:
    gf, gt = evaluation.make_resident(gf), evaluation.make_resident(gt)
  return forecast, truth, gf, gt


@pytest.mark.parametrize('resident', [True, False])
@pytest.mark.parametrize('how', ['1', 'verify'])
def test_derived_variables_over_several_chunks(resident, how, monkeypatch):
  """Chunk 2 of a structure used to be replayed by a program that looked for
  the derived variable in a chunk that did not have it yet (KeyError)."""
  from weatherbench2_amd import config, evaluation, metrics as gm
  forecast, truth, gf, gt = _chunks(resident)
  cfg = config.Eval(metrics={'mse': gm.MSE(), 'mae': gm.MAE()},
                    derived_variables={'doubled': _Doubled()})
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', '0')
  want = evaluation.evaluate_chunks(oc.chunk_pairs(gf, gt), cfg, False,
                                    batch_chunks=1)
  monkeypatch.setenv('WB2HIP_CHUNK_PROGRAM', how)
  forecast, truth, gf, gt = _chunks(resident)   # (the loop assigns in place)
  got = evaluation.evaluate_chunks(oc.chunk_pairs(gf, gt), cfg, False,
                                   batch_chunks=1)
  _same(got, want)
  assert 'doubled' in got.data_vars
  labels = list(got.coords['metric'])
  # against the oracle: MSE(2 z) = 4 MSE(z), time mean of the chunk values
  per = om.MSE().compute_chunk(forecast, truth)['geopotential']
  ax = per.dims.index('init_time')
  mean = np.asarray(per.data, dtype=np.float64).mean(axis=ax)
  res = got['doubled']
  order = [d for d in res.dims if d != 'metric']
  dims = [d for d in per.dims if d != 'init_time']
  vals = np.transpose(res.values[labels.index('mse')],
                      [order.index(d) for d in dims])
  helpers.assert_close(vals, 4.0 * mean, rtol=1e-6)


def test_configs_do_not_see_each_others_derived_variables():
  from weatherbench2_amd import config, evaluation, metrics as gm
  _, _, gf, gt = _chunks(True)
  with_dv = config.Eval(metrics={'mse': gm.MSE()},
                        derived_variables={'doubled': _Doubled()})
  without = config.Eval(metrics={'mae': gm.MAE()})
  got = evaluation.evaluate_chunks(oc.chunk_pairs(gf, gt),
                                   {'a': with_dv, 'b': without}, False,
                                   batch_chunks=1)
  assert 'doubled' in got['a'].data_vars
  assert 'doubled' not in got['b'].data_vars
  alone = evaluation.evaluate_chunks(oc.chunk_pairs(*_chunks(True)[2:]),
                                     without, False, batch_chunks=1)
  _same(got['b'], alone)


def test_strided_ensemble_view_that_needs_a_cast():
  """A members-leading chunk sliced out of a resident float32 forecast, with a
  float64 truth: the cast makes a compact copy -- the pass must address THAT
  (it used to keep the view's strides: out-of-bounds reads)."""
  import torch
  from weatherbench2_amd import metrics as gm
  from weatherbench2_amd import xarray_lite as xl
  dev = torch.device('cuda', 0)
  rs = np.random.RandomState(5)
  n_mem, n_init, n_lat, n_lon = 5, 3, 9, 16
  lat, lon = np.linspace(-80, 80, n_lat), np.arange(n_lon) * 22.5
  whole = rs.normal(size=(n_mem, n_init, n_lat, n_lon)).astype(np.float32)
  truth = rs.normal(size=(n_init, n_lat, n_lon))          # float64
  res = torch.as_tensor(whole, device=dev)
  time = np.datetime64('2020-01-01', 'ns') + np.arange(n_init) * np.timedelta64(
      1, 'D')
  for i in range(n_init):
    view = res[:, i:i + 1]                   # strided: members leading
    assert not view.is_contiguous()
    coords = {'realization': np.arange(n_mem), 'time': time[i:i + 1],
              'latitude': lat, 'longitude': lon}
    f = xl.Dataset({'z': xl.DataArray(
        view, ('realization', 'time', 'latitude', 'longitude'))}, coords)
    t_ = xl.Dataset({'z': xl.DataArray(
        truth[i:i + 1], ('time', 'latitude', 'longitude'))},
                    {k: v for k, v in coords.items() if k != 'realization'})
    got = gm.CRPS(ensemble_dim='realization').compute_chunk(f, t_)['z'].values
    from oracle.named import DS, NA
    of = DS({'z': NA(whole[:, i:i + 1].astype(np.float64),
                     ('realization', 'time', 'latitude', 'longitude'))},
            coords)
    ot = DS({'z': NA(truth[i:i + 1], ('time', 'latitude', 'longitude'))},
            {k: v for k, v in coords.items() if k != 'realization'})
    want = om.CRPS(ensemble_dim='realization').compute_chunk(of, ot)['z'].data
    helpers.assert_close(got, want, rtol=1e-9)


def test_fetch_thread_uploaders_do_not_pile_up():
  """evaluate_chunks starts a fetch thread per call; its uploader (4 x 32 MiB
  of pinned memory, copy threads, events) must go with it."""
  from weatherbench2_amd import config, evaluation, feeder, metrics as gm
  _, _, gf, gt = _chunks(False)
  # host chunks big enough to be staged (>= 1 MiB per variable)
  import dataclasses
  forecast, truth, _ = oc.make(n_init=3, n_lead=1, n_lat=181, n_lon=360)
  gf, gt = helpers.to_gpu_dataset(forecast), helpers.to_gpu_dataset(truth)
  cfg = config.Eval(metrics={'mse': gm.MSE()})
  del dataclasses
  gc.collect()
  others = feeder.UPLOADERS_ALIVE()   # (of threads of earlier tests)
  evaluation.evaluate_chunks(oc.chunk_pairs(gf, gt), cfg, False, prefetch=2)
  gc.collect()
  before = feeder.UPLOADERS_ALIVE()
  for _ in range(3):
    evaluation.evaluate_chunks(oc.chunk_pairs(gf, gt), cfg, False, prefetch=2)
  gc.collect()
  assert feeder.UPLOADERS_ALIVE() <= before <= others + 1, (
      others, before, feeder.UPLOADERS_ALIVE())


def test_energy_score_rejects_a_gappy_ensemble_with_a_value_error():
  import torch
  from weatherbench2_amd import engine, plan as plan_lib
  dev = torch.device('cuda', 0)
  lat, lon = np.linspace(-80, 80, 9), np.arange(16) * 22.5
  pl = plan_lib.build_plan(lat, lon, plan_lib.LATLON, None, dev,
                           rows_per_chunk=4)
  ens = torch.zeros((4, 2, 9, 16), dtype=torch.float32, device=dev)[:, :1]
  truth = torch.zeros((1, 9, 16), dtype=torch.float32, device=dev)
  assert not ens.is_contiguous()
  with pytest.raises(ValueError, match='contiguous'):
    engine.energy_score(pl, ens, 2 * 9 * 16, 4, None, truth, None, 1, False)
