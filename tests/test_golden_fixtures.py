"""Committed golden vectors (tests/golden/, see make_golden.py for provenance).

CPU: the oracle still reproduces the frozen vectors bit for bit and the
transcribed reference known answers.  GPU: the HIP path matches the frozen
vectors (so a simultaneous drift of oracle and kernels cannot go unnoticed).
"""
import json
import os

import numpy as np
import pytest

from oracle import metrics_np as om
from oracle import spectrum_np
from tests import helpers
from tests.golden import make_golden as mg

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def vectors():
  return np.load(os.path.join(HERE, 'oracle_vectors_v1.npz'))


@pytest.fixture(scope='module')
def known():
  return json.load(open(os.path.join(HERE, 'reference_known_answers.json')))


def test_oracle_reproduces_the_frozen_vectors(vectors):
  fresh = mg.oracle_vectors()
  assert set(fresh) == set(vectors.files)
  for k in vectors.files:
    if k.endswith('/dims'):
      assert list(fresh[k]) == list(vectors[k]), k
    else:
      np.testing.assert_array_equal(fresh[k], vectors[k], err_msg=k)


def test_reference_known_answers_hold_for_the_oracle(known):
  ka = known['lat_weights']
  w = om.get_lat_weights(np.array(ka['latitude'], dtype=np.float64)).data
  np.testing.assert_allclose(w, ka['expected'], rtol=1e-7)
  for case in known['central_reliability']['cases']:
    probs, desired = om.central_reliability(np.array(case['hist']))
    np.testing.assert_allclose(probs, case['prob'], rtol=1e-12)
    np.testing.assert_allclose(desired, case['desired'], rtol=1e-12)


@pytest.mark.gpu
def test_hip_deterministic_metrics_match_the_frozen_vectors(vectors):
  from weatherbench2_amd import metrics as gm
  truth, forecast, clim = mg.deterministic_case()
  g = helpers.to_gpu_dataset
  gregions = {k: helpers.to_gpu_region(v) for k, v in mg.regions().items()}
  with gm.fused_regions(gregions):
    for rname, region in gregions.items():
      for mname in mg.DET:
        metric = gm.ACC(g(clim)) if mname == 'ACC' else getattr(gm, mname)()
        got = metric.compute_chunk(g(forecast), g(truth),
                                   region=region)['geopotential']
        key = f'det/{mname}/{rname}'
        assert list(got.dims) == list(vectors[key + '/dims'])
        helpers.assert_close(got.values, vectors[key], rtol=1e-9, atol=1e-12,
                             err_msg=key)


@pytest.mark.gpu
def test_hip_ensemble_metrics_match_the_frozen_vectors(vectors):
  from weatherbench2_amd import metrics as gm
  truth, forecast = mg.ensemble_case()
  g = helpers.to_gpu_dataset
  all_regions = mg.regions()
  gregions = {k: helpers.to_gpu_region(all_regions[k])
              for k in ('global', 'europe')}
  with gm.fused_regions(gregions):
    for rname, region in gregions.items():
      for mname in mg.ENS:
        got = getattr(gm, mname)().compute_chunk(
            g(forecast), g(truth), region=region)['geopotential']
        key = f'ens/{mname}/{rname}'
        assert list(got.dims) == list(vectors[key + '/dims'])
        helpers.assert_close(got.values, vectors[key], rtol=2e-6, atol=1e-7,
                             err_msg=key)


@pytest.mark.gpu
def test_hip_spectrum_matches_the_frozen_vectors(vectors):
  from weatherbench2_amd import derived_variables as dv
  from weatherbench2_amd import xarray_lite as xl
  x, lat, lon = mg.spectrum_case()
  ds = xl.Dataset({'z': xl.DataArray(x, ('time', 'latitude', 'longitude'))},
                  {'latitude': lat, 'longitude': lon})
  got = dv.ZonalEnergySpectrum('z').compute(ds)
  np.testing.assert_allclose(got.values, vectors['spectrum/values'],
                             rtol=1e-10, atol=1e-6)
  np.testing.assert_allclose(got.coords['frequency'].values,
                             vectors['spectrum/frequency'], rtol=1e-12)
