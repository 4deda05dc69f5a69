"""Regenerates tests/golden/*.npz and reference_known_answers.json.

Provenance.  The reference (/root/reference, google-research/weatherbench2)
cannot be imported in the build image (xarray / apache_beam / absl are not
installable), so these vectors are NOT outputs of the reference
itself.  They are:

  reference_known_answers.json   the literal expected values of the reference's
                                 own tests, transcribed with file:line -- the
                                 numbers that pin the oracle
                                 (tests/test_oracle_golden.py, _thresholds.py);
  oracle_vectors_v1.npz          outputs of the NumPy oracle (oracle/) on seeded
                                 inputs (regenerated from the seeds by the
                                 tests), frozen here so that neither the oracle
                                 nor the HIP path can drift unnoticed.

Run from the repo root:   python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import fixtures  # noqa: E402
from oracle import metrics_np as om  # noqa: E402
from oracle import regions_np as oreg  # noqa: E402
from oracle import spectrum_np  # noqa: E402

KNOWN_ANSWERS = {
    'lat_weights': {
        'source': 'weatherbench2/metrics_test.py:63-82',
        'latitude': [-75, -45, -15, 15, 45, 75],
        'expected': [0.40192379, 1.09807621, 1.5, 1.5, 1.09807621, 0.40192379]},
    'wind_vector_rmse': {
        'source': 'weatherbench2/metrics_test.py:84-131',
        'expected_per_level': [0.0, 10.0, None]},
    'rmse_over_invalid_region': {
        'source': 'weatherbench2/metrics_test.py:133-152',
        'global': None, 'extra_tropics': 1.0},
    'gaussian_crps': {'source': 'weatherbench2/metrics_test.py:286-304',
                      'expected': 0.23385455},
    'gaussian_variance': {'source': 'weatherbench2/metrics_test.py:348-366',
                          'expected': 1.0},
    'gaussian_brier': {
        'source': 'weatherbench2/metrics_test.py:370-431',
        'error_0.02': {'gaussian_threshold': 0.04421,
                       'quantile_threshold': 0.257883},
        'error_1e6': {'gaussian_threshold': 0.70786,
                      'quantile_threshold': 0.707861}},
    'gaussian_ignorance': {'source': 'weatherbench2/metrics_test.py:436-475',
                           'error_0.02': 0.236055, 'error_1e6': 1.841019},
    'gaussian_rps': {'source': 'weatherbench2/metrics_test.py:480-535',
                     'error_0.02': 0.295746, 'error_1e6': 0.758203},
    'ensemble_brier': {'source': 'weatherbench2/metrics_test.py:989-1029',
                       'cases': [[0.0, 0.1, 0.0], [0.0, 1.0, 0.25],
                                 [-10.0, 0.1, 1.0]]},
    'ensemble_ignorance': {'source': 'weatherbench2/metrics_test.py:1294-1329',
                           'cases': [[0.0, 0.0], [-10.0, 'inf']]},
    'ensemble_rps': {'source': 'weatherbench2/metrics_test.py:1334-1388',
                     'cases': [[0.02, 0.0], [-2.0, 2.0]]},
    'seeps': {'source': 'weatherbench2/metrics_test.py:1393-1436',
              'perfect': 0.0, 'plus_half': 1.25},
    'central_reliability': {
        'source': 'weatherbench2/metrics_test.py:700-779',
        'cases': [
            {'hist': [0.2, 0.1, 0.7], 'prob': [0.1, 1.0],
             'desired': [1 / 3, 1.0]},
            {'hist': [0.2, 0.0, 0.1, 0.1, 0.6], 'prob': [0.1, 0.2, 1.0],
             'desired': [0.2, 0.6, 1.0]},
            {'hist': [0.1, 0.1, 0.5, 0.3], 'prob': [0.6, 1.0],
             'desired': [0.5, 1.0]},
            {'hist': [0.1, 0.1, 0.3, 0.2, 0.0, 0.3], 'prob': [0.5, 0.6, 1.0],
             'desired': [1 / 3, 2 / 3, 1.0]}]},
}


def regions():
  return {
      'global': oreg.SliceRegion(),
      'tropics': oreg.SliceRegion(lat_slice=slice(-20, 20)),
      'extra_tropics': oreg.ExtraTropicalRegion(),
      'europe': oreg.SliceRegion(
          lat_slice=slice(35, 75),
          lon_slice=[slice(360 - 12.5, None), slice(0, 42.5)]),
  }


def deterministic_case():
  """float32 random truth/forecast/climatology on the 10-degree mock grid."""
  truth, forecast = fixtures.get_random_truth_and_forecast(
      spatial_resolution_in_degrees=10, seed=802701)
  clim = fixtures.random_like(fixtures.mock_hourly_climatology_data(
      hour_interval=3, variables_3d=['geopotential'], variables_2d=[],
      spatial_resolution_in_degrees=10), seed=5)
  cast = lambda ds: ds.copy(data={k: v.data.astype(np.float32)
                                  for k, v in ds.items()})
  return cast(truth), cast(forecast), cast(clim)


def ensemble_case():
  truth, forecast = fixtures.get_random_truth_and_forecast(
      ensemble_size=5, spatial_resolution_in_degrees=10, lead_stop='2 day',
      seed=802701)
  cast = lambda ds: ds.copy(data={k: v.data.astype(np.float32)
                                  for k, v in ds.items()})
  return cast(truth), cast(forecast)


def spectrum_case():
  rs = np.random.RandomState(7)
  lat = np.arange(-60.0, 61.0, 5.0)
  lon = np.linspace(0, 360, 72, endpoint=False)
  x = rs.standard_normal((3, len(lat), len(lon)))
  return x, lat, lon


DET = ('MSE', 'RMSESqrtBeforeTimeAvg', 'MAE', 'Bias', 'ACC')
ENS = ('CRPS', 'CRPSSpread', 'CRPSSkill', 'EnsembleMeanMSE',
       'EnsembleMeanRMSESqrtBeforeTimeAvg', 'EnsembleVariance',
       'EnsembleStddevSqrtBeforeTimeAvg', 'DebiasedEnsembleMeanMSE')


def oracle_vectors() -> dict:
  out = {}
  truth, forecast, clim = deterministic_case()
  for rname, region in regions().items():
    for mname in DET:
      metric = om.ACC(clim) if mname == 'ACC' else getattr(om, mname)()
      res = metric.compute_chunk(forecast, truth, region=region)
      v = res['geopotential']
      out[f'det/{mname}/{rname}'] = v.data
      out[f'det/{mname}/{rname}/dims'] = np.array(v.dims)
  truth, forecast = ensemble_case()
  for rname in ('global', 'europe'):
    for mname in ENS:
      res = getattr(om, mname)().compute_chunk(forecast, truth,
                                               region=regions()[rname])
      v = res['geopotential']
      out[f'ens/{mname}/{rname}'] = v.data
      out[f'ens/{mname}/{rname}/dims'] = np.array(v.dims)
  x, lat, lon = spectrum_case()
  spec, freq, _ = spectrum_np.zonal_energy_spectrum(x, lat, lon, lat_axis=1,
                                                    lon_axis=2)
  out['spectrum/values'] = spec
  out['spectrum/frequency'] = freq
  return out


def main():
  with open(os.path.join(HERE, 'reference_known_answers.json'), 'w') as f:
    json.dump(KNOWN_ANSWERS, f, indent=1)
  np.savez_compressed(os.path.join(HERE, 'oracle_vectors_v1.npz'),
                      **oracle_vectors())
  print('wrote', os.listdir(HERE))


if __name__ == '__main__':
  main()
