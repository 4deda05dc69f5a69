"""CPU baseline leg of bench.py: times the NumPy oracle on host cores.

Test/measurement infrastructure only (see oracle/__init__.py).  The workload
is BASELINE configs[1] evaluated the way the reference does it
(evaluation.py:408-435): one (metric, region) at a time from the raw arrays,
5 metrics x 13 predefined regions per 13-level unit.

  python -m oracle.cpu_baseline --seconds 10     # one process, prints JSON

bench.py starts several of these at once (one per core it wants to load) and
adds the rates up; the module deliberately imports neither torch nor the
product package.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

N_LAT, N_LON, N_LEV = 721, 1440, 13


def predefined_regions():
  """scripts/evaluate.py:345-374 (the 13 slice regions), oracle classes."""
  from oracle import regions_np
  R = regions_np.SliceRegion
  return {
      'global': R(),
      'tropics': R(lat_slice=slice(-20, 20)),
      'extra-tropics': R(lat_slice=[slice(None, -20), slice(20, None)]),
      'northern-hemisphere': R(lat_slice=slice(20, None)),
      'southern-hemisphere': R(lat_slice=slice(None, -20)),
      'europe': R(lat_slice=slice(35, 75),
                  lon_slice=[slice(360 - 12.5, None), slice(0, 42.5)]),
      'north-america': R(lat_slice=slice(25, 60),
                         lon_slice=slice(360 - 120, 360 - 75)),
      'north-atlantic': R(lat_slice=slice(25, 65),
                          lon_slice=slice(360 - 70, 360 - 10)),
      'north-pacific': R(lat_slice=slice(25, 60),
                         lon_slice=slice(145, 360 - 130)),
      'east-asia': R(lat_slice=slice(25, 60), lon_slice=slice(102.5, 150)),
      'ausnz': R(lat_slice=slice(-45, -12.5), lon_slice=slice(120, 175)),
      'arctic': R(lat_slice=slice(60, 90)),
      'antarctic': R(lat_slice=slice(-90, -60)),
  }


def run(seconds: float, seed: int = 0) -> dict:
  """Evaluates whole units until `seconds` have passed; returns the counts."""
  from oracle import metrics_np as om
  from oracle.named import DS, NA
  rs = np.random.RandomState(seed)
  lat = np.linspace(-90, 90, N_LAT)
  lon = np.linspace(0, 360, N_LON, endpoint=False)
  dims = ('time', 'level', 'latitude', 'longitude')
  coords = {'time': np.array(['2020-01-01T00'], dtype='datetime64[ns]'),
            'level': np.arange(N_LEV), 'latitude': lat, 'longitude': lon}
  mk = lambda: rs.standard_normal((1, N_LEV, N_LAT, N_LON)).astype(np.float32)
  f, t = DS({'z': NA(mk(), dims)}, coords), DS({'z': NA(mk(), dims)}, coords)
  clim = DS({'z': NA(mk(), ('dayofyear',) + dims[1:])},
            {'dayofyear': np.array([1]), 'level': coords['level'],
             'latitude': lat, 'longitude': lon})
  metrics = {'mse': om.MSE(), 'rmse': om.RMSESqrtBeforeTimeAvg(),
             'mae': om.MAE(), 'bias': om.Bias(), 'acc': om.ACC(clim)}
  regions = predefined_regions()
  # The clock is checked after every (metric, region) evaluation, so a run ends
  # within one evaluation of `seconds` however loaded the host is (a whole unit
  # is 65 evaluations and takes minutes when 256 processes share the memory
  # bus); a partly evaluated unit counts by the fraction of its 65 evaluations.
  pairs = [(m, r) for r in regions.values() for m in metrics.values()]
  t0 = time.perf_counter()
  done = 0
  dt = 0.0
  while dt < seconds:
    m, region = pairs[done % len(pairs)]
    m.compute_chunk(f, t, region=region)
    done += 1
    dt = time.perf_counter() - t0
  units = done / len(pairs)
  return {'units': units, 'seconds': dt,
          'points': units * N_LEV * N_LAT * N_LON,
          'metrics': len(metrics), 'regions': len(regions)}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--seconds', type=float, default=10.0)
  ap.add_argument('--seed', type=int, default=0)
  args = ap.parse_args()
  json.dump(run(args.seconds, args.seed), sys.stdout)
  print()


if __name__ == '__main__':
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  main()
