"""How representative is the `cpu_baseline` port?  Times the NumPy oracle
(bench.py's cpu_baseline leg, kind "port") and the REFERENCE's own metric code
(on the mini-xarray of oracle/refshim -- NOT real xarray) on the same full-size
13 x 721 x 1440 float32 unit: 5 metrics x 13 predefined regions, one (metric,
region) at a time like evaluation.py:408-435, one process.  Build container
only (needs /root/reference); the GPU box's baseline stays the port.

  python tools/cpu_reference_vs_port.py [n_pairs]
"""
import os
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
sys.path.insert(0, os.path.join(ROOT, 'oracle', 'refshim'))

import numpy as np  # noqa: E402
import xarray as xr  # noqa: E402  (the stand-in)
from weatherbench2 import metrics as rm, regions as rr  # noqa: E402

from oracle import cpu_baseline as cb  # noqa: E402
from oracle import metrics_np as om  # noqa: E402
from oracle.named import DS, NA  # noqa: E402


def main():
  n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 65
  rs = np.random.RandomState(0)
  lat = np.linspace(-90, 90, cb.N_LAT)
  lon = np.linspace(0, 360, cb.N_LON, endpoint=False)
  dims = ('time', 'level', 'latitude', 'longitude')
  coords = {'time': np.array(['2020-01-01T00'], dtype='datetime64[ns]'),
            'level': np.arange(cb.N_LEV), 'latitude': lat, 'longitude': lon}
  mk = lambda: rs.standard_normal((1, cb.N_LEV, cb.N_LAT, cb.N_LON)).astype(
      np.float32)
  fa, ta, ca = mk(), mk(), mk()
  cdims = ('dayofyear',) + dims[1:]
  ccoords = {'dayofyear': np.array([1]), 'level': coords['level'],
             'latitude': lat, 'longitude': lon}
  # the port
  f, t = DS({'z': NA(fa, dims)}, coords), DS({'z': NA(ta, dims)}, coords)
  clim = DS({'z': NA(ca, cdims)}, ccoords)
  ometrics = [om.MSE(), om.RMSESqrtBeforeTimeAvg(), om.MAE(), om.Bias(),
              om.ACC(clim)]
  oregions = list(cb.predefined_regions().values())
  # the reference
  xf = xr.Dataset({'z': (dims, fa)}, coords)
  xt = xr.Dataset({'z': (dims, ta)}, coords)
  xc = xr.Dataset({'z': (cdims, ca)}, ccoords)
  rmetrics = [rm.MSE(), rm.RMSESqrtBeforeTimeAvg(), rm.MAE(), rm.Bias(),
              rm.ACC(climatology=xc)]
  rregions = [rr.SliceRegion(lat_slice=r.lat_slice, lon_slice=r.lon_slice)
              for r in oregions]
  pts = cb.N_LEV * cb.N_LAT * cb.N_LON
  for label, metrics, regions, a, b in (('port (oracle)', ometrics, oregions, f,
                                         t),
                                        ('reference on the mini-xarray',
                                         rmetrics, rregions, xf, xt)):
    pairs = [(m, r) for r in regions for m in metrics][:n_pairs]
    t0 = time.perf_counter()
    acc = 0.0
    for m, r in pairs:
      res = m.compute_chunk(a, b, region=r)['z']
      acc += float(np.nansum(np.asarray(res.data)))
    dt = time.perf_counter() - t0
    units = len(pairs) / 65.0
    print(f'{label:32s} {len(pairs)} (metric, region) evaluations in {dt:6.1f} s'
          f' = {units * pts / dt / 1e6:6.2f} M grid-point-evals/s  (checksum '
          f'{acc:.6f})')


if __name__ == '__main__':
  main()
