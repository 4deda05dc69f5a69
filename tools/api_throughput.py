"""What a caller of the DROP-IN API gets (not the engine-level bench): one
device-resident chunk of `units` (init, lead) units of 13 x 721 x 1440 pushed
through `_metric_and_region_loop` (5 metrics x 13 regions, evaluation.py:388-438
signature) per call.  Prints ms per call and grid-point-evals/s, plus a profile
of where the host time goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from weatherbench2_amd import config, evaluation, metrics as gm, regions as R
from weatherbench2_amd import xarray_lite as xl
import bench

dev = torch.device('cuda', 0)
n_time, n_lead, n_lev = int(os.environ.get('N_TIME', 4)), 4, 13
lat = np.linspace(-90, 90, 721); lon = np.linspace(0, 360, 1440, endpoint=False)
times = np.datetime64('2020-01-01T00') + np.arange(n_time) * np.timedelta64(12, 'h')
leads = np.arange(n_lead) * np.timedelta64(6, 'h')
g = torch.Generator(device=dev).manual_seed(0)
f = torch.randn((n_time, n_lead, n_lev, 721, 1440), device=dev, generator=g)
t = torch.randn((n_time, n_lead, n_lev, 721, 1440), device=dev, generator=g)
dims = ('time', 'prediction_timedelta', 'level', 'latitude', 'longitude')
coords = {'time': times.astype('datetime64[ns]'),
          'prediction_timedelta': leads.astype('timedelta64[ns]'),
          'level': np.arange(n_lev), 'latitude': lat, 'longitude': lon}
forecast = xl.Dataset({'z': xl.DataArray(f, dims)}, coords)
truth = xl.Dataset({'z': xl.DataArray(t, dims)}, coords)
# climatology by (dayofyear, hour): small table, gathered through slab tables
clim = xl.Dataset(
    {'z': xl.DataArray(torch.randn((4, 3, n_lev, 721, 1440), device=dev,
                                   generator=g),
                       ('hour', 'dayofyear', 'level', 'latitude', 'longitude'))},
    {'hour': np.array([0, 6, 12, 18]), 'dayofyear': np.array([1, 2, 3]),
     'level': np.arange(n_lev), 'latitude': lat, 'longitude': lon})
regions = {k: v for k, v in bench.predefined_regions().items()}
cfg = config.Eval(metrics={'mse': gm.MSE(), 'acc': gm.ACC(climatology=clim),
                           'bias': gm.Bias(), 'mae': gm.MAE(),
                           'rmse': gm.RMSESqrtBeforeTimeAvg()},
                  regions=regions)
pts = n_time * n_lead * n_lev * 721 * 1440


def call():
  gm.clear_caches() if os.environ.get('CLEAR') else None
  return evaluation._metric_and_region_loop(forecast, truth, cfg, False,
                                            compute_chunk=True)


for i in range(3):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  out = call()
  torch.cuda.synchronize(); dt = time.perf_counter() - t0
  print(f'call {i}: {dt * 1e3:.2f} ms  {pts / dt / 1e9:.2f} G evals/s')
# steady state with fresh arrays each call (new identities => no result cache)
reps = 10
variants = []
for i in range(reps):
  fi = xl.Dataset({'z': xl.DataArray(f.roll(i + 1, 0), dims)}, coords)
  variants.append(fi)
torch.cuda.synchronize(); t0 = time.perf_counter()
for fi in variants:
  evaluation._metric_and_region_loop(fi, truth, cfg, False, compute_chunk=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f'steady: {dt * 1e3:.2f} ms per call, {pts / dt / 1e9:.2f} G evals/s '
      f'({n_time * n_lead} units)')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for fi in variants[:5]:
  gm.clear_caches()
  evaluation._metric_and_region_loop(fi, truth, cfg, False, compute_chunk=True)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
