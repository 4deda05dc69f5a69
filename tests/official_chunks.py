"""A small by-init dataset in the shape of the official 0.25-degree evaluation
(docs/source/official-evaluation.md:537-556: 3-D and surface variables, the u/v
pairs of the wind-vector metrics, `input_chunks=init_time=1,lead_time=1`), as
oracle containers and as product Datasets.  Shared by the CPU and GPU tests of
the chunk batching."""
import numpy as np

from oracle import evaluation_np as oe
from oracle import metrics_np as om
from oracle import regions_np as oreg
from oracle.named import DS, NA

VARS_3D = ('geopotential', 'u_component_of_wind', 'v_component_of_wind')
VARS_2D = ('2m_temperature', '10m_u_component_of_wind',
           '10m_v_component_of_wind')
WIND = (('u_component_of_wind', 'v_component_of_wind', 'wind_vector'),
        ('10m_u_component_of_wind', '10m_v_component_of_wind',
         '10m_wind_vector'))


def make(n_init=6, n_lead=3, n_level=2, n_lat=19, n_lon=36, seed=0,
         dtype=np.float32, nan_frac=0.0):
  """(forecast, truth at valid time, climatology) as oracle DS."""
  rs = np.random.RandomState(seed)
  lat = np.linspace(-90, 90, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  level = np.array([500, 850, 700, 300][:n_level])
  init = (np.datetime64('2020-01-01T00', 'ns') +
          np.arange(n_init) * np.timedelta64(12, 'h'))
  lead = (np.arange(n_lead) * np.timedelta64(6, 'h')).astype('timedelta64[ns]')
  valid = init[:, None] + lead[None, :]
  n_time = 2 * n_init + n_lead
  time = (np.datetime64('2020-01-01T00', 'ns') +
          np.arange(n_time) * np.timedelta64(6, 'h'))

  def field(shape):
    x = rs.normal(size=shape).astype(dtype)
    if nan_frac:
      x[rs.rand(*shape) < nan_frac] = np.nan
    return x
  fcoords = {'init_time': init, 'lead_time': lead, 'level': level,
             'latitude': lat, 'longitude': lon,
             'valid_time': NA(valid, ('init_time', 'lead_time'))}
  d3 = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
  d2 = ('init_time', 'lead_time', 'latitude', 'longitude')
  fvars = {k: NA(field((n_init, n_lead, n_level, n_lat, n_lon)), d3)
           for k in VARS_3D}
  fvars.update({k: NA(field((n_init, n_lead, n_lat, n_lon)), d2)
                for k in VARS_2D})
  forecast = DS(fvars, fcoords)
  tcoords = {'time': time, 'level': level, 'latitude': lat, 'longitude': lon}
  tvars = {k: NA(field((n_time, n_level, n_lat, n_lon)),
                 ('time', 'level', 'latitude', 'longitude')) for k in VARS_3D}
  tvars.update({k: NA(field((n_time, n_lat, n_lon)),
                      ('time', 'latitude', 'longitude')) for k in VARS_2D})
  truth = oe.truth_at_valid_time(DS(tvars, tcoords), forecast)
  hours = np.array([0, 6, 12, 18])
  days = 1 + np.arange(8)
  ccoords = {'hour': hours, 'dayofyear': days, 'level': level,
             'latitude': lat, 'longitude': lon}
  cvars = {k: NA(rs.normal(size=(4, 8, n_level, n_lat, n_lon)).astype(dtype),
                 ('hour', 'dayofyear', 'level', 'latitude', 'longitude'))
           for k in VARS_3D}
  cvars.update({k: NA(rs.normal(size=(4, 8, n_lat, n_lon)).astype(dtype),
                      ('hour', 'dayofyear', 'latitude', 'longitude'))
                for k in VARS_2D})
  return forecast, truth, DS(cvars, ccoords)


def land_sea_mask(n_lat=19, n_lon=36, seed=3):
  rs = np.random.RandomState(seed)
  return np.clip(rs.uniform(-0.5, 1.2, size=(n_lat, n_lon)), 0.0, 1.0)


def oracle_regions(lat, lon, lsm):
  land = oreg.LandRegion(NA(lsm, ('latitude', 'longitude')), lat, lon)
  return {
      'global': oreg.SliceRegion(),
      'tropics': oreg.SliceRegion(lat_slice=slice(-20, 20)),
      'europe': oreg.SliceRegion(
          lat_slice=slice(35, 75),
          lon_slice=[slice(360 - 12.5, None), slice(0, 42.5)]),
      'global_land': land,
      'tropics_land': oreg.CombinedRegion(
          [oreg.SliceRegion(lat_slice=slice(-20, 20)), land]),
  }


def oracle_metrics(climatology):
  wv = [om.WindVectorMSE(u_name=u, v_name=v, vector_name=n)
        for u, v, n in WIND]
  return {'mse': om.MSE(wind_vector_mse=wv), 'acc': om.ACC(climatology),
          'bias': om.Bias(), 'mae': om.MAE()}


def product_metrics(gm, climatology):
  wv = [gm.WindVectorMSE(u_name=u, v_name=v, vector_name=n)
        for u, v, n in WIND]
  return {'mse': gm.MSE(wind_vector_mse=wv),
          'acc': gm.ACC(climatology=climatology), 'bias': gm.Bias(),
          'mae': gm.MAE()}


def chunk_pairs(forecast, truth, order='init'):
  """(forecast, truth) product chunks of one init time x one lead time, in
  init-major or lead-major order."""
  n_i, n_l = forecast.sizes['init_time'], forecast.sizes['lead_time']
  keys = [(i, l) for i in range(n_i) for l in range(n_l)]
  if order == 'lead':
    keys = [(i, l) for l in range(n_l) for i in range(n_i)]
  return [(forecast.isel(init_time=slice(i, i + 1), lead_time=slice(l, l + 1)),
           truth.isel(init_time=slice(i, i + 1), lead_time=slice(l, l + 1)))
          for i, l in keys]


def expected_time_mean(forecast, truth, climatology, regions, skipna):
  """{(metric, region): {var: array over (lead_time[, level])}}: the oracle's
  per-chunk values averaged over init_time (xbeam.Mean semantics)."""
  per_chunk = oe.metric_and_region_loop(
      forecast, truth, oracle_metrics(climatology), regions, skipna,
      compute_chunk=True)
  out = {}
  for key, ds in per_chunk.items():
    out[key] = {}
    for name, var in ds.items():
      ax = var.dims.index('init_time')
      data = np.asarray(var.data, dtype=np.float64)
      with np.errstate(all='ignore'):
        mean = (np.nanmean(data, axis=ax) if skipna else data.mean(axis=ax))
      out[key][name] = (tuple(d for d in var.dims if d != 'init_time'), mean)
  return out
