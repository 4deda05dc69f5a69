"""Restatement of the reference's test fixtures (TEST INFRASTRUCTURE).

  schema.mock_truth_data                 schema.py:62-94
  schema.mock_forecast_data              schema.py:97-115
  schema.mock_hourly_climatology_data    schema.py:118-126
  utils.random_like                      utils.py:290-295
  test_utils.insert_nan                  test_utils.py:52-63
  metrics_test.get_random_truth_and_forecast   metrics_test.py:28-58

Datasets are `oracle.named.DS` (dict of named arrays + coords).
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from oracle.named import DS, NA

ALL_3D_VARIABLES = ('geopotential', 'temperature', 'u_component_of_wind',
                    'v_component_of_wind', 'specific_humidity')
ALL_2D_VARIABLES = ('2m_temperature',)
EARTH_RADIUS_M = 1000 * (6357 + 6378) / 2  # schema.py:59


def mock_truth_data(*, variables_3d=ALL_3D_VARIABLES,
                    variables_2d=ALL_2D_VARIABLES, levels=(500, 700, 850),
                    spatial_resolution_in_degrees=10.0,
                    time_start='2020-01-01', time_stop='2021-01-01',
                    time_resolution='1 day', dtype=np.float32) -> DS:
  num_latitudes = round(180 / spatial_resolution_in_degrees) + 1
  num_longitudes = round(360 / spatial_resolution_in_degrees)
  freq = pd.Timedelta(time_resolution)
  coords = {
      'time': pd.date_range(time_start, time_stop, freq=freq,
                            inclusive='left').values,
      'latitude': np.linspace(-90, 90, num_latitudes),
      'longitude': np.linspace(0, 360, num_longitudes, endpoint=False),
      'level': np.array(levels),
  }
  dims_3d = ('time', 'level', 'longitude', 'latitude')
  shape_3d = tuple(coords[d].size for d in dims_3d)
  data_vars = {k: NA(np.zeros(shape_3d, dtype), dims_3d) for k in variables_3d}
  if not data_vars:
    del coords['level']
  dims_2d = ('time', 'longitude', 'latitude')
  shape_2d = tuple(coords[d].size for d in dims_2d)
  for k in variables_2d:
    data_vars[k] = NA(np.zeros(shape_2d, dtype), dims_2d)
  return DS(data_vars, coords)


def mock_forecast_data(*, lead_start='0 day', lead_stop='10 day',
                       lead_resolution='1 day', ensemble_size=None,
                       **kwargs) -> DS:
  lead_time = pd.timedelta_range(pd.Timedelta(lead_start),
                                 pd.Timedelta(lead_stop),
                                 freq=pd.Timedelta(lead_resolution)).values
  ds = mock_truth_data(**kwargs)
  ds = ds.expand_dims('prediction_timedelta', coord=lead_time)
  if ensemble_size is not None:
    ds = ds.expand_dims('realization', size=ensemble_size)
  return ds


def mock_hourly_climatology_data(*, hour_interval=1, **kwargs) -> DS:
  hours = np.arange(0, 24, hour_interval)
  ds = mock_truth_data(**kwargs)
  ds = ds.isel(time=0)
  ds = ds.expand_dims('dayofyear', coord=1 + np.arange(366))
  ds = ds.expand_dims('hour', coord=hours)
  return ds


def random_like(dataset: DS, seed: int = 0) -> DS:
  rs = np.random.RandomState(seed)
  return dataset.copy(data={k: rs.normal(size=v.shape)
                            for k, v in dataset.items()})


def insert_nan(ds: DS, frac_nan: float = 0.1, seed=802701) -> DS:
  rng = np.random.RandomState(seed)
  out = {}
  for name, v in ds.items():
    mask = rng.rand(*v.shape) < frac_nan
    out[name] = np.where(mask, np.nan, v.data)
  return ds.copy(data=out)


def get_random_truth_and_forecast(variables=('geopotential',),
                                  ensemble_size=None, seed=802701,
                                  lead_start='0 day', lead_stop='10 day',
                                  **data_kwargs):
  data_kwargs_to_use = dict(
      variables_3d=variables, variables_2d=[], time_start='2019-12-01',
      time_stop='2019-12-02', spatial_resolution_in_degrees=30,
      time_resolution='3 hours')
  data_kwargs_to_use.update(data_kwargs)
  truth = random_like(mock_truth_data(**data_kwargs_to_use), seed=seed)
  forecast = random_like(
      mock_forecast_data(ensemble_size=ensemble_size, lead_start=lead_start,
                         lead_stop=lead_stop, **data_kwargs_to_use),
      seed=seed + 1)
  return truth, forecast
