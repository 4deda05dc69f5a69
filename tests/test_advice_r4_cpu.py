"""Regressions for the round-3 review findings that need no GPU."""
import numpy as np
import pytest

from weatherbench2_amd import engine, evaluation
from weatherbench2_amd import xarray_lite as xl


def test_digest_of_empty_arrays():
  a = engine.digest(np.zeros((0, 3), dtype=np.int64))
  b = engine.digest(np.zeros((0, 4), dtype=np.int64))
  assert isinstance(a, bytes) and a != b
  assert engine.digest(np.zeros((0,), dtype=np.float32)) != a


def test_datetime_labels_match_across_units():
  """truth.time in [ns] against valid times in [h]: `.sel` matches them
  (pandas compares instants), so does the label lookup."""
  have = (np.datetime64('2020-01-01T00', 'ns') +
          np.arange(8) * np.timedelta64(6, 'h'))
  want = (np.datetime64('2020-01-01T00', 'h') +
          np.array([[0, 6], [12, 42]]).astype('timedelta64[h]'))
  pos = evaluation._positions(have, want, 'time')
  np.testing.assert_array_equal(pos, [[0, 1], [2, 7]])
  with pytest.raises(KeyError):
    evaluation._positions(have, want + np.timedelta64(1, 'h'), 'time')
  lead_s = np.array([0, 21600], dtype='timedelta64[s]')
  lead_ns = lead_s.astype('timedelta64[ns]')
  assert xl.label_list(lead_s) == xl.label_list(lead_ns)


def test_by_init_persistence_keeps_every_truth_variable():
  """evaluation.py:651-675: truth.sel(time=init_time) keeps all variables,
  also those without a time dim (expanded along lead_time like the rest)."""
  rs = np.random.RandomState(0)
  time = (np.datetime64('2020-01-01T00', 'ns') +
          np.arange(6) * np.timedelta64(12, 'h'))
  lat, lon = np.linspace(-90, 90, 5), np.arange(0, 360, 60.0)
  truth = xl.Dataset(
      {'z': xl.DataArray(rs.normal(size=(6, 5, 6)),
                         ('time', 'latitude', 'longitude')),
       't2m': xl.DataArray(rs.normal(size=(6, 5, 6)),
                           ('time', 'latitude', 'longitude')),
       'orography': xl.DataArray(rs.normal(size=(5, 6)),
                                 ('latitude', 'longitude'))},
      {'time': time, 'latitude': lat, 'longitude': lon})
  lead = (np.arange(2) * np.timedelta64(12, 'h')).astype('timedelta64[ns]')
  init = time[1:3]
  fc = xl.Dataset(
      {'z': xl.DataArray(rs.normal(size=(2, 2, 5, 6)),
                         ('init_time', 'lead_time', 'latitude', 'longitude'))},
      {'init_time': init, 'lead_time': lead, 'latitude': lat, 'longitude': lon,
       'valid_time': xl.DataArray(init[:, None] + lead[None, :],
                                  ('init_time', 'lead_time'))})
  out, _ = evaluation._persistence_like_forecast_chunk(fc, None, truth, ['z'])
  assert set(out.keys()) == {'z', 't2m', 'orography'}
  assert out['z'].dims == ('lead_time', 'init_time', 'latitude', 'longitude')
  for name in ('z', 't2m'):
    want = np.broadcast_to(np.asarray(truth[name].data)[1:3][None],
                           (2, 2, 5, 6))
    np.testing.assert_array_equal(np.asarray(out[name].values), want)
  assert out['orography'].dims[0] == 'lead_time'
  np.testing.assert_array_equal(np.asarray(out['orography'].values)[1],
                                np.asarray(truth['orography'].data))


def test_merge_metrics_is_the_merge_of_expanded_datasets():
  """`merge_metrics` (one stack per variable) against the step-by-step
  `merge([ds.expand_dims(metric=[name])])` it replaces: metric labels sorted,
  dim order of the first dataset IN THE LIST that holds the variable, NaN
  where a metric lacks a variable."""
  import numpy as np
  from weatherbench2_amd import xarray_lite as xl
  rng = np.random.default_rng(0)
  coords = {'level': np.array([500, 850]),
            'prediction_timedelta': np.arange(3)}
  a = xl.Dataset(coords=coords)
  a.data_vars['z'] = xl.DataArray(rng.normal(size=(3, 2)),
                                  ('prediction_timedelta', 'level'), coords, 'z')
  a.data_vars['t'] = xl.DataArray(rng.normal(size=(3, 2)),
                                  ('prediction_timedelta', 'level'), coords, 't')
  b = xl.Dataset(coords=coords)
  b.data_vars['z'] = xl.DataArray(rng.normal(size=(2, 3)),
                                  ('level', 'prediction_timedelta'), coords, 'z')
  named = [('stddev', a), ('rmse', b)]  # list order != sorted label order
  got = xl.merge_metrics(named)
  want = xl.merge([d.expand_dims({'metric': [n]}) for n, d in named])
  assert list(got.coords['metric']) == list(want.coords['metric']) == [
      'rmse', 'stddev']
  for var in ('z', 't'):
    assert got[var].dims == want[var].dims, var
    np.testing.assert_array_equal(got[var].values, want[var].values)
  assert got['z'].dims == ('metric', 'prediction_timedelta', 'level')
  assert np.isnan(got['t'].values[0]).all()


def test_float32_copy_of_a_weight_field_only_when_it_loses_nothing():
  """plan._as_float32_field: K1 may read a 2-D weight field as float32 only if
  every value IS a float32 number (ERA5's land-sea mask, a 0/1 mask)."""
  import numpy as np
  import torch
  from weatherbench2_amd import plan as plan_lib
  up = lambda a, dtype: torch.as_tensor(np.ascontiguousarray(a), dtype=dtype)
  rs = np.random.RandomState(0)
  exact = rs.uniform(0, 1, (5, 7)).astype(np.float32).astype(np.float64)
  got = plan_lib._as_float32_field(exact, up)
  assert got is not None and got.dtype == torch.float32
  np.testing.assert_array_equal(got.numpy().astype(np.float64), exact)
  assert plan_lib._as_float32_field((exact > 0.5).astype(np.float64),
                                    up) is not None
  assert plan_lib._as_float32_field(rs.uniform(0, 1, (5, 7)), up) is None
  assert plan_lib._as_float32_field(None, up) is None


def test_bench_tables_never_read_a_slab_twice_per_launch():
  """bench.coprime: the unit strides of the bench's slab tables are coprime to
  the pool, so the 16 units of a launch are distinct for every input (a launch
  that reads a slab twice measures cache hits: profiles/r04_xcd_balance.md)."""
  import math
  import numpy as np
  import bench
  for pool in range(16, 97):
    for k in (1, 3, 5, 7):
      m = bench.coprime(k, pool)
      assert m >= k and math.gcd(m, pool) == 1
      for start in (0, 5, pool - 3):
        u = (start + np.arange(16)) % pool
        assert len(set(((u * m + 3) % pool).tolist())) == 16, (pool, k)
