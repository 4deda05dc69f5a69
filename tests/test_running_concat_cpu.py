"""evaluation.RunningConcat on the host: the sink of `temporal_mean=False`
(reference: config.py:55, evaluation.py:735-752 -- the chunk results are
written as they are, no `TemporalMean`)."""
import numpy as np
import pytest

from weatherbench2_amd import evaluation
from weatherbench2_amd import xarray_lite as xl


def _chunk(values, times, leads, extra=None):
  coords = {'init_time': np.asarray(times), 'lead_time': np.asarray(leads),
            'level': np.array([500, 850])}
  coords.update(extra or {})
  dims = ('metric', 'init_time', 'level', 'lead_time')
  coords['metric'] = ['mse', 'mae']
  return xl.Dataset({'z': xl.DataArray(values, dims, coords, 'z')}, coords)


def test_slices_land_under_their_labels_in_first_seen_order():
  rs = np.random.RandomState(0)
  times = np.array(['2020-01-01', '2020-01-02', '2020-01-03'],
                   dtype='datetime64[ns]')
  leads = np.array([6, 12], dtype='timedelta64[h]').astype('timedelta64[ns]')
  full = rs.standard_normal((2, 3, 2, 2)).astype(np.float32)
  sink = evaluation.RunningConcat('init_time', split_dim='lead_time')
  # lead-major order, the last init time of lead 12 h never comes
  for l in range(2):
    for i in range(3):
      if (i, l) == (2, 1):
        continue
      sink.add(_chunk(full[:, i:i + 1, :, l:l + 1], times[i:i + 1],
                      leads[l:l + 1]))
  out = sink.result()
  np.testing.assert_array_equal(out.coords['init_time'], times)
  np.testing.assert_array_equal(out.coords['lead_time'], leads)
  got = out['z'].values
  assert out['z'].dims == ('metric', 'init_time', 'level', 'lead_time')
  assert got.dtype == np.float32 and got.shape == full.shape
  want = full.copy()
  want[:, 2, :, 1] = np.nan
  np.testing.assert_array_equal(got, want)


def test_many_rows_and_chunks_of_several_steps():
  rs = np.random.RandomState(1)
  n_t = 150   # more than the initial 64 rows: the storage grows twice
  times = np.arange(n_t).astype('datetime64[D]').astype('datetime64[ns]')
  leads = np.array([0], dtype='timedelta64[ns]')
  full = rs.standard_normal((2, n_t, 2, 1))
  sink = evaluation.RunningConcat('init_time', split_dim='lead_time')
  for i in range(0, n_t, 7):
    sink.add(_chunk(full[:, i:i + 7], times[i:i + 7], leads))
  out = sink.result()
  np.testing.assert_array_equal(out['z'].values, full)
  np.testing.assert_array_equal(out.coords['init_time'], times)


def test_a_combination_that_comes_twice_is_refused():
  times = np.array(['2020-01-01'], dtype='datetime64[ns]')
  leads = np.array([0], dtype='timedelta64[ns]')
  sink = evaluation.RunningConcat('init_time', split_dim='lead_time')
  sink.add(_chunk(np.zeros((2, 1, 2, 1)), times, leads))
  with pytest.raises(ValueError, match='came twice'):
    sink.add(_chunk(np.zeros((2, 1, 2, 1)), times, leads))


def test_without_a_split_dim_and_with_a_layout_change():
  times = np.arange(4).astype('datetime64[D]').astype('datetime64[ns]')
  sink = evaluation.RunningConcat('time')
  coords = {'time': times[:2], 'level': np.array([1, 2, 3])}
  a = np.arange(6.0).reshape(2, 3)
  sink.add(xl.Dataset({'x': xl.DataArray(a, ('time', 'level'), coords)},
                      coords))
  coords2 = {'time': times[2:], 'level': np.array([1, 2, 3])}
  sink.add(xl.Dataset({'x': xl.DataArray(a + 10, ('time', 'level'), coords2)},
                      coords2))
  out = sink.result()
  np.testing.assert_array_equal(out['x'].values, np.concatenate([a, a + 10]))
  bad = {'time': times[:1], 'level': np.array([1, 2])}
  with pytest.raises(ValueError, match='layout changed'):
    sink.add(xl.Dataset({'x': xl.DataArray(np.zeros((1, 2)),
                                           ('time', 'level'), bad)}, bad))
  with pytest.raises(ValueError, match="no 'time' dim"):
    sink.add(xl.Dataset({'y': xl.DataArray(np.zeros(3), ('level',), coords)},
                        coords))
