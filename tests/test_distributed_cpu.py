"""world_size-2 gloo test of the sharded temporal mean (no GPU needed).

The per-chunk results are synthetic Datasets: what is under test is the
sharding + (sum, count) all-reduce logic that on the GPU box runs over RCCL.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from weatherbench2_amd import evaluation
from weatherbench2_amd import xarray_lite as xl


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def _chunk_results(n_chunks, seed=0):
  rs = np.random.RandomState(seed)
  out = []
  for _ in range(n_chunks):
    a = rs.standard_normal((2, 3, 2, 4))  # (metric, region, init_time, lead)
    a[rs.rand(*a.shape) < 0.1] = np.nan
    out.append(xl.Dataset(
        {'z': xl.DataArray(a, ('metric', 'region', 'init_time', 'lead_time'))},
        {'metric': np.array(['a', 'b'], dtype=object),
         'region': np.array(['r0', 'r1', 'r2'], dtype=object)}))
  return out


def _worker(rank, world, port, n_chunks, skipna, queue):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    chunks = _chunk_results(n_chunks)
    lo, hi = evaluation.shard_bounds(n_chunks, world, rank)
    mean = evaluation.RunningMean('init_time', skipna, device='cpu')
    for i in range(lo, hi):
      mean.add(chunks[i])
    result = mean.result()
    queue.put((rank, lo, hi, result['z'].values))
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('skipna', [False, True])
def test_two_rank_running_mean_matches_single_process(skipna):
  world, n_chunks = 2, 5
  ctx = mp.get_context('spawn')
  queue = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker,
                       args=(r, world, port, n_chunks, skipna, queue))
           for r in range(world)]
  for p in procs:
    p.start()
  got = [queue.get(timeout=120) for _ in procs]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  full = np.concatenate([c['z'].values for c in _chunk_results(n_chunks)],
                        axis=2)
  with np.errstate(all='ignore'):
    import warnings
    with warnings.catch_warnings():
      warnings.simplefilter('ignore')
      want = (np.nanmean if skipna else np.mean)(full, axis=2)
  shards = sorted((lo, hi) for _, lo, hi, _ in got)
  assert shards == [(0, 3), (3, 5)]
  for _, _, _, values in got:  # every rank holds the global mean
    np.testing.assert_allclose(values, want, rtol=1e-13, equal_nan=True)


def _lead_chunks(n_init=4, n_lead=3, seed=3):
  """(init_time=1, lead_time=1) chunk results in LEAD-major order."""
  rs = np.random.RandomState(seed)
  leads = (np.arange(n_lead) * np.timedelta64(6, 'h')).astype('timedelta64[ns]')
  values = rs.standard_normal((n_init, n_lead, 2))
  out = []
  for l in range(n_lead):
    for i in range(n_init):
      out.append(xl.Dataset(
          {'z': xl.DataArray(values[i:i + 1, l:l + 1],
                             ('init_time', 'lead_time', 'level'))},
          {'init_time': np.arange(i, i + 1), 'lead_time': leads[l:l + 1],
           'level': np.array([500, 850])}))
  return out, values, leads


def _lead_worker(rank, world, port, queue):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    chunks, _, _ = _lead_chunks()
    lo, hi = evaluation.shard_bounds(len(chunks), world, rank)
    mean = evaluation.RunningMean('init_time', False, device='cpu',
                                  split_dim='lead_time')
    for i in range(lo, hi):
      mean.add(chunks[i])
    res = mean.result()
    queue.put((rank, res['z'].values, np.asarray(res.coords['lead_time'])))
  finally:
    dist.destroy_process_group()


def test_two_ranks_with_different_lead_labels_agree_on_the_layout():
  """Lead-major chunk lists (the official chunking splits the lead dim too):
  rank 0's shard holds leads 0 and 1, rank 1's leads 1 and 2 -- the ranks
  exchange their label sets before the all-reduce, rows they never met count
  as zeros, and both end with the global per-lead mean."""
  world = 2
  ctx = mp.get_context('spawn')
  queue = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_lead_worker, args=(r, world, port, queue))
           for r in range(world)]
  for p in procs:
    p.start()
  got = [queue.get(timeout=120) for _ in procs]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  _, values, leads = _lead_chunks()
  for _, z, lead_labels in got:
    np.testing.assert_array_equal(lead_labels, leads)
    np.testing.assert_allclose(z, values.mean(0), rtol=1e-14)


def _concat_worker(rank, world, port, queue):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    chunks, _, _ = _lead_chunks()
    lo, hi = evaluation.shard_bounds(len(chunks), world, rank)
    sink = evaluation.RunningConcat('init_time', device='cpu',
                                    split_dim='lead_time')
    for i in range(lo, hi):
      sink.add(chunks[i])
    res = sink.result()
    queue.put((rank, res['z'].values, np.asarray(res.coords['lead_time']),
               np.asarray(res.coords['init_time']), res['z'].dims))
  finally:
    dist.destroy_process_group()


def test_two_ranks_keep_every_chunk_without_the_temporal_mean():
  """`temporal_mean=False` over two ranks: each rank files its shard's chunks,
  result() exchanges the kept slices and every rank ends with all of them under
  the same labels (rank order = list order)."""
  world = 2
  ctx = mp.get_context('spawn')
  queue = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_concat_worker, args=(r, world, port, queue))
           for r in range(world)]
  for p in procs:
    p.start()
  got = [queue.get(timeout=120) for _ in procs]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  _, values, leads = _lead_chunks()
  for _, z, lead_labels, init_labels, dims in got:
    assert dims == ('init_time', 'lead_time', 'level')
    np.testing.assert_array_equal(lead_labels, leads)
    np.testing.assert_array_equal(init_labels, np.arange(values.shape[0]))
    np.testing.assert_array_equal(z, values)


def test_shard_bounds_cover_everything():
  for n in (1, 7, 8, 2920):
    for world in (1, 2, 3, 8):
      spans = [evaluation.shard_bounds(n, world, r) for r in range(world)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
      sizes = [hi - lo for lo, hi in spans]
      assert max(sizes) - min(sizes) <= 1


def test_select_truth_at_valid_time_matches_label_lookup():
  """evaluation.py:474-475 on host arrays (the device path is the same code
  with torch.index_select; exercised in tests/test_eval_gpu.py)."""
  import numpy as np
  import pytest
  from weatherbench2_amd import evaluation, xarray_lite as xl
  t0 = np.datetime64('2020-01-01T00', 'ns')
  time = t0 + np.arange(10) * np.timedelta64(12, 'h')
  init = time[[0, 2, 3]]
  lead = np.arange(3) * np.timedelta64(24, 'h')
  rs = np.random.RandomState(0)
  truth = xl.Dataset(
      {'z': xl.DataArray(rs.rand(2, 10, 4, 5), ('level', 'time', 'latitude',
                                                'longitude')),
       'orog': xl.DataArray(rs.rand(4, 5), ('latitude', 'longitude'))},
      {'level': np.array([500, 850]), 'time': time,
       'latitude': np.linspace(-45, 45, 4), 'longitude': np.arange(5) * 72.0})
  forecast = xl.Dataset(
      {'z': xl.DataArray(rs.rand(3, 3, 2, 4, 5),
                         ('init_time', 'prediction_timedelta', 'level',
                          'latitude', 'longitude'))},
      {'init_time': init, 'prediction_timedelta': lead,
       'level': np.array([500, 850]), 'latitude': truth.coords['latitude'],
       'longitude': truth.coords['longitude']})
  got = evaluation.select_truth_at_valid_time(truth, forecast)
  assert got['z'].dims == ('level', 'init_time', 'prediction_timedelta',
                           'latitude', 'longitude')
  # like xarray's vectorised .sel: `time` stays as a coordinate over the
  # indexer's dims (thresholds.py:140 reads the day of year from it)
  assert got.coords['time'].dims == ('init_time', 'prediction_timedelta')
  np.testing.assert_array_equal(got.coords['time'].values,
                                got.coords['valid_time'].values)
  assert got['orog'].dims == ('latitude', 'longitude')
  for i in range(3):
    for l in range(3):
      k = int(np.where(time == init[i] + lead[l])[0][0])
      np.testing.assert_array_equal(got['z'].values[:, i, l],
                                    truth['z'].values[:, k])
  np.testing.assert_array_equal(got.coords['valid_time'].values,
                                init[:, None] + lead[None, :])
  late = xl.Dataset(dict(forecast.data_vars),
                    {**forecast.coords, 'init_time': init + np.timedelta64(9, 'D')})
  with pytest.raises(KeyError):
    evaluation.select_truth_at_valid_time(truth, late)


def test_make_latitude_increasing():
  # evaluation.py:41-47
  import numpy as np
  from weatherbench2_amd import evaluation, xarray_lite as xl
  lat = np.array([60.0, 20.0, -20.0, -60.0])
  x = np.arange(2 * 4 * 3, dtype=np.float32).reshape(2, 4, 3)
  ds = xl.Dataset({'z': xl.DataArray(x, ('time', 'latitude', 'longitude')),
                   's': xl.DataArray(np.ones(2), ('time',))},
                  {'time': np.arange(2), 'latitude': lat,
                   'longitude': np.arange(3) * 120.0})
  out = evaluation.make_latitude_increasing(ds)
  np.testing.assert_array_equal(out.coords['latitude'], lat[::-1])
  np.testing.assert_array_equal(out['z'].values, x[:, ::-1])
  np.testing.assert_array_equal(out['s'].values, np.ones(2))
  assert evaluation.make_latitude_increasing(out) is out  # already increasing


def test_strided_views_become_slab_tables_not_copies():
  """metrics._physical_slabs: overlapping windows / expanded broadcasts keep
  their storage and get a physical slab table (host logic, no GPU needed)."""
  import numpy as np
  import torch
  from weatherbench2_amd import evaluation, metrics as gm
  n_row, n_col = 3, 4
  base = torch.arange(10 * 2 * n_row * n_col, dtype=torch.float32).reshape(
      10, 2, n_row, n_col)                      # (time, level, lat, lon)
  index = np.array([[0, 2, 4], [3, 5, 7]])      # a=0, b=3, c=2
  view = evaluation._affine_time_view(base, 0, index)
  assert view is not None and view.shape == (2, 3, 2, n_row, n_col)
  assert view.untyped_storage().data_ptr() == base.untyped_storage().data_ptr()
  for i in range(2):
    for l in range(3):
      assert torch.equal(view[i, l], base[index[i, l]])
  x, table = gm._physical_slabs(view, None, n_row, n_col)
  assert x is view                                   # handed over as it is
  want = (index[:, :, None] * 2 + np.arange(2)[None, None, :]).ravel()
  np.testing.assert_array_equal(table, want)
  # a logical table (e.g. a broadcast over another dim) composes with it
  logical = np.array([5, 0, 11], dtype=np.int64)
  _, composed = gm._physical_slabs(view, logical, n_row, n_col)
  np.testing.assert_array_equal(composed, want[logical])
  # expanded broadcast: stride 0
  e = base[:1].expand(4, 2, n_row, n_col)
  x, table = gm._physical_slabs(e, None, n_row, n_col)
  assert x is e
  np.testing.assert_array_equal(table, np.tile(np.arange(2), 4))
  # slabs that are not intact (transposed) are copied
  t = base.transpose(-1, -2)
  x, table = gm._physical_slabs(t, None, n_col, n_row)
  assert x.is_contiguous() and table is None
  # irregular index: no affine view
  assert evaluation._affine_time_view(base, 0, np.array([[0, 2, 5]])) is None


def test_public_get_lat_weights_matches_the_reference_known_answer():
  # metrics_test.py:63-82 through the product's public function
  import numpy as np
  from weatherbench2_amd import metrics as gm, xarray_lite as xl
  lat = np.array([-75.0, -45.0, -15.0, 15.0, 45.0, 75.0])
  ds = xl.Dataset({'z': xl.DataArray(np.zeros((6, 2)), ('latitude',
                                                         'longitude'))},
                  {'latitude': lat, 'longitude': np.array([0.0, 180.0])})
  w = gm.get_lat_weights(ds)
  assert w.dims == ('latitude',)
  expected = 3 * np.array([1 - np.sqrt(3) / 2, (np.sqrt(3) - 1) / 2, 0.5, 0.5,
                           (np.sqrt(3) - 1) / 2, 1 - np.sqrt(3) / 2])
  np.testing.assert_allclose(w.values, expected, rtol=1e-12)
  np.testing.assert_allclose(w.values.mean(), 1.0)


def test_chunk_prefetch_keeps_order_overlaps_and_surfaces_errors():
  """evaluation._prefetched: chunks[i] (the lazy sequence's IO) runs ahead on
  one background thread, items come out in order, at most `depth` + 1 fetches
  are outstanding, a failing fetch raises at its own position."""
  import threading
  import time
  from weatherbench2_amd import evaluation

  class Lazy:
    def __init__(self, n, fail_at=None):
      self.n, self.fail_at = n, fail_at
      self.log, self.threads = [], set()

    def __len__(self):
      return self.n

    def __getitem__(self, i):
      self.threads.add(threading.current_thread().name)
      self.log.append(('start', i))
      time.sleep(0.02)
      if i == self.fail_at:
        raise OSError(f'chunk {i} unreadable')
      self.log.append(('done', i))
      return ('forecast', i), ('truth', i)

  lazy = Lazy(8)
  seen, consumed_at = [], {}
  for f, t_ in evaluation._prefetched(lazy, 1, 7, 2):
    assert f[1] == t_[1]
    seen.append(f[1])
    consumed_at[f[1]] = len(lazy.log)
    time.sleep(0.03)  # "GPU work": fetches of later chunks proceed meanwhile
  assert seen == [1, 2, 3, 4, 5, 6]
  assert [i for kind, i in lazy.log if kind == 'start'] == [1, 2, 3, 4, 5, 6]
  assert all(n.startswith('wb2hip-prefetch') for n in lazy.threads)
  # overlap: when chunk 2 was handed out, the fetch of chunk 3 had started
  started_before = {i for kind, i in lazy.log[:consumed_at[2]] if kind == 'start'}
  assert 3 in started_before
  # bounded: never more than depth + 1 = 3 fetches beyond the consumed chunk
  for i in seen:
    ahead = {j for kind, j in lazy.log[:consumed_at[i]] if kind == 'start'}
    assert max(ahead) <= i + 3
  # depth 0: plain in-thread iteration
  lazy0 = Lazy(3)
  assert [f[1] for f, _ in evaluation._prefetched(lazy0, 0, 3, 0)] == [0, 1, 2]
  assert lazy0.threads == {threading.current_thread().name}
  # an IO error surfaces at its chunk, after the earlier ones were delivered
  bad = Lazy(6, fail_at=3)
  got = []
  with pytest.raises(OSError, match='chunk 3'):
    for f, _ in evaluation._prefetched(bad, 0, 6, 2):
      got.append(f[1])
  assert got == [0, 1, 2]
