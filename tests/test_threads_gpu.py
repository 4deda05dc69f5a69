"""Concurrent callers (SURVEY 8b "Threading"): Beam's DirectRunner may call
compute_chunk from several worker threads
(/root/reference/weatherbench2/evaluation.py:583-599, :696-697).  Every cache of
the product is per thread and a worker thread launches on a HIP stream of its
own, so concurrent calls must give bit-identical results to sequential ones."""
import threading

import numpy as np
import pytest

from oracle import fixtures
from tests import helpers

pytestmark = pytest.mark.gpu


def _chunks(n):
  out = []
  for i in range(n):
    truth, forecast = fixtures.get_random_truth_and_forecast(
        variables=('geopotential',), spatial_resolution_in_degrees=2.0,
        seed=100 + i)
    out.append((forecast, truth))
  return out


def test_four_threads_bit_identical_and_on_their_own_streams():
  import torch
  from weatherbench2_amd import engine, metrics as gm
  dev = torch.device('cuda')
  regions = helpers.predefined_regions(oracle=False)
  g = helpers.to_gpu_dataset
  chunks = [(g(f), g(t)) for f, t in _chunks(8)]
  suite = {'mse': gm.MSE(), 'mae': gm.MAE(), 'bias': gm.Bias(),
           'rmse': gm.RMSESqrtBeforeTimeAvg()}

  def evaluate(chunk):
    f, t = chunk
    out = {}
    with gm.fused_regions(regions):
      for mname, metric in suite.items():
        for rname, region in regions.items():
          out[mname, rname] = metric.compute_chunk(
              f, t, region=region)['geopotential'].values.copy()
    return out

  sequential = [evaluate(c) for c in chunks]
  default_stream = torch.cuda.default_stream(dev).cuda_stream
  results = [None] * len(chunks)
  streams = [None] * 4
  errors = []

  def worker(w):
    try:
      for rep in range(3):          # several rounds: the caches are re-entered
        for i in range(w, len(chunks), 4):
          results[i] = evaluate(chunks[i])
      streams[w] = engine.current_stream_ptr(dev)
    except Exception as e:  # surfaced below
      errors.append(e)

  threads = [threading.Thread(target=worker, args=(w,)) for w in range(4)]
  for th in threads:
    th.start()
  for th in threads:
    th.join()
  assert not errors, errors
  assert len(set(streams)) == 4 and default_stream not in streams
  for want, got in zip(sequential, results):
    assert want.keys() == got.keys()
    for k in want:
      np.testing.assert_array_equal(got[k], want[k], err_msg=str(k))


def test_scope_drops_results_of_a_refilled_buffer():
  """a caller that refills the SAME host buffer with the next chunk
  must never see the previous chunk's numbers -- inside a loop scope (dropped
  at exit) and for bare calls (whole-buffer hash)."""
  from weatherbench2_amd import metrics as gm
  (f0, t0), (f1, t1) = _chunks(2)
  g = helpers.to_gpu_dataset
  gf, gt = g(f0), g(t0)
  regions = {'global': None}
  def run():
    with gm.fused_regions(regions):
      return gm.MSE().compute_chunk(gf, gt)['geopotential'].values.copy()
  first = run()
  bare_first = gm.MSE().compute_chunk(gf, gt)['geopotential'].values.copy()
  np.testing.assert_array_equal(first, bare_first)
  # refill in place (same ndarray objects, same ids)
  np.copyto(gf['geopotential'].data, f1['geopotential'].data)
  np.copyto(gt['geopotential'].data, t1['geopotential'].data)
  second = run()
  bare_second = gm.MSE().compute_chunk(gf, gt)['geopotential'].values.copy()
  want = gm.MSE().compute_chunk(g(f1), g(t1))['geopotential'].values
  np.testing.assert_array_equal(second, want)
  np.testing.assert_array_equal(bare_second, want)
  assert not np.array_equal(first, second)


def test_results_cross_threads_ordered():
  """a worker thread launches on a private non-blocking
  stream.  A result it produced must be complete when ANOTHER thread reads it
  (main thread: default stream; another worker: its own stream), and inputs the
  main thread produces on the default stream AFTER a worker adopted its stream
  must be complete when that worker's pass reads them.  The hand-off goes
  through the default stream: publish at scope exit, wait at every entry and
  read (engine._adopt_thread_stream / publish_thread_stream / order_read)."""
  import torch
  from weatherbench2_amd import metrics as gm
  from weatherbench2_amd import xarray_lite as xl
  dev = torch.device('cuda', 0)
  n_lat, n_lon, n_lev, n_time = 181, 360, 13, 24
  lat = np.linspace(-90, 90, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  dims = ('time', 'level', 'latitude', 'longitude')
  coords = {'time': np.arange(n_time).astype('datetime64[D]').astype(
      'datetime64[ns]'), 'level': np.arange(n_lev), 'latitude': lat,
            'longitude': lon}
  regions = helpers.predefined_regions(oracle=False)
  handoff = {}
  go, done = threading.Event(), threading.Event()
  errors = []

  def producer():
    try:
      # first touch: this thread adopts its private stream NOW
      gm.MSE().compute_chunk(handoff['f0'], handoff['t0'])
      go.wait()
      # inputs made on the main thread's default stream after the adoption
      with gm.fused_regions(regions):
        handoff['result'] = gm.MSE().compute_chunk_regions(
            handoff['f1'], handoff['t1'], regions)
      done.set()
    except Exception as e:
      errors.append(e)
      done.set()

  def ds(t):
    return xl.Dataset({'z': xl.DataArray(t, dims)}, coords)
  small = torch.zeros((n_time, n_lev, n_lat, n_lon), device=dev)
  handoff['f0'], handoff['t0'] = ds(small), ds(small.clone())
  th = threading.Thread(target=producer)
  th.start()
  # a long chain of default-stream work that ends in the worker's inputs
  g = torch.Generator(device=dev).manual_seed(3)
  a = torch.randn((n_time, n_lev, n_lat, n_lon), device=dev, generator=g)
  b = torch.randn((n_time, n_lev, n_lat, n_lon), device=dev, generator=g)
  for _ in range(40):
    a = a * 1.0001 + 0.001
    b = b * 0.9999 - 0.001
  handoff['f1'], handoff['t1'] = ds(a), ds(b)
  go.set()
  done.wait(120)
  th.join()
  assert not errors, errors
  # read on the MAIN thread (default stream) without any explicit sync
  got_main = handoff['result']['z'].values.copy()
  want = ((a.double() - b.double()) ** 2)
  w = torch.as_tensor(np.cos(np.deg2rad(lat)), device=dev)  # rough check below
  exact = gm.MSE().compute_chunk_regions(ds(a), ds(b), regions)['z'].values
  np.testing.assert_array_equal(got_main, exact)
  assert np.isfinite(got_main).all() and (got_main > 0).all()
  del want, w
  # read on ANOTHER worker thread
  seen = {}

  def reader():
    seen['values'] = handoff['result']['z'].values.copy()
  th2 = threading.Thread(target=reader)
  th2.start()
  th2.join()
  np.testing.assert_array_equal(seen['values'], exact)


def test_host_climatology_uploads_only_the_gathered_slabs(monkeypatch):
  """ACC with a HOST climatology covering the whole year
  must not upload the whole array per chunk -- only the (dayofyear, hour,
  level) slabs the chunk's valid times select cross PCIe."""
  import torch
  from weatherbench2_amd import engine, metrics as gm
  from weatherbench2_amd import xarray_lite as xl
  rs = np.random.RandomState(0)
  n_lat, n_lon = 33, 64
  lat = np.linspace(-90, 90, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  times = (np.datetime64('2020-03-01T00', 'ns')
           + np.arange(4) * np.timedelta64(12, 'h').astype('timedelta64[ns]'))
  dims = ('time', 'level', 'latitude', 'longitude')
  coords = {'time': times, 'level': np.array([500, 850]), 'latitude': lat,
            'longitude': lon}
  f = rs.standard_normal((4, 2, n_lat, n_lon)).astype(np.float32)
  t = rs.standard_normal((4, 2, n_lat, n_lon)).astype(np.float32)
  clim = rs.standard_normal((2, 366, 2, n_lat, n_lon)).astype(np.float32)
  cds = xl.Dataset({'z': xl.DataArray(clim, ('hour', 'dayofyear', 'level',
                                             'latitude', 'longitude'))},
                   {'hour': np.array([0, 12]), 'dayofyear': np.arange(1, 367),
                    'level': np.array([500, 850]), 'latitude': lat,
                    'longitude': lon})
  uploaded = []
  real = engine.as_device_tensor
  monkeypatch.setattr(engine, 'as_device_tensor',
                      lambda x, device, dtype=None: (
                          uploaded.append(getattr(x, 'nbytes', 0)),
                          real(x, device, dtype))[1])
  fds = xl.Dataset({'z': xl.DataArray(f, dims)}, coords)
  tds = xl.Dataset({'z': xl.DataArray(t, dims)}, coords)
  got = gm.ACC(climatology=cds).compute_chunk(fds, tds)['z'].values
  slab = n_lat * n_lon * 4
  assert max(uploaded) <= 8 * slab  # 4 valid times x 2 levels, not 2 x 366 x 2
  assert clim.nbytes not in uploaded
  # and the numbers are those of the full-array path
  from oracle import metrics_np as om
  from oracle.named import DS, NA
  want = om.ACC(climatology=DS({'z': NA(clim, ('hour', 'dayofyear', 'level',
                                               'latitude', 'longitude'))},
                               {'hour': np.array([0, 12]),
                                'dayofyear': np.arange(1, 367),
                                'level': np.array([500, 850]),
                                'latitude': lat, 'longitude': lon})
                ).compute_chunk(DS({'z': NA(f, dims)}, coords),
                                DS({'z': NA(t, dims)}, coords))['z'].data
  helpers.assert_close(got, want, rtol=1e-9, atol=1e-12)
  del torch
