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
