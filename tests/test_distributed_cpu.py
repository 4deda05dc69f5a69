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


def test_shard_bounds_cover_everything():
  for n in (1, 7, 8, 2920):
    for world in (1, 2, 3, 8):
      spans = [evaluation.shard_bounds(n, world, r) for r in range(world)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
      sizes = [hi - lo for lo, hi in spans]
      assert max(sizes) - min(sizes) <= 1
