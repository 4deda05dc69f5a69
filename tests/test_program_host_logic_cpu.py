"""Host logic of the chunk-replay machinery without a GPU: destination
offsets of the map suite against RunningMean's own per-element table, the
arena permutation of fused ensemble launches, the run-length destination
tables.  (The replay itself is covered by the -m gpu tests; reference:
evaluation.py:583-599, 735-744.)"""
import numpy as np
import pytest

from weatherbench2_amd import evaluation


def test_map_suite_offsets_are_runningmeans_destinations():
  """MapSuite._offsets[m, dst] = first accumulator element of the slab of
  (metric m, the chunk's non-time dims): the entry RunningMean's per-element
  table holds for that slab's first grid point, for every position of the lead
  dim and any rows."""
  from weatherbench2_amd import map_suite
  rs = np.random.RandomState(0)
  cases = [
      (('metric', 'lead_time', 'level', 'latitude', 'longitude'), 'lead_time'),
      (('metric', 'level', 'lead_time', 'latitude', 'longitude'), 'lead_time'),
      (('metric', 'lead_time', 'latitude', 'longitude'), 'lead_time'),
      (('metric', 'level', 'latitude', 'longitude'), None),
  ]
  sizes = {'metric': 3, 'lead_time': 2, 'level': 4, 'latitude': 5,
           'longitude': 6}
  suite = map_suite.MapSuite.__new__(map_suite.MapSuite)
  for dims, split in cases:
    shape = tuple(sizes[d] for d in dims)
    acc = evaluation._Accumulator(dims, shape, split, 'cpu')
    rows = None
    if split is not None:
      # rows as RunningMean hands them out, after a few other labels came first
      acc.rows(np.arange(100, 103))
      rows = acc.rows(np.array([101, 7]))
      acc.shape = shape
      table = acc.destinations(rows).reshape(shape)
    else:
      table = np.arange(int(np.prod(shape))).reshape(shape)
    order = tuple(d for d in dims[1:-2])
    got = suite._offsets(acc, dims, shape, order, rows)
    want = table[..., 0, 0].reshape(sizes['metric'], -1)
    np.testing.assert_array_equal(got, want)
    # and the slabs are whole: consecutive elements behind each offset
    n_point = sizes['latitude'] * sizes['longitude']
    flat = table.reshape(-1, n_point)
    assert (np.diff(flat, axis=1) == 1).all()
  del rs


def test_run_length_destinations_expand_to_the_per_element_table():
  dims = ('metric', 'lead_time', 'level', 'latitude', 'longitude')
  shape = (2, 3, 4, 5, 6)
  acc = evaluation._Accumulator(dims, shape, 'lead_time', 'cpu')
  rows = acc.rows(np.array([30, 10, 20]))
  acc.shape = shape
  per_element = acc.destinations(rows)
  table, run = acc.destination_runs(rows)
  assert run == 4 * 5 * 6 and table.size * run == per_element.size
  expanded = (table[:, None] + np.arange(run)[None, :]).ravel()
  np.testing.assert_array_equal(expanded, per_element)
  # big rows start small and are added by doubling
  big = evaluation._Accumulator(('metric', 'lead_time', 'latitude', 'longitude'),
                                (3, 1, 2000, 3000), 'lead_time', 'meta')
  assert big.total.shape[0] == 1
  small = evaluation._Accumulator(dims, shape, 'lead_time', 'cpu')
  assert small.total.shape[0] == 8


class _FakeLaunch:
  def __init__(self, n_metric, n_region, n_total):
    self.n_metric, self.n_total = n_metric, n_total
    self.n_values = n_metric * n_region * n_total


def test_fused_ensemble_launches_permute_the_arena_consistently(monkeypatch):
  """Three ensemble passes of 2, 3 and 1 slabs between two other launches; the
  first and the third share a member stride and are fused where the first one
  stood: every old arena element (launch, metric, region, slab) must land on
  the element of the fused block that holds the same (metric, region, slab)."""
  import torch
  from weatherbench2_amd import program
  n_metric, n_region = 8, 3

  class Plan:
    def __init__(self):
      self.n_region = n_region
  plan = Plan()

  def ens(n, stride):
    la = program._EnsLaunch.__new__(program._EnsLaunch)
    la.plan, la.skipna, la.n_total = plan, False, n
    la.n_metric, la.n_values = n_metric, n_metric * n_region * n
    la.n_member, la.member_stride, la.dtype = 5, stride, torch.float32
    return la
  other_a, other_b = _FakeLaunch(5, n_region, 4), _FakeLaunch(5, n_region, 2)
  e1, e2, e3 = ens(2, 100), ens(3, 7), ens(1, 100)
  launches = [other_a, e1, e2, other_b, e3]

  class Fused:
    def __init__(self, singles, device):
      self.counts = [s.n_total for s in singles]
      self.n_total = sum(self.counts)
      self.n_metric, self.plan = n_metric, plan
      self.n_values = n_metric * n_region * self.n_total
    perm = program._EnsFused.perm
  monkeypatch.setattr(program, '_EnsFused', Fused)
  out, perm = program._fuse_ensemble_launches(launches, 'cpu')
  assert [type(x).__name__ for x in out] == ['_FakeLaunch', 'Fused',
                                             '_EnsLaunch', '_FakeLaunch']
  assert sorted(perm.tolist()) == list(range(perm.size))
  # label every old element, move it, read it back through the new layout
  labels = []
  for li, la in enumerate(launches):
    m, r, o = np.meshgrid(np.arange(la.n_metric), np.arange(n_region),
                          np.arange(la.n_total), indexing='ij')
    labels.append(np.stack([np.full(m.size, li), m.ravel(), r.ravel(),
                            o.ravel()], axis=1))
  labels = np.concatenate(labels)
  moved = np.empty_like(labels)
  moved[perm] = labels
  off = 0
  for la in out:
    block = moved[off:off + la.n_values].reshape(la.n_metric, n_region,
                                                 la.n_total, 4)
    if isinstance(la, Fused):
      # slabs 0-1 from launch 1 (e1), slab 2 from launch 4 (e3)
      assert (block[:, :, :2, 0] == 1).all() and (block[:, :, 2, 0] == 4).all()
      np.testing.assert_array_equal(block[:, :, :2, 3],
                                    np.broadcast_to(np.arange(2),
                                                    (n_metric, n_region, 2)))
      assert (block[:, :, 2, 3] == 0).all()
    for m in range(la.n_metric):
      assert (block[m, :, :, 1] == m).all()
    for r in range(n_region):
      assert (block[:, r, :, 2] == r).all()
    off += la.n_values
  # nothing to fuse: untouched
  same, none = program._fuse_ensemble_launches([other_a, e2, other_b], 'cpu')
  assert none is None and same == [other_a, e2, other_b]


def test_accumulator_without_a_count_map_keeps_the_steps_on_the_host():
  """`lazy_count`: no count tensor until somebody asks; rows grow without one;
  the steps per row come back through steps() / materialise into count."""
  import torch
  dims = ('metric', 'lead_time', 'latitude', 'longitude')
  acc = evaluation._Accumulator(dims, (2, 1, 3, 4), 'lead_time', 'cpu',
                                lazy_count=True)
  assert acc._count is None
  rows = acc.rows(np.arange(20))          # more labels than the 8 first rows
  assert acc._count is None and acc.total.shape[0] >= 20
  for r in rows[:3].tolist():
    acc.pending[r] = acc.pending.get(r, 0) + 2
  acc.pending[1] += 5
  acc.settle()                             # nothing to settle into
  assert acc._count is None and acc.pending[1] == 7
  np.testing.assert_array_equal(acc.steps([1, None, 0, 19]), [7, 0, 2, 0])
  count = acc.count                        # asked for: built from the steps
  assert acc._count is count and not acc.pending
  assert count.shape == acc.total.shape
  assert torch.equal(count[1], torch.full((2, 3, 4), 7.0, dtype=torch.float64))
  assert float(count[5].sum()) == 0.0
  # from here on it behaves like any accumulator
  acc.pending[5] = 1
  acc.settle()
  assert float(acc.count[5, 0, 0, 0]) == 1.0
  plain = evaluation._Accumulator(dims, (2, 1, 3, 4), 'lead_time', 'cpu')
  assert plain._count is not None


def test_evaluate_chunks_checks_its_configs_before_touching_a_device():
  from weatherbench2_amd import config, metrics as gm
  cfg = config.Eval(metrics={'mse': gm.MSE()})
  import dataclasses
  other = dataclasses.replace(cfg, evaluate_persistence=True)
  with pytest.raises(ValueError, match='no eval config'):
    evaluation.evaluate_chunks([(None, None)], {}, False)
  with pytest.raises(ValueError, match='baseline switches'):
    evaluation.evaluate_chunks([(None, None)], {'a': cfg, 'b': other}, False)
