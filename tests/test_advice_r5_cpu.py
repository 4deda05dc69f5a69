"""Host-logic checks behind the round-4 advisor findings: the lead-label
layout of RunningMean (xbeam.Mean's combiner, /root/reference/weatherbench2/
evaluation.py:735-744) under an RCCL communicator, duplicate labels in one
chunk result, the label order of the result, `_ByRegion` lookups, the
incremental build's dependency scan."""
import numpy as np
import pytest

from weatherbench2_amd import evaluation
from weatherbench2_amd import xarray_lite as xl

LEADS = (np.array([12, 0, 6]) * np.timedelta64(1, 'h')).astype(
    'timedelta64[ns]')   # NOT monotonic: the dataset's own order


def _chunk(i, l, values):
  return xl.Dataset(
      {'z': xl.DataArray(values[i:i + 1, l:l + 1],
                         ('init_time', 'lead_time', 'level'))},
      {'init_time': np.arange(i, i + 1), 'lead_time': LEADS[l:l + 1],
       'level': np.array([500, 850])})


def test_comm_with_a_split_dim_needs_the_full_label_list():
  values = np.random.RandomState(0).normal(size=(2, 3, 2))
  mean = evaluation.RunningMean('init_time', False, comm=object(),
                                split_dim='lead_time')
  mean.add(_chunk(0, 1, values))
  with pytest.raises(ValueError, match='split_labels'):
    mean._split_labels(['z'])
  # with the list every rank lays its rows out alike, also for labels a rank
  # never met
  mean = evaluation.RunningMean('init_time', False, comm=object(),
                                split_dim='lead_time', split_labels=LEADS)
  mean.add(_chunk(0, 1, values))
  np.testing.assert_array_equal(mean._split_labels(['z'])['z'], LEADS)
  stranger = evaluation.RunningMean(
      'init_time', False, split_dim='lead_time', split_labels=LEADS[:1])
  stranger.add(_chunk(0, 1, values))
  with pytest.raises(ValueError, match='not in the split_labels'):
    stranger._split_labels(['z'])


def test_label_order_sorted_first_seen_or_given():
  values = np.random.RandomState(1).normal(size=(3, 3, 2))
  for order, want in (('sorted', np.sort(LEADS)), ('first_seen', LEADS)):
    mean = evaluation.RunningMean('init_time', False, split_dim='lead_time',
                                  split_order=order)
    for i in range(3):
      for l in range(3):       # the chunk list in dataset order
        mean.add(_chunk(i, l, values))
    got = mean.result()
    np.testing.assert_array_equal(got.coords['lead_time'], want)
    pos = [list(LEADS).index(x) for x in want]
    np.testing.assert_allclose(got['z'].values, values.mean(0)[pos],
                               rtol=1e-15)
  with pytest.raises(ValueError):
    evaluation.RunningMean('init_time', split_order='random')


def test_repeated_labels_in_one_chunk_result_are_rejected():
  values = np.random.RandomState(2).normal(size=(1, 2, 2))
  twice = xl.Dataset(
      {'z': xl.DataArray(values, ('init_time', 'lead_time', 'level'))},
      {'init_time': np.arange(1), 'lead_time': LEADS[[1, 1]],
       'level': np.array([500, 850])})
  mean = evaluation.RunningMean('init_time', False, split_dim='lead_time')
  with pytest.raises(ValueError, match='repeated'):
    mean.add(twice)


def test_by_region_raises_key_errors():
  import torch
  from weatherbench2_amd import metrics as gm
  by = gm._ByRegion(torch.zeros((5, 2, 3)), ['global', 'tropics'])
  assert tuple(by['tropics'].shape) == (5, 3)
  with pytest.raises(KeyError):
    by['nowhere']
  assert by.get('nowhere') is None and 'nowhere' not in by


def test_build_dependencies_follow_the_includes():
  """Every header under csrc/ counts (a glob, not a hand-kept list), and the
  macro block of member counts parses without a trailing blank line."""
  import os
  from weatherbench2_amd import build
  names = {os.path.basename(h) for h in build._headers()}
  assert {'gauss_math.hpp', 'gauss_tables.inc', 'sort3_networks.inc',
          'ensemble_kernels.hpp', 'wb2hip.h'} <= names
  sizes = build.exact_sizes()
  assert (50, 64) not in sizes and (51, 64) in sizes and (100, 128) in sizes
  assert len(sizes) == len(set(sizes)) >= 17
