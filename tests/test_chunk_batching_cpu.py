"""Host logic of the chunk batching (no GPU): the zero-copy concatenation of
(init_time=1, lead_time=1) chunks (evaluation.py:693-705 chunking), the
windows `evaluate_chunks` forms from them, and the temporal mean keyed by lead
labels (xbeam.Mean combines per chunk key, evaluation.py:735-744)."""
import numpy as np
import pytest
import torch

from tests import helpers, official_chunks as oc
from weatherbench2_amd import evaluation
from weatherbench2_amd import xarray_lite as xl


def _product(n_init=4, n_lead=3):
  forecast, truth, _ = oc.make(n_init=n_init, n_lead=n_lead, n_lat=7, n_lon=8)
  return helpers.to_gpu_dataset(forecast), helpers.to_gpu_dataset(truth)


def test_slab_concat_indexes_its_bases():
  rs = np.random.RandomState(0)
  bases = [rs.normal(size=(2, 3, 4, 5)), rs.normal(size=(1, 4, 5)),
           rs.normal(size=(3, 4, 5))]
  flat = np.concatenate([b.reshape(-1, 4, 5) for b in bases])
  index = rs.permutation(flat.shape[0]).reshape(2, 5)
  sc = xl.SlabConcat(bases, index)
  assert sc.shape == (2, 5, 4, 5) and sc.ndim == 4 and sc.dtype == flat.dtype
  np.testing.assert_array_equal(np.asarray(sc), flat[index])
  np.testing.assert_array_equal(np.asarray(sc[1]), flat[index[1]])
  np.testing.assert_array_equal(np.asarray(sc.permute_outer((1, 0))),
                                flat[index.T])
  tens = xl.SlabConcat([torch.from_numpy(b) for b in bases], index)
  np.testing.assert_array_equal(tens.materialize().numpy(), flat[index])
  # addresses: base pointer of the owning array + slab offset inside it
  addr = tens.addresses()
  step = 4 * 5 * 8
  for pos, g in np.ndenumerate(index):
    b = int(np.searchsorted(tens.offsets, g, side='right') - 1)
    assert addr[pos] == tens.bases[b].data_ptr() + (g - tens.offsets[b]) * step
  with pytest.raises(IndexError):
    xl.SlabConcat(bases, [[flat.shape[0]]])
  with pytest.raises(ValueError):
    xl.SlabConcat([bases[0], rs.normal(size=(2, 4, 6))], [0])


@pytest.mark.parametrize('order', ['init', 'lead'])
def test_concat_chunks_is_the_concatenation(order):
  forecast, truth = _product()
  pairs = oc.chunk_pairs(forecast, truth, order)
  joined = evaluation.concat_chunks([p[0] for p in pairs], 'init_time',
                                    'lead_time')
  assert joined is not None
  # labels come back in chunk order (blocks by first appearance)
  np.testing.assert_array_equal(np.sort(joined.coords['init_time']),
                                forecast.coords['init_time'])
  i_of = [list(forecast.coords['init_time']).index(v)
          for v in joined.coords['init_time']]
  l_of = [list(forecast.coords['lead_time']).index(v)
          for v in joined.coords['lead_time']]
  for name, da in joined.items():
    assert isinstance(da.data, xl.SlabConcat)
    assert da.dims == forecast[name].dims
    want = np.asarray(forecast[name].data)[i_of][:, l_of]
    np.testing.assert_array_equal(np.asarray(da.data), want)
  vt = joined.coords['valid_time']
  np.testing.assert_array_equal(
      vt.values, np.asarray(forecast.coords['valid_time'].values)[i_of][:, l_of])


def test_concat_chunks_refuses_what_is_not_a_rectangle():
  forecast, truth = _product()
  pairs = oc.chunk_pairs(forecast, truth)
  fs = [p[0] for p in pairs]
  assert evaluation.concat_chunks(fs[:5], 'init_time', 'lead_time') is None
  assert evaluation.concat_chunks([fs[0], fs[0]], 'init_time',
                                  'lead_time') is None
  # a lazily gathered variable cannot be addressed in place
  lazy = xl.Dataset(coords=fs[1].coords)
  for k, v in fs[1].items():
    lazy[k] = xl.DataArray(
        xl.SlabGather(np.asarray(v.data), np.arange(
            int(np.prod(v.shape[:-2]))).reshape(v.shape[:-2])), v.dims)
  assert evaluation.concat_chunks([fs[0], lazy], 'init_time',
                                  'lead_time') is None
  # one lead per piece when the window is ragged
  pieces = evaluation._batches(pairs[:5], 'init_time', 'lead_time')
  assert sorted(p[0].sizes['init_time'] for p in pieces) == [1, 2, 2]


def test_running_mean_keeps_lead_blocks_apart():
  rs = np.random.RandomState(1)
  leads = (np.arange(3) * np.timedelta64(6, 'h')).astype('timedelta64[ns]')
  values = rs.normal(size=(5, 3, 2))  # (init, lead, level)
  values[1, 2, 0] = np.nan
  for skipna in (False, True):
    mean = evaluation.RunningMean('init_time', skipna, split_dim='lead_time')
    for i in range(5):
      for l in (2, 0, 1):  # any order
        mean.add(xl.Dataset(
            {'z': xl.DataArray(values[i:i + 1, l:l + 1],
                               ('init_time', 'lead_time', 'level'))},
            {'init_time': np.arange(i, i + 1), 'lead_time': leads[l:l + 1],
             'level': np.array([500, 850])}))
    got = mean.result()
    np.testing.assert_array_equal(got.coords['lead_time'], leads)
    want = (np.nanmean if skipna else np.mean)(values, axis=0)
    np.testing.assert_allclose(got['z'].values, want, rtol=1e-15,
                               equal_nan=True)
  # whole-lead chunks behave as before (one block, labels kept as given)
  mean = evaluation.RunningMean('init_time', False, split_dim='lead_time')
  for i in range(5):
    mean.add(xl.Dataset(
        {'z': xl.DataArray(values[i:i + 1], ('init_time', 'lead_time',
                                             'level'))},
        {'init_time': np.arange(i, i + 1), 'lead_time': leads,
         'level': np.array([500, 850])}))
  np.testing.assert_allclose(mean.result()['z'].values, values.mean(0),
                             rtol=1e-15, equal_nan=True)


def test_running_mean_rows_are_per_label_whatever_the_chunks_carry():
  """A window of two leads, then single leads, then all three: every label
  keeps ONE accumulator row."""
  rs = np.random.RandomState(2)
  leads = (np.arange(3) * np.timedelta64(6, 'h')).astype('timedelta64[ns]')
  values = rs.normal(size=(4, 3, 2))

  def piece(i, sel):
    return xl.Dataset(
        {'z': xl.DataArray(values[i:i + 1][:, sel],
                           ('init_time', 'lead_time', 'level'))},
        {'init_time': np.arange(i, i + 1), 'lead_time': leads[sel],
         'level': np.array([500, 850])})
  mean = evaluation.RunningMean('init_time', False, split_dim='lead_time')
  mean.add(piece(0, [1, 2]))
  mean.add(piece(0, [0]))
  for sel in ([2], [0], [1]):
    mean.add(piece(1, sel))
  mean.add(piece(2, [0, 1, 2]))
  mean.add(piece(3, [2, 0]))
  mean.add(piece(3, [1]))
  got = mean.result()
  np.testing.assert_array_equal(got.coords['lead_time'], leads)
  np.testing.assert_allclose(got['z'].values, values.mean(0), rtol=1e-15)


class _EchoMetric:
  """A stand-in metric that reads concatenated chunks in place (so that
  evaluate_chunks may form windows) -- the loop itself is replaced below."""
  _reads_slabs_in_place = True


def test_default_window_follows_the_bytes_of_the_first_chunk(monkeypatch):
  """evaluate_chunks(batch_chunks=None): the window is what holds
  AUTO_BATCH_BYTES of (forecast + truth) input, at most AUTO_BATCH_MAX chunks;
  an explicit batch_chunks wins; the prefetch depth follows the window only
  once it is known."""
  from weatherbench2_amd import config
  forecast, truth = _product(n_init=6, n_lead=3)
  pairs = oc.chunk_pairs(forecast, truth, 'init')
  per_chunk = sum(evaluation._input_bytes(ds) for ds in pairs[0])
  assert per_chunk == sum(
      int(np.prod(da.shape)) * 4 for ds in pairs[0]
      for da in ds.data_vars.values())
  windows = []

  def loop(fc, tr, eval_config, skipna, compute_chunk=True):
    n_time = fc.sizes.get('init_time', 1)
    n_lead = fc.sizes.get('lead_time', 1)
    windows.append(n_time * n_lead)
    out = xl.Dataset(coords={'init_time': fc.coords['init_time'],
                             'lead_time': fc.coords['lead_time']})
    out.data_vars['x'] = xl.DataArray(
        np.ones((n_time, n_lead)), ('init_time', 'lead_time'), out.coords, 'x')
    return out
  monkeypatch.setattr(evaluation, '_metric_and_region_loop', loop)
  cfg = config.Eval(metrics={'echo': _EchoMetric()})

  def run(**kw):
    del windows[:]
    res = evaluation.evaluate_chunks(pairs, cfg, prefetch=0, **kw)
    np.testing.assert_array_equal(res['x'].values, np.ones(3))
    return list(windows)
  assert run() == [18]                        # all 18 chunks fit one window
  monkeypatch.setattr(evaluation, 'AUTO_BATCH_MAX', 6)
  assert run() == [6, 6, 6]
  monkeypatch.setattr(evaluation, 'AUTO_BATCH_BYTES', 3 * per_chunk)
  assert run() == [3] * 6
  monkeypatch.setattr(evaluation, 'AUTO_BATCH_BYTES', per_chunk // 2)
  assert run() == [1] * 18                    # a chunk larger than the budget
  assert run(batch_chunks=9) == [9, 9]        # explicit: not capped by either
  # a metric that cannot read windows in place: chunk by chunk
  cfg_plain = config.Eval(metrics={'echo': object()})
  del windows[:]
  evaluation.evaluate_chunks(pairs, cfg_plain, prefetch=0)
  assert windows == [1] * 18


def test_prefetch_depth_may_change_while_running():
  fetched = []

  class Lazy:
    def __len__(self):
      return 10

    def __getitem__(self, i):
      fetched.append(i)
      return i
  depth = [1]
  seen = []
  for item in evaluation._prefetched(Lazy(), 0, 10, lambda: depth[0]):
    seen.append(item)
    if item == 2:
      depth[0] = 4
  assert seen == list(range(10)) and sorted(fetched) == list(range(10))


@pytest.mark.parametrize('order', ['init', 'lead'])
def test_concat_chunks_remembered_layout_gives_the_same_concatenation(order):
  """Windows of one evaluation come with the same rectangle again and again:
  the second concat takes the remembered layout (one check + one list per
  variable) and must equal the first way of doing it -- other arrays, same
  index; a chunk that does not fit is still refused."""
  forecast, truth = _product(n_init=4, n_lead=3)
  other_f, _ = _product(n_init=4, n_lead=3)
  for name, da in other_f.items():   # other values, same labels and shapes
    other_f.data_vars[name] = xl.DataArray(da.data + 1.0, da.dims,
                                           other_f.coords, name)
  evaluation._CONCAT_PLANS.clear()
  first = evaluation.concat_chunks(
      [p[0] for p in oc.chunk_pairs(forecast, truth, order)], 'init_time',
      'lead_time')
  assert len(evaluation._CONCAT_PLANS) == 1
  chunks = [p[0] for p in oc.chunk_pairs(other_f, truth, order)]
  fast = evaluation.concat_chunks(chunks, 'init_time', 'lead_time')
  evaluation._CONCAT_PLANS.clear()
  slow = evaluation.concat_chunks(chunks, 'init_time', 'lead_time')
  for name in slow.keys():
    a, b = fast[name].data, slow[name].data
    assert fast[name].dims == slow[name].dims and a.shape == b.shape
    np.testing.assert_array_equal(a.index, b.index)
    np.testing.assert_array_equal(a.index, first[name].data.index)
    np.testing.assert_array_equal(a.offsets, b.offsets)
    assert len(a.bases) == len(b.bases)
    assert all(x is y for x, y in zip(a.bases, b.bases))
    np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
  for k in slow.coords:
    x, y = fast.coords[k], slow.coords[k]
    np.testing.assert_array_equal(
        np.asarray(x.values if isinstance(x, xl.DataArray) else x),
        np.asarray(y.values if isinstance(y, xl.DataArray) else y))
  # a strided array in the second window: no concatenation in place
  bad = list(chunks)
  name = next(iter(bad[3].keys()))
  da = bad[3][name]
  wide = np.zeros(da.shape[:-1] + (2 * da.shape[-1],), dtype=da.dtype)
  bad[3] = xl.Dataset(dict(bad[3].data_vars), bad[3].coords)
  bad[3].data_vars[name] = xl.DataArray(wide[..., ::2], da.dims,
                                        bad[3].coords, name)
  assert evaluation.concat_chunks(bad, 'init_time', 'lead_time') is None
  # ... nor with another dtype, another shape, other dims or a variable missing
  # in one chunk (every refusal of the remembered layout's whole-list checks)
  def replaced(make):
    out = list(chunks)
    ref = out[5][name]
    out[5] = xl.Dataset(dict(out[5].data_vars), out[5].coords)
    new = make(ref)
    if new is None:
      del out[5].data_vars[name]
    else:
      out[5].data_vars[name] = new
    return out
  as_array = lambda da: np.asarray(da.values)
  assert evaluation.concat_chunks(chunks, 'init_time', 'lead_time') is not None
  for make in (
      lambda da: xl.DataArray(as_array(da).astype(np.float16)
                              if isinstance(da.data, np.ndarray) else
                              da.data.to(torch.float16), da.dims,
                              chunks[5].coords, name),
      lambda da: xl.DataArray(da.data[..., :-1].copy()
                              if isinstance(da.data, np.ndarray) else
                              da.data[..., :-1].contiguous(), da.dims,
                              chunks[5].coords, name),
      lambda da: xl.DataArray(da.data, da.dims[:-2] + da.dims[:-3:-1],
                              chunks[5].coords, name),
      lambda da: None):
    assert evaluation.concat_chunks(replaced(make), 'init_time',
                                    'lead_time') is None
