"""The metric x region loop and the temporal mean, MI355X-native.

Mirrors (reference = /root/reference/weatherbench2/evaluation.py):

  _metric_and_region_loop(forecast, truth, eval_config, skipna, compute_chunk)
      :388-438  same signature and the same result layout -- a Dataset whose
      variables carry leading (metric, region) dims (regions concatenated,
      metrics merged, NaN-filled where a metric lacks a variable).  Instead of
      re-running every metric per region, the regions are announced up front so
      each fused HIP pass serves all of them; derived variables are computed
      first and assigned INTO forecast/truth exactly like the reference does
      (:402-405).  Metric objects that are not ours (any object with
      `compute_chunk`) still work: they are simply called per region.

  TemporalMean / xbeam.Mean(dim, skipna)      :735-744
      `RunningMean`: per-rank (sum, count) accumulators on the device
      (wb2_time_accumulate), combined across ranks with ONE all-reduce -- RCCL
      over xGMI when the process group is NCCL, gloo in the CPU tests.

  truth.sel(time=forecast.valid_time)         :474-475
      `select_truth_at_valid_time`: the by-init truth gather, on the device.

  evaluate_in_memory's per-config driver      :441-517
      `evaluate_chunks`: init-time chunks sharded contiguously over ranks
      (SURVEY.md 8e), one fused pass per chunk, one all-reduce at the end.

  _evaluate_all_metrics                       :441-483
      same name and leading arguments, but the datasets arrive already opened
      (dataset opening / NetCDF writing are the IO layer, out of scope) and the
      result is returned.  Its baseline substitutions are GATHERS and cost no
      copy: forecast := climatology.sel(dayofyear, hour) of the valid time
      (:452-460), the probabilistic climatology (:461-471 + utils.py:47-70),
      the persistence forecast (:165-193) come back as `xarray_lite.SlabGather`
      arrays -- an index table over the resident source -- which the fused
      deterministic passes read through their slab tables.

  _climatology_like_forecast_chunk / _persistence_like_forecast_chunk
                                              :618-675
      the per-chunk versions of the Beam pipeline, same gathers.
"""
from __future__ import annotations

import contextlib
import logging
import typing as t

import numpy as np

from weatherbench2_amd import config
from weatherbench2_amd import metrics as metrics_lib
from weatherbench2_amd import xarray_lite as xl


def _metric_and_region_loop(
    forecast,
    truth,
    eval_config: config.Eval,
    skipna: bool,
    compute_chunk: bool = False,
) -> xl.Dataset:
  """Compute metric results looping over metrics and regions in eval config."""
  given_forecast, given_truth = forecast, truth
  # Derived variables are computed on what the caller handed in and assigned
  # into it, like the reference does (:402-405): reference DerivedVariable
  # objects keep seeing xarray Datasets, ours see either kind.
  for name, dv in eval_config.derived_variables.items():
    forecast[name] = dv.compute(forecast)
    truth[name] = dv.compute(truth)
  forecast = xl.as_dataset(forecast)
  truth = xl.as_dataset(truth)

  results = []
  regions = eval_config.regions
  acc = next((m for m in eval_config.metrics.values()
              if isinstance(m, metrics_lib.ACC) and m.climatology is not None),
             None)
  with contextlib.ExitStack() as stack:
    stack.enter_context(metrics_lib.fused_regions(regions))
    if acc is not None:
      stack.enter_context(metrics_lib.fused_climatology(acc.climatology))
    stack.enter_context(metrics_lib.fused_wind_vectors(_wind_pairs(eval_config)))
    for name, metric in eval_config.metrics.items():
      if compute_chunk or not eval_config.temporal_mean:
        eval_fn = metric.compute_chunk
      else:
        eval_fn = metric.compute
      if regions is not None and isinstance(metric, metrics_lib.Metric):
        # our metrics answer for every region at once (leading `region` dim)
        regions_fn = (metric.compute_chunk_regions
                      if eval_fn == metric.compute_chunk
                      else metric.compute_regions)
        result = xl.as_dataset(regions_fn(
            forecast, truth, regions, skipna)).expand_dims({'metric': [name]})
      elif regions is not None:
        tmp_results = []
        for region_name, region in regions.items():
          tmp_result = xl.as_dataset(eval_fn(
              forecast=forecast, truth=truth, region=region, skipna=skipna))
          tmp_results.append(tmp_result.expand_dims(
              {'metric': [name], 'region': [region_name]}))
        result = xl.concat(tmp_results, 'region')
      else:
        result = xl.as_dataset(eval_fn(
            forecast=forecast, truth=truth, skipna=skipna)).expand_dims(
                {'metric': [name]})
      results.append(result)
  return xl.like_input(xl.merge(results), given_forecast, given_truth)


def _wind_pairs(eval_config) -> list:
  """(u_name, v_name) of every wind-vector metric the loop will evaluate, on
  its own or inside an MSE / RMSE (scripts/evaluate.py:279-311, 420-430)."""
  pairs: list = []
  for metric in eval_config.metrics.values():
    nested = (list(getattr(metric, 'wind_vector_mse', None) or []) +
              list(getattr(metric, 'wind_vector_rmse', None) or []))
    for m in [metric] + nested:
      if isinstance(m, metrics_lib.WindVectorMSE):
        pair = (m.u_name, m.v_name)
        if pair not in pairs:
          pairs.append(pair)
  return pairs


def make_latitude_increasing(dataset):
  """Flips the latitude axis if it is decreasing (evaluation.py:41-47), where
  the data lives (`torch.flip` for device arrays): the latitude weights of
  metrics.py:40-60 require increasing latitudes."""
  import torch
  ds = xl.as_dataset(dataset)
  lat = np.asarray(ds.coords['latitude'])
  if not (len(lat) > 1 and (np.diff(lat) < 0).all()):
    return dataset
  coords = dict(ds.coords)
  coords['latitude'] = lat[::-1].copy()
  out = xl.Dataset(coords=coords, attrs=dict(ds.attrs))
  for name, da in ds.data_vars.items():
    if 'latitude' not in da.dims:
      out.data_vars[name] = da
      continue
    ax = da.dims.index('latitude')
    if isinstance(da.data, torch.Tensor):
      data = torch.flip(da.data, (ax,))
    else:
      data = np.flip(np.asarray(da.data), ax)
    out.data_vars[name] = xl.DataArray(data, da.dims, coords, name)
  return xl.like_input(out, dataset)


def _affine_time_view(data, ax: int, index: np.ndarray):
  """index[i, l] == a + b*i + c*l (regularly spaced inits and leads): the
  selection is an overlapping strided VIEW of `data` (dims `ax` -> (i, l)),
  which the fused passes read through a slab table without copying; None if
  the index is not affine or the tensor is not contiguous."""
  import torch
  if not data.is_contiguous() or index.ndim != 2:
    return None
  n_i, n_l = index.shape
  a = int(index[0, 0])
  b = int(index[1, 0] - a) if n_i > 1 else 0
  c = int(index[0, 1] - a) if n_l > 1 else 0
  if b < 0 or c < 0:
    return None
  want = a + b * np.arange(n_i)[:, None] + c * np.arange(n_l)[None, :]
  if not np.array_equal(want, index):
    return None
  st = list(data.stride())
  size = list(data.shape[:ax]) + [n_i, n_l] + list(data.shape[ax + 1:])
  stride = st[:ax] + [b * st[ax], c * st[ax]] + st[ax + 1:]
  return torch.as_strided(data, size, stride,
                          data.storage_offset() + a * st[ax])


def select_truth_at_valid_time(truth, forecast, time_dim: str = 'time',
                               init_dim: str = 'init_time',
                               lead_dim: str = 'prediction_timedelta'):
  """`truth.sel(time=forecast.valid_time)` (evaluation.py:474-475) for a
  by-init forecast: truth gets the forecast's (init_time, lead) dims.

  For device-resident truth with regularly spaced init and lead times the
  result is an overlapping strided VIEW (no copy: the fused passes resolve it
  through their slab tables, metrics._physical_slabs); irregular selections
  fall back to one `index_select` on the device.  Labels missing from
  `truth.time` raise KeyError like `.sel`.
  """
  import torch
  given = (truth, forecast)
  truth, forecast = xl.as_dataset(truth), xl.as_dataset(forecast)
  init = np.asarray(forecast.coords[init_dim])
  lead = np.asarray(forecast.coords[lead_dim])
  valid = forecast.coords.get('valid_time')
  if isinstance(valid, xl.DataArray) and set(valid.dims) == {init_dim, lead_dim}:
    valid = np.asarray(valid.transpose(init_dim, lead_dim).values)
  else:
    valid = init[:, None] + lead[None, :]
  have = np.asarray(truth.coords[time_dim])
  pos = {v: i for i, v in enumerate(have.tolist())}
  try:
    index = np.array([pos[v] for v in valid.ravel().tolist()], dtype=np.int64)
  except KeyError as e:
    raise KeyError(f'not all valid times found in truth.{time_dim}: {e}') from e
  coords = {k: v for k, v in truth.coords.items()
            if k != time_dim and not (isinstance(v, xl.DataArray)
                                      and time_dim in v.dims)}
  coords[init_dim] = init
  coords[lead_dim] = lead
  coords['valid_time'] = xl.DataArray(valid, (init_dim, lead_dim))
  # xarray's vectorised .sel keeps the indexed coordinate, now over the
  # indexer's dims: thresholds.py:140 reads truth['time'] from it
  coords[time_dim] = xl.DataArray(valid, (init_dim, lead_dim))
  out = xl.Dataset(coords=coords, attrs=dict(truth.attrs))
  for name, da in truth.data_vars.items():
    if time_dim not in da.dims:
      out.data_vars[name] = da
      continue
    ax = da.dims.index(time_dim)
    shape = da.shape[:ax] + valid.shape + da.shape[ax + 1:]
    if isinstance(da.data, torch.Tensor):
      view = _affine_time_view(da.data, ax, index.reshape(valid.shape))
      if view is not None:  # regular init / lead steps: no copy at all
        data = view
      else:
        idx = torch.as_tensor(index, device=da.data.device)
        data = torch.index_select(da.data, ax, idx).reshape(shape)
    else:
      data = np.take(np.asarray(da.data), index, axis=ax).reshape(shape)
    dims = da.dims[:ax] + (init_dim, lead_dim) + da.dims[ax + 1:]
    out.data_vars[name] = xl.DataArray(data, dims, coords, name)
  return xl.like_input(out, *given)


class RunningMean:
  """xbeam.Mean's (sum, count) combiner, kept on the device.

  add(chunk_result) accumulates `sum` and `count` over `dim` (NaNs add to
  neither when skipna); result() all-reduces both across the process group (if
  one is initialised) and divides.

  Chunks may split another dim as well -- the official 0.25-degree runs use
  `input_chunks=init_time=1,lead_time=1` (docs/source/official-evaluation.md:
  537-549) --: xbeam.Mean combines per key of the remaining chunk offsets, so
  results with different `split_dim` labels accumulate separately and are
  laid side by side, in label order, by result() (`split_dim`: the forecast's
  lead dim when the results have one).
  """

  def __init__(self, dim: str, skipna: bool = False, device=None, comm=None,
               split_dim: t.Optional[str] = None):
    self.dim = dim
    self.skipna = skipna
    self.device = device
    self.split_dim = split_dim
    # an RCCL communicator from engine.comm_init_rank: the exchange then goes
    # through the C ABI (wb2_time_mean_allreduce) instead of torch.distributed
    self.comm = comm
    self._acc: dict = {}     # (var, split labels) -> (sum, count, dims, shape)
    self._coords: dict = {}
    self._labels: dict = {}  # split labels key -> label array

  def _split_key(self, chunk: xl.Dataset, da: xl.DataArray):
    d = self.split_dim
    if d is None or d not in da.dims or d not in chunk.coords:
      return None
    labels = np.asarray(chunk.coords[d])
    key = (labels.dtype.str, labels.tobytes())
    self._labels.setdefault(key, labels)
    return key

  def _tensors(self, name, key, dims, shape):
    import torch
    if (name, key) not in self._acc:
      total = torch.zeros(shape, dtype=torch.float64, device=self.device)
      self._acc[(name, key)] = (total, torch.zeros_like(total), dims, shape)
    total, count, d, s = self._acc[(name, key)]
    if d != dims or s != shape:
      raise ValueError(f'{name}: chunk layout changed {d}{s} -> {dims}{shape}')
    return total, count

  def add(self, chunk: xl.Dataset):
    import torch
    from weatherbench2_amd import engine
    for k, c in chunk.coords.items():
      # coordinates that vary along the averaged dim go with it (valid_time of
      # a by-init chunk), like xarray's mean; the split dim's labels are put
      # back together by result()
      if k != self.dim and k != self.split_dim and not (
          isinstance(c, xl.DataArray) and self.dim in c.dims):
        self._coords.setdefault(k, c)
    for name, da in chunk.data_vars.items():
      if self.dim not in da.dims:
        raise ValueError(f'{name} has no {self.dim!r} dim: {da.dims}')
      axis = da.dims.index(self.dim)
      dims = tuple(d for d in da.dims if d != self.dim)
      shape = tuple(n for d, n in zip(da.dims, da.shape) if d != self.dim)
      raw = da.data
      if isinstance(raw, torch.Tensor):
        # map-valued results (Spatial* metrics, rank histograms) already live
        # on the device: accumulate there, no host round trip
        if self.device is None and raw.is_cuda:
          self.device = raw.device
        engine.order_read(raw)  # produced on another thread's stream?
        values = raw.to(torch.float64)
      else:
        values = torch.as_tensor(np.ascontiguousarray(da.values),
                                 dtype=torch.float64)
      total, count = self._tensors(name, self._split_key(chunk, da), dims,
                                   shape)
      if self.device is not None and torch.device(self.device).type == 'cuda':
        engine.time_accumulate(values.to(self.device).contiguous(), axis,
                               self.skipna, total, count)
      else:  # host accumulators (CPU tests of the sharding logic)
        values = values.cpu()
        ok = ~torch.isnan(values) if self.skipna else torch.ones_like(
            values, dtype=torch.bool)
        total += torch.where(ok, values, torch.zeros_like(values)).sum(axis)
        count += ok.to(torch.float64).sum(axis)

  def _agree_on_layout(self):
    """Every rank enters the all-reduce with the same accumulators in the same
    order: ranks whose shard never met some split label (lead-major chunk
    lists) get zero accumulators for it."""
    import torch
    import torch.distributed as dist
    mine = [(n, k, self._acc[(n, k)][2], self._acc[(n, k)][3],
             None if k is None else self._labels[k]) for n, k in self._acc]
    everyone: list = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, mine)
    for layout in everyone:
      for n, k, dims, shape, labels in layout:
        if k is not None:
          self._labels.setdefault(k, labels)
        if (n, k) not in self._acc:
          total = torch.zeros(shape, dtype=torch.float64, device=self.device)
          self._acc[(n, k)] = (total, torch.zeros_like(total), dims, shape)

  def _ordered(self) -> list:
    """Accumulators by variable, the blocks of a split dim by their first
    label (chunk offsets of a sorted lead coordinate)."""
    def order(item):
      name, key = item
      if key is None:
        return (name, 0, 0, b'')
      first = self._labels[key].ravel()[:1]
      if first.size and first.dtype.kind in 'mMiu':
        return (name, 1, int(first.astype('int64')[0]), key[1])
      if first.size and first.dtype.kind == 'f':
        return (name, 1, float(first[0]), key[1])
      return (name, 1, 0, key[1])
    return sorted(self._acc, key=order)

  def result(self) -> xl.Dataset:
    import torch
    import torch.distributed as dist
    in_torch_group = (dist.is_available() and dist.is_initialized()
                      and dist.get_world_size() > 1)
    in_group = self.comm is not None or in_torch_group
    if in_group and not self._acc:
      # every rank must enter the collective with the same layout: a rank
      # without a single chunk would leave the others waiting in the all-reduce
      raise ValueError('RunningMean.result() on a rank that accumulated '
                       'nothing: give every rank at least one chunk '
                       '(evaluate_chunks checks this up front)')
    if in_torch_group and self.comm is None:
      self._agree_on_layout()
    names = self._ordered()
    if self.comm is not None and names:
      from weatherbench2_amd import engine
      flat_t = torch.cat([self._acc[n][0].reshape(-1) for n in names])
      flat_c = torch.cat([self._acc[n][1].reshape(-1) for n in names])
      engine.time_mean_allreduce(flat_t, flat_c, self.comm)
      offset = 0
      for n in names:
        total, count, dims, shape = self._acc[n]
        size = total.numel()
        self._acc[n] = (flat_t[offset:offset + size].reshape(shape),
                        flat_c[offset:offset + size].reshape(shape), dims, shape)
        offset += size
    elif in_torch_group and names:
      flat = torch.cat([torch.stack([self._acc[n][0], self._acc[n][1]]
                                    ).reshape(-1) for n in names])
      dist.all_reduce(flat)  # the path's only exchange step
      offset = 0
      for n in names:
        total, count, dims, shape = self._acc[n]
        size = total.numel()
        self._acc[n] = (flat[offset:offset + size].reshape(shape),
                        flat[offset + size:offset + 2 * size].reshape(shape),
                        dims, shape)
        offset += 2 * size
    coords = dict(self._coords)
    pieces: dict = {}
    for name, key in names:
      total, count, dims, shape = self._acc[(name, key)]
      mean = (total / count).cpu().numpy()  # 0/0 -> NaN like an empty mean
      pieces.setdefault(name, []).append((key, dims, mean))
    split_labels = None
    out_vars = {}
    for name, parts in pieces.items():
      dims = parts[0][1]
      if len(parts) == 1 and parts[0][0] is None:
        out_vars[name] = (dims, parts[0][2])
        continue
      ax = dims.index(self.split_dim)
      labels = np.concatenate([self._labels[k] for k, _, _ in parts])
      out_vars[name] = (dims, np.concatenate([m for _, _, m in parts],
                                             axis=ax))
      if split_labels is not None and not np.array_equal(split_labels, labels):
        raise ValueError(f'{name}: {self.split_dim} labels differ between '
                         'variables')
      split_labels = labels
    if split_labels is not None:
      coords[self.split_dim] = split_labels
    out = xl.Dataset(coords=coords)
    for name, (dims, mean) in out_vars.items():
      out.data_vars[name] = xl.DataArray(mean, dims, coords, name)
    return out


def make_resident(dataset, device=None) -> xl.Dataset:
  """Uploads every data variable of `dataset` to HBM once and returns a Dataset
  of device tensors with the same dims and coordinates.

  Meant for the inputs that recur across chunks -- the climatology of an ACC /
  SEEPS / threshold metric (366 x 4 x 13 x 721 x 1440 float32 = 79 GB for one
  variable fits the 288 GB of an MI355X), a truth dataset every lead time
  selects from: the metrics gather from a resident array by slab index and
  nothing but the forecast crosses PCIe per chunk (4 instead of 12 B per point
  for the headline pass).  Coordinates stay on the host: labels are host work."""
  import torch
  from weatherbench2_amd import engine
  ds = xl.as_dataset(dataset)
  dev = torch.device(device) if device is not None else engine.require_gpu()
  out = xl.Dataset(coords=ds.coords, attrs=ds.attrs)
  for name, var in ds.data_vars.items():
    out[name] = xl.DataArray(engine.as_device_tensor(var.data, dev), var.dims,
                             ds.coords, name)
  return out


def shard_bounds(n_items: int, world_size: int, rank: int) -> tuple[int, int]:
  """Contiguous, balanced [lo, hi) block of `n_items` for `rank`."""
  base, extra = divmod(n_items, world_size)
  lo = rank * base + min(rank, extra)
  return lo, lo + base + (1 if rank < extra else 0)


def _prefetched(chunks, lo: int, hi: int, depth: int):
  """Yields chunks[lo], ..., chunks[hi - 1] in order, fetching up to `depth`
  items ahead on ONE background thread (so the fetches themselves stay in
  order -- sequences that read a file sequentially keep doing so)."""
  if depth <= 0 or hi - lo <= 1:
    for i in range(lo, hi):
      yield chunks[i]
    return
  import collections
  from concurrent import futures
  from weatherbench2_amd import engine
  # the fetch thread may upload (a lazy chunks[i] calling make_resident): it
  # must not get a private stream -- its uploads are consumed by THIS thread
  with futures.ThreadPoolExecutor(
      max_workers=1, thread_name_prefix='wb2hip-prefetch',
      initializer=engine.disable_thread_stream) as pool:
    pending: collections.deque = collections.deque()
    nxt = lo
    try:
      while nxt < hi or pending:
        while nxt < hi and len(pending) <= depth:
          pending.append(pool.submit(chunks.__getitem__, nxt))
          nxt += 1
        yield pending.popleft().result()
    finally:
      for fut in pending:
        fut.cancel()


def _chunk_substitution(eval_config, truth, climatology, by_init: bool):
  """The forecast replacement of `_EvaluateAllMetrics._evaluate`
  (evaluation.py:677-733) as one function of a chunk, or None."""
  steps = []
  if getattr(eval_config, 'evaluate_climatology', False):
    if climatology is None:
      raise ValueError('eval_config.evaluate_climatology is set: pass the '
                       'climatology dataset (climatology=...)')
    steps.append(lambda f, tc: _climatology_like_forecast_chunk(
        f, tc, climatology, list(f.keys()), by_init)[0])
  if getattr(eval_config, 'evaluate_probabilistic_climatology', False):
    if truth is None:
      raise ValueError('eval_config.evaluate_probabilistic_climatology is set: '
                       'pass the full ground-truth dataset (truth=...)')
    made: list = []

    def prob(f, tc):
      if not made:  # once: label work over truth.time
        made.append(make_probabilistic_climatology(
            truth, eval_config.probabilistic_climatology_start_year,
            eval_config.probabilistic_climatology_end_year,
            eval_config.probabilistic_climatology_hour_interval,
            variables=list(f.keys())))
      return _climatology_like_forecast_chunk(f, tc, made[0], list(f.keys()),
                                              by_init)[0]
    steps.append(prob)
  elif getattr(eval_config, 'evaluate_persistence', False):
    if truth is None:
      raise ValueError('eval_config.evaluate_persistence is set: pass the full '
                       'ground-truth dataset (truth=...)')
    steps.append(lambda f, tc: _persistence_like_forecast_chunk(
        f, tc, truth, list(f.keys()), by_init)[0])
  if not steps:
    return None

  def substitute(forecast, truth_chunk):
    # like the pipeline, every step sees the ORIGINAL chunk's labels
    out = forecast
    for step in steps:
      out = step(forecast, truth_chunk)
    return out
  return substitute


def _plain_slabs(da: xl.DataArray) -> bool:
  """Data a concatenation can address slab by slab: a C-contiguous numpy array
  or torch tensor with the two spatial dims last."""
  data = da.data
  if isinstance(data, (xl.SlabGather, xl.SlabConcat)) or da.ndim < 2:
    return False
  if set(da.dims[-2:]) != {'latitude', 'longitude'}:
    return False
  if isinstance(data, np.ndarray):
    return bool(data.flags.c_contiguous)
  return bool(data.is_contiguous())


def _block_matrix(cells: dict, n_i: int, n_l: int, ax_i, ax_l):
  """np.block over a (time block, lead block) grid of arrays."""
  rows = []
  for bi in range(n_i):
    row = [cells[(bi, bl)] for bl in range(n_l)]
    rows.append(row[0] if ax_l is None or n_l == 1
                else np.concatenate(row, axis=ax_l))
  return rows[0] if ax_i is None or n_i == 1 else np.concatenate(rows,
                                                                  axis=ax_i)


def concat_chunks(datasets: t.Sequence, time_dim: str,
                  lead_dim: t.Optional[str] = None):
  """The chunks of a (time block x lead block) rectangle as ONE Dataset over
  the concatenated `time_dim` (and `lead_dim`) labels -- without moving data:
  every variable becomes an `xarray_lite.SlabConcat`, an index over the
  chunks' own arrays, which the fused deterministic passes read slab by slab
  through device addresses.  Coordinates that follow the two dims (valid_time,
  the 2-D `time` of a by-init truth chunk) are put together the same way.

  Returns None when the chunks do not form such a rectangle or hold data that
  cannot be addressed in place (lazy gathers, transposed or strided arrays):
  the caller then evaluates them one by one."""
  datasets = [xl.as_dataset(d) for d in datasets]
  first = datasets[0]
  if len(datasets) == 1:
    return first
  if time_dim not in first.dims:
    return None
  split = (time_dim,) + ((lead_dim,) if lead_dim and lead_dim in first.dims
                         else ())
  names = list(first.keys())
  # position of every chunk in the rectangle, by its labels
  blocks: list = [[] for _ in split]
  where = []
  for ds in datasets:
    if list(ds.keys()) != names:
      return None
    pos = []
    for j, d in enumerate(split):
      if d not in ds.coords or isinstance(ds.coords[d], xl.DataArray):
        return None
      labels = np.asarray(ds.coords[d])
      key = (labels.dtype.str, labels.tobytes())
      known = [k for k, _ in blocks[j]]
      if key not in known:
        blocks[j].append((key, labels))
        known.append(key)
      pos.append(known.index(key))
    where.append(tuple(pos) if len(pos) == 2 else (pos[0], 0))
  n_i = len(blocks[0])
  n_l = len(blocks[1]) if len(split) == 2 else 1
  if len(set(where)) != len(where) or len(where) != n_i * n_l:
    return None
  for j in range(len(split)):  # blocks must not share labels
    labels = np.concatenate([lab for _, lab in blocks[j]])
    if len(set(labels.tolist())) != len(labels):
      return None
  # everything else must be common to the chunks
  for ds in datasets[1:]:
    for k, c in first.coords.items():
      cdims = tuple(c.dims) if isinstance(c, xl.DataArray) else (k,)
      if any(d in split for d in cdims):
        continue
      other = ds.coords.get(k)
      if other is None or not np.array_equal(
          np.asarray(c.values if isinstance(c, xl.DataArray) else c),
          np.asarray(other.values if isinstance(other, xl.DataArray)
                     else other)):
        return None
  coords = {}
  for k, c in first.coords.items():
    cdims = tuple(c.dims) if isinstance(c, xl.DataArray) else (k,)
    if not any(d in split for d in cdims):
      coords[k] = c
      continue
    if not isinstance(c, xl.DataArray):  # the split dims' own labels
      j = split.index(k)
      coords[k] = np.concatenate([lab for _, lab in blocks[j]])
      continue
    cells = {}
    for ds, at in zip(datasets, where):
      other = ds.coords.get(k)
      if not isinstance(other, xl.DataArray) or tuple(other.dims) != cdims:
        return None
      cells[at] = np.asarray(other.values)
    ax_i = cdims.index(time_dim) if time_dim in cdims else None
    ax_l = cdims.index(split[1]) if len(split) == 2 and split[1] in cdims else (
        None)
    if ax_i is None:  # follows the lead dim only: the first time block's
      cells = {(0, bl): cells[(0, bl)] for bl in range(n_l)}
    if ax_l is None:
      cells = {(bi, 0): cells[(bi, 0)] for bi in range(n_i if ax_i is not None
                                                       else 1)}
    coords[k] = xl.DataArray(
        _block_matrix(cells, n_i if ax_i is not None else 1,
                      n_l if ax_l is not None else 1, ax_i, ax_l), cdims)
  out = xl.Dataset(coords=coords, attrs=dict(first.attrs))
  for name in names:
    ref = first[name]
    if not any(d in ref.dims for d in split):
      out.data_vars[name] = xl.DataArray(ref.data, ref.dims, coords, name)
      continue
    if any(d in ref.dims[-2:] for d in split):
      return None
    ax_i = ref.dims.index(time_dim) if time_dim in ref.dims else None
    ax_l = (ref.dims.index(split[1])
            if len(split) == 2 and split[1] in ref.dims else None)
    bases, cells, offset = [], {}, 0
    kind = type(ref.data)
    for ds, at in zip(datasets, where):
      da = ds[name]
      if (da.dims != ref.dims or not _plain_slabs(da)
          or type(da.data) is not kind or da.dtype != ref.dtype
          or tuple(da.shape[-2:]) != tuple(ref.shape[-2:])):
        return None
      at = (at[0] if ax_i is not None else 0, at[1] if ax_l is not None else 0)
      if at in cells:  # the variable does not follow one of the split dims
        continue
      outer = tuple(da.shape[:-2])
      n = int(np.prod(outer, dtype=np.int64))
      cells[at] = offset + np.arange(n, dtype=np.int64).reshape(outer)
      bases.append(da.data)
      offset += n
    index = _block_matrix(cells, n_i if ax_i is not None else 1,
                          n_l if ax_l is not None else 1, ax_i, ax_l)
    out.data_vars[name] = xl.DataArray(xl.SlabConcat(bases, index), ref.dims,
                                       coords, name)
  return out


def _batches(pairs: list, time_dim: str, lead_dim: t.Optional[str]):
  """Splits a window of (forecast, truth) chunks into the largest pieces that
  `concat_chunks` accepts: the whole window if it is a rectangle, else one
  piece per lead block, else the chunks themselves."""
  def joined(group):
    f = concat_chunks([p[0] for p in group], time_dim, lead_dim)
    t_ = concat_chunks([p[1] for p in group], time_dim, lead_dim)
    return None if f is None or t_ is None else (f, t_)

  whole = joined(pairs) if len(pairs) > 1 else None
  if whole is not None:
    return [whole]
  if len(pairs) > 1 and lead_dim is not None:
    by_lead: dict = {}
    for p in pairs:
      c = p[0].coords.get(lead_dim)
      if c is None or isinstance(c, xl.DataArray):
        return [(p[0], p[1]) for p in pairs]
      labels = np.asarray(c)
      by_lead.setdefault((labels.dtype.str, labels.tobytes()), []).append(p)
    if len(by_lead) > 1:
      out = []
      for group in by_lead.values():
        one = joined(group) if len(group) > 1 else None
        out += [one] if one is not None else [(p[0], p[1]) for p in group]
      return out
  return [(p[0], p[1]) for p in pairs]


# K1 chunking of evaluate_chunks (pinned: the result must not depend on how
# many chunks share a launch); 32 rows is the measured optimum of launches of
# 100+ slabs (profiles/r01_rows_per_chunk.md)
EVALUATE_ROWS_PER_CHUNK = 32


def evaluate_chunks(
    chunks: t.Sequence[tuple],
    eval_config: config.Eval,
    skipna: bool = False,
    device=None,
    prefetch: int = 2,
    *,
    truth=None,
    climatology=None,
    by_init: bool = True,
    batch_chunks: int = 1,
) -> xl.Dataset:
  """Evaluates (forecast, truth) chunks and returns the temporal mean.

  The baseline switches of `eval_config` are honoured the way the Beam pipeline
  does (evaluation.py:677-733): with `evaluate_climatology` /
  `evaluate_probabilistic_climatology` / `evaluate_persistence` the forecast of
  every chunk is replaced by a gather from `climatology` / from the years of
  `truth` / from `truth` at the init time (the full datasets, passed here; made
  resident they are read in place) -- a switch that is set without its dataset
  raises instead of silently evaluating the forecast.

  `chunks` is the full, ordered list (or any indexable) of chunk pairs; each
  rank of the current torch.distributed group (if any) evaluates a contiguous
  shard, like Beam's workers do for `input_chunks=init_time=1,lead_time=1`
  (docs/source/official-evaluation.md:537-549), and the shards meet in one
  all-reduce.  Chunks may split the lead dim as well as the time dim (that
  configuration does): results accumulate per lead label (`RunningMean`).

  `batch_chunks` = k evaluates k consecutive chunks in ONE pass of the metric x
  region loop: they are concatenated without copying (`concat_chunks`: a
  (time x lead) rectangle of chunks becomes one Dataset whose variables index
  the chunks' own arrays), so one fused launch reads every variable of all k
  chunks and the host work of the loop is paid once per k chunks.  The K1
  chunking is pinned for the whole call: the result is bit-identical for every
  `batch_chunks` (windows that do not form a rectangle fall back to smaller
  pieces, down to single chunks).

  `chunks[i]` is where a lazy sequence does its IO (the reference reads its
  chunks on a thread pool around the same workers, evaluation.py:696-697):
  with `prefetch` > 0 the next `prefetch` items are fetched by a background
  thread while the GPU works on chunk i.  Evaluation order, and therefore the
  result, does not depend on it; an exception raised by a fetch surfaces at the
  chunk it belongs to.
  """
  import torch.distributed as dist
  world, rank = 1, 0
  if dist.is_available() and dist.is_initialized():
    world, rank = dist.get_world_size(), dist.get_rank()
  if len(chunks) < world:
    raise ValueError(f'{len(chunks)} chunks cannot be sharded over {world} '
                     'ranks (every rank must take part in the all-reduce)')
  lo, hi = shard_bounds(len(chunks), world, rank)
  substitute = _chunk_substitution(eval_config, truth, climatology, by_init)
  batch_chunks = max(1, int(batch_chunks))
  mean: t.Optional[RunningMean] = None
  window: list = []

  def flush():
    nonlocal mean
    if not window:
      return
    first = window[0][0]
    time_dim = 'time' if 'time' in first.dims else 'init_time'
    lead_dim = _lead_dim(first)
    lead_dim = lead_dim if lead_dim in first.dims else None
    if mean is None:
      mean = RunningMean(time_dim, skipna, device, split_dim=lead_dim)
    for forecast, truth_chunk in _batches(window, time_dim, lead_dim):
      mean.add(_metric_and_region_loop(forecast, truth_chunk, eval_config,
                                       skipna, compute_chunk=True))
    window.clear()

  with metrics_lib.pinned_rows_per_chunk(EVALUATE_ROWS_PER_CHUNK):
    for forecast, truth_chunk in _prefetched(
        chunks, lo, hi, max(prefetch, batch_chunks - 1 if prefetch else 0)):
      forecast = xl.as_dataset(forecast)
      if substitute is not None:
        forecast = xl.as_dataset(substitute(forecast, truth_chunk))
      window.append((forecast, xl.as_dataset(truth_chunk)))
      if len(window) >= batch_chunks:
        flush()
    flush()
  assert mean is not None
  return mean.result()


# ---------------------------------------------------------------------------
# Baseline substitutions (evaluation.py:165-193, 452-472, 618-675;
# utils.py:47-70): label work on the host, the data stays where it is
# ---------------------------------------------------------------------------
def _index_values(ds: xl.Dataset, name: str) -> np.ndarray:
  c = ds.coords[name]
  return np.asarray(c.values if isinstance(c, xl.DataArray) else c)


def _positions(have: np.ndarray, want: np.ndarray, what: str) -> np.ndarray:
  """Positions of the labels `want` in the index `have` (KeyError like .sel)."""
  pos = {v: i for i, v in enumerate(np.asarray(have).tolist())}
  try:
    flat = [pos[v] for v in np.asarray(want).ravel().tolist()]
  except KeyError as e:
    raise KeyError(f'not all values found in index {what!r}: {e}') from e
  return np.array(flat, dtype=np.int64).reshape(np.shape(want))


def _slab_source(da: xl.DataArray):
  """(base with the two spatial dims last and C-contiguous, its outer dims,
  index of an existing gather or None, slab dims) of a variable."""
  import torch
  spatial = tuple(d for d in da.dims if d in ('latitude', 'longitude'))
  if len(spatial) != 2:
    raise ValueError(f'{da.name}: needs latitude and longitude, has {da.dims}')
  data, dims = da.data, tuple(da.dims)
  if dims[-2:] != spatial:
    moved = da.transpose(*[d for d in dims if d not in spatial], *spatial)
    data, dims = moved.data, tuple(moved.dims)
  if isinstance(data, xl.SlabGather):
    return data.base, dims[:-2], data.index, spatial
  if isinstance(data, torch.Tensor):
    data = data if data.is_contiguous() else data.contiguous()
  else:
    data = np.ascontiguousarray(data)
  return data, dims[:-2], None, spatial


def _gather_dataset(source: xl.Dataset, names, selectors: dict, new_dims: tuple,
                    new_shape: tuple, coords: dict) -> xl.Dataset:
  """xarray's vectorised `source[names].sel/isel({dim: indexer})` where every
  indexer has the dims `new_dims`: `selectors[dim]` holds POSITIONS along `dim`
  (-1 = no such label: the hole of an outer join, NaN).  The result's dims
  follow xarray's rule (Variable._broadcast_indexes_vectorized): walk the
  variable's dims in order, an indexed dim contributes `new_dims` (once), any
  other dim itself.  No data moves: every variable comes back as a SlabGather
  over the source array."""
  out_coords = {k: v for k, v in source.coords.items()
                if k not in selectors and not (
                    isinstance(v, xl.DataArray)
                    and any(d in selectors for d in v.dims))}
  out_coords.update(coords)
  out = xl.Dataset(coords=out_coords, attrs=dict(source.attrs))
  for name in names:
    da = source[name]
    base, outer, prior, spatial = _slab_source(da)
    missing = [d for d in selectors if d not in outer]
    if missing:
      raise ValueError(f'{name}: dims {missing} to select are not in {da.dims}')
    sizes = dict(zip(outer, (prior.shape if prior is not None
                             else base.shape[:-2])))
    out_dims: list = []
    for d in outer:
      for nd in (new_dims if d in selectors else (d,)):
        if nd not in out_dims:
          out_dims.append(nd)
    out_shape = tuple(new_shape[new_dims.index(d)] if d in new_dims
                      else sizes[d] for d in out_dims)
    index = np.zeros(out_shape, dtype=np.int64)
    hole = np.zeros(out_shape, dtype=bool)
    stride = 1
    for d in reversed(outer):
      if d in selectors:
        pos = np.asarray(selectors[d], dtype=np.int64)
        shape = [1] * len(out_dims)
        for nd, n in zip(new_dims, new_shape):
          shape[out_dims.index(nd)] = n
        order = [nd for nd in out_dims if nd in new_dims]
        pos = np.transpose(pos, [new_dims.index(nd) for nd in order]
                           ).reshape(shape)
        hole = hole | (pos < 0)
        index = index + np.maximum(pos, 0) * stride
      else:
        shape = [1] * len(out_dims)
        shape[out_dims.index(d)] = sizes[d]
        index = index + (np.arange(sizes[d], dtype=np.int64) * stride
                         ).reshape(shape)
      stride *= sizes[d]
    if prior is not None:  # a gather of a gather: compose the tables
      index = prior.ravel()[index]
    index = np.where(hole, -1, index)
    out.data_vars[name] = xl.DataArray(
        xl.SlabGather(base, index), tuple(out_dims) + spatial, out_coords, name)
  return out


def _when(forecast: xl.Dataset, time_dim: str):
  """(datetime64 values, dims, coords that travel with the indexer) of
  `forecast[time_dim]`."""
  c = forecast.coords[time_dim]
  if isinstance(c, xl.DataArray):
    values, dims = np.asarray(c.values), tuple(c.dims)
  else:
    values, dims = np.asarray(c), (time_dim,)
  carried = {}
  for k, v in forecast.coords.items():
    vdims = tuple(v.dims) if isinstance(v, xl.DataArray) else (k,)
    if all(d in dims for d in vdims) and (isinstance(v, xl.DataArray)
                                          or k in dims):
      carried[k] = v
  return values, dims, carried


def _dayofyear_hour(values: np.ndarray):
  import pandas as pd
  idx = pd.DatetimeIndex(np.asarray(values).ravel())
  shape = np.shape(values)
  return (np.asarray(idx.dayofyear).reshape(shape),
          np.asarray(idx.hour).reshape(shape))


def _climatology_variables(climatology: xl.Dataset, variables) -> dict:
  """{forecast variable: climatology variable}: by name, else `<name>_mean`
  (evaluation.py:633-639)."""
  variables = list(variables)
  if all(v in climatology for v in variables):
    return {v: v for v in variables}
  renamed = {v: f'{v}_mean' for v in variables}
  absent = [k for k in renamed.values() if k not in climatology]
  if absent:
    raise KeyError(f'{absent} not found in the climatology')
  return renamed


def climatology_like_forecast(forecast, climatology, time_dim: str,
                              variables=None, hour_if_present: bool = False):
  """`climatology[variables].sel(dayofyear=forecast[time_dim].dt.dayofyear,
  hour=forecast[time_dim].dt.hour)` (evaluation.py:452-460; with
  `hour_if_present` the Beam version, :629-646, which selects `hour` only when
  the climatology has one and falls back to `<var>_mean` names).

  Zero-copy: every variable of the result is a SlabGather over the climatology
  array (resident in HBM after `make_resident`; a host climatology crosses
  PCIe as the distinct slabs one chunk touches)."""
  given = (forecast, climatology)
  forecast, climatology = xl.as_dataset(forecast), xl.as_dataset(climatology)
  if variables is None:
    variables = list(forecast.keys())
  values, dims, carried = _when(forecast, time_dim)
  doy, hour = _dayofyear_hour(values)
  selectors = {'dayofyear': _positions(_index_values(climatology, 'dayofyear'),
                                       doy, 'dayofyear')}
  coords = dict(carried)
  coords['dayofyear'] = xl.DataArray(doy, dims)
  if not hour_if_present or 'hour' in climatology.coords:
    selectors['hour'] = _positions(_index_values(climatology, 'hour'), hour,
                                   'hour')
    coords['hour'] = xl.DataArray(hour, dims)
  if hour_if_present:
    names = _climatology_variables(climatology, variables)
  else:
    for v in variables:
      if v not in climatology:
        raise KeyError(v)
    names = {v: v for v in variables}
  picked = _gather_dataset(climatology, list(names.values()), selectors, dims,
                           np.shape(values), coords)
  out = xl.Dataset(coords=picked.coords, attrs=picked.attrs)
  for v, cname in names.items():
    da = picked.data_vars[cname]
    out.data_vars[v] = xl.DataArray(da.data, da.dims, out.coords, v)
  return xl.like_input(out, *given)


def make_probabilistic_climatology(ds, start_year: int, end_year: int,
                                   hour_interval: int, variables=None):
  """utils.py:47-70: the years of `ds` stacked as ensemble members -- dims
  (hour, number, dayofyear, ...), day 366 (and any other missing time stamp)
  NaN.  Nothing is copied: member `number` of (hour, dayofyear) is the time
  step of `ds` with that (year, dayofyear, hour), kept as a SlabGather index
  over `ds`'s own array, -1 where the year has no such step."""
  import pandas as pd
  given = ds
  ds = xl.as_dataset(ds)
  hours = np.arange(0, 24, hour_interval)
  years = np.arange(start_year, end_year + 1)
  times = pd.DatetimeIndex(_index_values(ds, 'time'))
  t_year, t_doy, t_hour = (np.asarray(times.year), np.asarray(times.dayofyear),
                           np.asarray(times.hour))
  for year in years:  # `.sel(time=str(year))` of a year without data: KeyError
    if not (t_year == year).any():
      raise KeyError(str(year))
  used = np.isin(t_hour, hours) & np.isin(t_year, years)
  doys = np.unique(t_doy[used])  # concat's outer join: the sorted union
  where = np.full((len(hours), len(years), len(doys)), -1, dtype=np.int64)
  hi = np.searchsorted(hours, t_hour[used])
  yi = t_year[used] - start_year
  di = np.searchsorted(doys, t_doy[used])
  if len(np.unique(np.stack([hi, yi, di]), axis=1)[0]) != used.sum():
    raise ValueError('several time steps share one (year, dayofyear, hour): '
                     'the probabilistic climatology needs at most hourly data')
  where[hi, yi, di] = np.nonzero(used)[0]
  names = [k for k in (variables or ds.keys())]
  for k in names:
    if 'time' not in ds[k].dims:
      raise ValueError(f'{k} has no time dim: {ds[k].dims}')
  coords = {'hour': hours, 'number': np.arange(len(years)), 'dayofyear': doys}
  out = _gather_dataset(ds, names, {'time': where},
                        ('hour', 'number', 'dayofyear'), where.shape, coords)
  return xl.like_input(out, given)


def create_persistence_forecast(forecast, obs):
  """evaluation.py:165-193 (by-valid layout: `forecast.init_time` is a
  coordinate over (time, lead_time)): the observation at the initialisation
  time of every (time, lead_time), for the times at least the longest lead time
  after the first one.  A gather from `obs` by time label."""
  given = (forecast, obs)
  forecast, obs = xl.as_dataset(forecast), xl.as_dataset(obs)
  logging.warning('by-valid with evaluate_persistence is not 100% correct.')
  init = forecast.coords['init_time']
  if not isinstance(init, xl.DataArray) or 'time' not in init.dims:
    raise AttributeError(
        "forecast.init_time has no 'time' dim: create_persistence_forecast is "
        'for the by-valid layout (evaluation.py:184-187); by-init chunks use '
        '_persistence_like_forecast_chunk')
  init = init.transpose('time', *[d for d in init.dims if d != 'time'])
  time = _index_values(forecast, 'time')
  lead_dims = tuple(d for d in init.dims if d != 'time')
  lead_max = max(np.max(_index_values(forecast, d)) for d in lead_dims)
  keep = time >= time[0] + lead_max  # label slice(start, None), ascending time
  init_values = np.asarray(init.values)[keep]
  where = _positions(_index_values(obs, 'time'), init_values, 'time')
  coords = {'time': time[keep]}
  for d in lead_dims:
    coords[d] = _index_values(forecast, d)
  coords['init_time'] = xl.DataArray(init_values, init.dims)
  names = [k for k in obs.keys() if 'time' in obs[k].dims]
  out = _gather_dataset(obs, names, {'time': where}, tuple(init.dims),
                        init_values.shape, coords)
  for k in obs.keys():  # variables without a time dim pass through (.sel)
    if k not in names:
      out.data_vars[k] = obs[k]
  return xl.like_input(out, *given)


def _lead_dim(forecast: xl.Dataset) -> str:
  return 'lead_time' if 'lead_time' in forecast.dims or (
      'lead_time' in forecast.coords) else 'prediction_timedelta'


def _persistence_like_forecast_chunk(forecast_chunk, truth_chunk, truth,
                                     variables=None, by_init: bool = True):
  """evaluation.py:651-675: `truth.sel(time=init_time)` with the chunk's
  lead_time dim and valid_time coordinate -- every lead of an init time reads
  the SAME truth slab (a slab table with repeats, no expand / copy)."""
  if truth is None:
    raise ValueError('`truth` must not be `None`')
  if not by_init:
    raise NotImplementedError('Persistence not compatible with by-valid format.')
  given = (forecast_chunk, truth)
  forecast_chunk, truth = xl.as_dataset(forecast_chunk), xl.as_dataset(truth)
  lead_dim = _lead_dim(forecast_chunk)
  init = _index_values(forecast_chunk, 'init_time')
  lead = _index_values(forecast_chunk, lead_dim)
  pos = _positions(_index_values(truth, 'time'), init, 'time')
  where = np.broadcast_to(pos[None, :], (len(lead), len(init)))
  coords = {lead_dim: lead, 'init_time': init}
  if 'valid_time' in forecast_chunk.coords:
    coords['valid_time'] = forecast_chunk.coords['valid_time']
  names = [k for k in (variables or truth.keys()) if 'time' in truth[k].dims]
  # expand_dims puts the new dim first: (lead_time, init_time, ...)
  out = _gather_dataset(truth, names, {'time': where}, (lead_dim, 'init_time'),
                        where.shape, coords)
  return xl.like_input(out, *given), truth_chunk


def _climatology_like_forecast_chunk(forecast_chunk, truth_chunk, climatology,
                                     variables=None, by_init: bool = True):
  """evaluation.py:618-649."""
  time_dim = 'valid_time' if by_init else 'time'
  if variables is None:
    variables = list(xl.as_dataset(truth_chunk).keys())
  return (climatology_like_forecast(forecast_chunk, climatology, time_dim,
                                    variables, hour_if_present=True),
          truth_chunk)


def _evaluate_all_metrics(eval_name: str, eval_config: config.Eval,
                          data_config, skipna: bool, *, forecast, truth,
                          climatology=None) -> xl.Dataset:
  """Evaluate a set of eval metrics in memory (evaluation.py:441-483).

  The reference opens `forecast, truth, climatology` from `data_config.paths`
  and writes a NetCDF file; here they are arguments (already opened, with the
  reference's time conventions applied: init_time / lead_time / valid_time for
  by-init data) and the merged result is returned -- IO is out of scope.
  Everything between is the reference's sequence: the baseline substitutions
  selected by `eval_config` (as zero-copy gathers), `truth.sel(time=
  forecast.valid_time)` for by-init data, then the metric x region loop."""
  del eval_name  # names the output file in the reference
  given = (forecast, truth)
  forecast, truth = xl.as_dataset(forecast), xl.as_dataset(truth)
  by_init = bool(getattr(data_config, 'by_init', True))
  time_dim = 'valid_time' if by_init else 'time'
  if eval_config.evaluate_climatology:
    if climatology is None:
      raise ValueError('evaluate_climatology needs a climatology dataset')
    forecast = climatology_like_forecast(forecast, climatology, time_dim)
  if eval_config.evaluate_probabilistic_climatology:
    probabilistic_climatology = make_probabilistic_climatology(
        truth, eval_config.probabilistic_climatology_start_year,
        eval_config.probabilistic_climatology_end_year,
        eval_config.probabilistic_climatology_hour_interval,
        variables=list(forecast.keys()))
    forecast = climatology_like_forecast(forecast, probabilistic_climatology,
                                         time_dim)
  if eval_config.evaluate_persistence:
    if by_init:
      # the reference's in-memory driver raises AttributeError here (:184-187
      # expect the by-valid init_time coordinate); its Beam driver defines
      # by-init persistence (:651-675) and that is what runs
      forecast, _ = _persistence_like_forecast_chunk(
          forecast, truth, truth, list(forecast.keys()), by_init=True)
    else:
      forecast = create_persistence_forecast(forecast, truth)
  forecast = xl.as_dataset(forecast)
  if by_init:
    truth = select_truth_at_valid_time(truth, forecast,
                                       lead_dim=_lead_dim(forecast))
  results = _metric_and_region_loop(forecast, truth, eval_config, skipna=skipna)
  return xl.like_input(xl.as_dataset(results), *given)
