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
import os
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
        result = xl.as_dataset(regions_fn(forecast, truth, regions, skipna))
      elif regions is not None:
        tmp_results = []
        for region_name, region in regions.items():
          tmp_result = xl.as_dataset(eval_fn(
              forecast=forecast, truth=truth, region=region, skipna=skipna))
          tmp_results.append(tmp_result.expand_dims({'region': [region_name]}))
        result = xl.concat(tmp_results, 'region')
      else:
        result = xl.as_dataset(eval_fn(
            forecast=forecast, truth=truth, skipna=skipna))
      results.append((name, result))
  # expand_dims({'metric': name}) + xr.merge of :424-437, in one step
  return xl.like_input(xl.merge_metrics(results), given_forecast, given_truth)


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
  pos = {v: i for i, v in enumerate(xl.label_list(have))}
  try:
    index = np.array([pos[v] for v in xl.label_list(valid)], dtype=np.int64)
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


# destination tables of RunningMean: one entry per run of this many (or more)
# consecutive elements behind the split dim, one per element below
_RUN_MIN = 256
# an accumulator with a split dim starts with at most this many bytes of rows
_FIRST_ROWS_BYTES = 256 << 20
# results with this many elements per row or more (maps) keep no count map
# when NaNs are not skipped (every element has its row's number of steps)
_LAZY_COUNT_MIN = 1 << 20


class _Accumulator:
  """(sum, count) of one result variable.  Without a split dim: tensors of the
  result's shape (minus the averaged dim).  With one: [row, *rest], one row per
  label of the split dim in order of first appearance, the rest dims in order."""

  def __init__(self, dims, shape, split, device, lazy_count: bool = False):
    import torch
    self.dims, self.shape, self.split = dims, shape, split
    self.labels: list = []     # label arrays (0-d), row order
    self.row_of: dict = {}     # label value -> row
    self.dst: dict = {}        # label-vector key -> device int64 table
    # time steps counted on the host (map_suite.py: without skipna every
    # element of a row gains the same count -- no pass over the count maps per
    # chunk): row (None without a split dim) -> steps not yet in `count`
    self.pending: dict = {}
    if split is None:
      self.rest_shape = shape
      alloc = shape
    else:
      self.pos = dims.index(split)
      self.rest_shape = shape[:self.pos] + shape[self.pos + 1:]
      # rows are added by doubling: 8 to start with, fewer when a row is big
      # (map-valued results: a row of `deterministic_spatial` is 1.3 GB)
      row_bytes = 8 * int(np.prod(self.rest_shape, dtype=np.int64))
      alloc = (int(max(1, min(8, _FIRST_ROWS_BYTES // max(row_bytes, 1)))),
               ) + self.rest_shape
    self.total = torch.zeros(alloc, dtype=torch.float64, device=device)
    # `lazy_count` (map-valued results without skipna: every element of a row
    # has the row's number of time steps): no count MAP unless somebody asks
    # for one -- the steps stay in `pending`, result() divides by them
    self._count = None if lazy_count else torch.zeros_like(self.total)

  @property
  def count(self):
    import torch
    if self._count is None:
      self._count = torch.zeros_like(self.total)
      self.settle()
    return self._count

  @count.setter
  def count(self, value):
    self._count = value

  def steps(self, rows) -> np.ndarray:
    """Host-side time steps of `rows` (None entries: rows that never came)."""
    return np.array([0.0 if r is None else float(self.pending.get(r, 0))
                     for r in rows])

  def rows(self, labels: np.ndarray) -> np.ndarray:
    import torch
    if len(set(labels.tolist())) != len(labels):
      # two entries of one chunk would share an accumulator row: the scatter
      # kernel's read-modify-write per element would lose one of them
      raise ValueError(f'repeated {self.split} labels in one chunk result: '
                       f'{labels}')
    out = np.empty(len(labels), dtype=np.int64)
    for j, (value, label) in enumerate(zip(labels.tolist(), labels)):
      row = self.row_of.get(value)
      if row is None:
        row = self.row_of[value] = len(self.labels)
        self.labels.append(label)
      out[j] = row
    while len(self.labels) > self.total.shape[0]:
      grow = torch.zeros_like(self.total)
      self.total = torch.cat([self.total, grow])
      if self._count is not None:
        self._count = torch.cat([self._count, torch.zeros_like(grow)])
    return out

  def settle(self):
    """Brings the host-side step counts into `count` (if there is one)."""
    if self._count is None:
      return
    for row, steps in self.pending.items():
      if row is None:
        self.count += float(steps)
      else:
        self.count[row] += float(steps)
    self.pending = {}

  def destination_runs(self, rows: np.ndarray) -> tuple:
    """(first accumulator element of every RUN of result elements, run
    length): the elements behind the split dim are consecutive in the result
    and in the accumulator -- a map-valued result needs one entry per slab."""
    block = int(np.prod(self.rest_shape, dtype=np.int64))
    run = int(np.prod(self.shape[self.pos + 1:], dtype=np.int64))
    n_outer = int(np.prod(self.shape[:self.pos], dtype=np.int64))
    dst = (rows[None, :] * block +
           np.arange(n_outer, dtype=np.int64)[:, None] * run).ravel()
    if dst.size and (dst.min() < 0 or dst.max() + run > self.total.numel()):
      raise ValueError('accumulator destination out of range')  # (host check)
    return dst, run

  def destinations(self, rows: np.ndarray) -> np.ndarray:
    """Accumulator element of every element of a result of `self.shape`."""
    block = int(np.prod(self.rest_shape, dtype=np.int64))
    rest = np.arange(block, dtype=np.int64).reshape(self.rest_shape)
    where = [1] * len(self.shape)
    where[self.pos] = len(rows)
    dst = (rows.reshape(where) * block + np.expand_dims(rest, self.pos)).ravel()
    if dst.size and (dst.min() < 0 or dst.max() >= self.total.numel()):
      raise ValueError('accumulator destination out of range')  # (host check)
    return dst


class RunningMean:
  """xbeam.Mean's (sum, count) combiner, kept on the device.

  add(chunk_result) accumulates `sum` and `count` over `dim` (NaNs add to
  neither when skipna); result() all-reduces both across the process group (if
  one is initialised) and divides.  Sums continue value by value in the order
  the time steps arrive (wb2_time_accumulate): the result does not depend on
  how many of them one add() brings.

  Chunks may split another dim as well -- the official 0.25-degree runs use
  `input_chunks=init_time=1,lead_time=1` (docs/source/official-evaluation.md:
  537-549) --: xbeam.Mean combines per key of the remaining chunk offsets, so
  with `split_dim` (the forecast's lead dim) every label of that dim has its
  own accumulator row, whatever mix of labels a chunk result carries, and
  result() lays the rows out in label order.
  """

  def __init__(self, dim: str, skipna: bool = False, device=None, comm=None,
               split_dim: t.Optional[str] = None, split_labels=None,
               split_order: str = 'sorted'):
    self.dim = dim
    self.skipna = skipna
    self.device = device
    self.split_dim = split_dim
    # the FULL list of `split_dim` labels, in output order: required with an
    # RCCL `comm` (which can exchange numbers, not label lists), optional
    # otherwise (the ranks' label sets are then united, first seen first)
    self.split_labels = (None if split_labels is None
                         else np.asarray(split_labels))
    # without the full list: 'sorted' (whatever order the chunks arrive in) or
    # 'first_seen' (rank 0's labels first -- the dataset's own order when the
    # chunk list is walked in order, like xbeam.Mean's output, also for lead
    # coordinates that are not monotonic; evaluate_chunks uses it)
    if split_order not in ('sorted', 'first_seen'):
      raise ValueError(f'split_order={split_order!r}')
    self.split_order = split_order
    # an RCCL communicator from engine.comm_init_rank: the exchange then goes
    # through the C ABI (wb2_time_mean_allreduce) instead of torch.distributed
    self.comm = comm
    self._acc: dict = {}     # var -> _Accumulator
    self._coords: dict = {}

  def _on_gpu(self) -> bool:
    import torch
    return self.device is not None and torch.device(self.device).type == 'cuda'

  def _on_gpu_or_unset(self) -> bool:
    """(the device is adopted from the first device-resident result)"""
    return self.device is None or self._on_gpu()

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
      labels = chunk.coords.get(self.split_dim) if self.split_dim else None
      split = self.split_dim if (
          labels is not None and not isinstance(labels, xl.DataArray)
          and self.split_dim in dims) else None
      raw = da.data
      if isinstance(raw, torch.Tensor):
        # map-valued results (Spatial* metrics, rank histograms) already live
        # on the device: accumulate there, no host round trip
        if self.device is None and raw.is_cuda:
          self.device = raw.device
        engine.order_read(raw)  # produced on another thread's stream?
        # float32 results go to the accumulate kernel as they are (widened
        # there, exactly); only the host fallback below needs float64
        values = raw if raw.dtype in (torch.float32, torch.float64) and (
            self._on_gpu()) else raw.to(torch.float64)
      else:
        values = torch.as_tensor(np.ascontiguousarray(da.values),
                                 dtype=torch.float64)
      acc = self._acc.get(name)
      if acc is None:
        row = int(np.prod(shape, dtype=np.int64)) // (
            1 if split is None else max(shape[dims.index(split)], 1))
        lazy = (not self.skipna and self.comm is None and self._on_gpu()
                and isinstance(raw, torch.Tensor) and row >= _LAZY_COUNT_MIN)
        acc = self._acc[name] = _Accumulator(dims, shape, split, self.device,
                                             lazy_count=lazy)
      if acc.pending:
        acc.settle()
      if acc.dims != dims or acc.split != split or (
          acc.rest_shape != (shape if split is None else
                             shape[:acc.pos] + shape[acc.pos + 1:])):
        raise ValueError(f'{name}: chunk layout changed {acc.dims}'
                         f'{acc.shape} -> {dims}{shape}')
      rows = None
      if split is not None:
        labels = np.asarray(labels)
        rows = acc.rows(labels)
      if self._on_gpu():
        dst, run = None, 1
        if rows is not None:
          key = (shape, rows.tobytes())
          hit = acc.dst.get(key)
          if hit is None:
            acc.shape = shape
            # long runs behind the split dim (maps): one entry per run
            inner = int(np.prod(shape[acc.pos + 1:], dtype=np.int64))
            table, run = (acc.destination_runs(rows) if inner >= _RUN_MIN
                          else (acc.destinations(rows), 1))
            hit = acc.dst[key] = (engine.upload_table(table, self.device), run)
          dst, run = hit
        if acc._count is None:  # (no count map: the row's steps, on the host)
          engine.time_accumulate(values.to(self.device).contiguous(), axis,
                                 False, acc.total, None, dst, run)
          for r in ([None] if rows is None else rows.tolist()):
            acc.pending[r] = acc.pending.get(r, 0) + da.shape[axis]
        else:
          engine.time_accumulate(values.to(self.device).contiguous(), axis,
                                 self.skipna, acc.total, acc.count, dst, run)
      else:  # host accumulators (CPU tests of the sharding logic)
        values = values.cpu()
        ok = ~torch.isnan(values) if self.skipna else torch.ones_like(
            values, dtype=torch.bool)
        s = torch.where(ok, values, torch.zeros_like(values)).sum(axis)
        c = ok.to(torch.float64).sum(axis)
        if rows is None:
          acc.total += s
          acc.count += c
        else:
          idx = torch.as_tensor(rows)
          acc.total.index_add_(0, idx, s.movedim(acc.pos, 0))
          acc.count.index_add_(0, idx, c.movedim(acc.pos, 0))

  def _split_labels(self, names) -> dict:
    """{var: labels in output order}: the caller's full list if given, else
    the union over the ranks, sorted or in order of first appearance
    (`split_order`)."""
    import torch.distributed as dist
    mine = {n: np.array(self._acc[n].labels) for n in names
            if self._acc[n].split is not None}
    if self.split_labels is not None:
      for n, have in mine.items():
        extra = set(have.tolist()) - set(self.split_labels.tolist())
        if extra:
          raise ValueError(f'{n}: {self.split_dim} labels {sorted(extra)} are '
                           'not in the split_labels given')
      return {n: self.split_labels for n in mine}
    if self.comm is not None and mine:
      # every rank must lay its rows out alike and an RCCL communicator cannot
      # tell the ranks each other's labels
      raise ValueError(
          f'RunningMean(comm=..., split_dim={self.split_dim!r}) needs '
          'split_labels: the full list of labels, the same on every rank')

    def first_seen(parts):
      if self.split_order == 'sorted':
        return np.unique(np.concatenate(parts))
      seen, order = set(), []
      for part in parts:
        for value, label in zip(part.tolist(), part):
          if value not in seen:
            seen.add(value)
            order.append(label)
      return np.array(order, dtype=parts[0].dtype) if order else parts[0][:0]
    out = {}
    if dist.is_available() and dist.is_initialized() and (
        dist.get_world_size() > 1):
      everyone: list = [None] * dist.get_world_size()
      dist.all_gather_object(everyone, mine)
      for n in mine:
        out[n] = first_seen([e[n] for e in everyone if n in e])
    else:
      out = {n: first_seen([v]) for n, v in mine.items()}
    return out

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
    names = sorted(self._acc)
    for n in names:
      self._acc[n].settle()
    labels = self._split_labels(names)
    # (sum, count) in output layout: rows in label order, labels a rank never
    # met (lead-major chunk lists) as zeros
    def laid_out(n):
      acc = self._acc[n]
      # without a count map (and nothing to exchange) the divisor is the
      # row's number of steps, broadcast
      lazy = acc._count is None and not in_group
      if acc.split is None:
        if lazy:
          return acc.total, torch.as_tensor(acc.steps([None])[0],
                                            dtype=torch.float64,
                                            device=acc.total.device)
        return acc.total, acc.count
      rows = [acc.row_of.get(v) for v in labels[n].tolist()]
      steps = None
      if lazy:
        steps = torch.as_tensor(acc.steps(rows), dtype=torch.float64,
                                device=acc.total.device).reshape(
                                    (-1,) + (1,) * len(acc.rest_shape))
      if rows == list(range(len(rows))):
        # the rows as they lie (the usual case): views, no copy -- the maps of
        # `deterministic_spatial` are gigabytes per variable
        return acc.total[:len(rows)], (
            steps if lazy else acc.count[:len(rows)])
      pad = acc.total.shape[0]
      idx = torch.as_tensor([pad if r is None else r for r in rows],
                            device=acc.total.device)
      zero = torch.zeros((1,) + acc.rest_shape, dtype=torch.float64,
                         device=acc.total.device)
      picked = [torch.cat([x, zero]).index_select(0, idx)
                for x in ((acc.total,) if lazy else (acc.total, acc.count))]
      return (picked[0], steps) if lazy else tuple(picked)
    # without a process group nothing is exchanged: one variable at a time
    # (its mean leaves for the host before the next one is laid out)
    sums = {n: laid_out(n) for n in names} if in_group else {}
    if self.comm is not None and names:
      from weatherbench2_amd import engine
      flat_t = torch.cat([sums[n][0].reshape(-1) for n in names])
      flat_c = torch.cat([sums[n][1].reshape(-1) for n in names])
      engine.time_mean_allreduce(flat_t, flat_c, self.comm)
      offset = 0
      for n in names:
        shape, size = sums[n][0].shape, sums[n][0].numel()
        sums[n] = (flat_t[offset:offset + size].reshape(shape),
                   flat_c[offset:offset + size].reshape(shape))
        offset += size
    elif in_torch_group and names:
      flat = torch.cat([torch.stack(sums[n]).reshape(-1) for n in names])
      dist.all_reduce(flat)  # the path's only exchange step
      offset = 0
      for n in names:
        shape, size = sums[n][0].shape, sums[n][0].numel()
        sums[n] = (flat[offset:offset + size].reshape(shape),
                   flat[offset + size:offset + 2 * size].reshape(shape))
        offset += 2 * size
    coords = dict(self._coords)
    split_labels = None
    out_vars = {}
    from weatherbench2_amd import feeder
    for n in names:
      acc = self._acc[n]
      total, count = sums[n] if in_group else laid_out(n)
      # 0/0 -> NaN like an empty mean; maps leave through the pinned ring
      mean = feeder.download(total / count)
      del total, count
      if acc.split is not None:
        mean = np.moveaxis(mean, 0, acc.pos)
        if split_labels is not None and not np.array_equal(split_labels,
                                                           labels[n]):
          raise ValueError(f'{n}: {self.split_dim} labels differ between '
                           'variables')
        split_labels = labels[n]
      out_vars[n] = (acc.dims, mean)
    if split_labels is not None:
      coords[self.split_dim] = split_labels
    out = xl.Dataset(coords=coords)
    for n, (dims, mean) in out_vars.items():
      out.data_vars[n] = xl.DataArray(mean, dims, coords, n)
    return out


class _KeptRows:
  """Storage of one result variable of `RunningConcat`: [row, *rest], the rows
  are those of the sink's registry for the variable's key dims (one per (time
  label, split label) pair in order of first appearance)."""

  def __init__(self, dims, shape, key_dims, device):
    import torch
    self.dims, self.key_dims = tuple(dims), tuple(key_dims)
    self.key_pos = [self.dims.index(d) for d in self.key_dims]
    self.rest_dims = tuple(d for d in self.dims if d not in self.key_dims)
    self.rest_shape = tuple(n for d, n in zip(dims, shape)
                            if d not in self.key_dims)
    self.filled: set = set()   # rows this variable has a value for
    self.total = torch.zeros((64,) + self.rest_shape, dtype=torch.float64,
                             device=device)
    self.count = torch.zeros_like(self.total)

  def claim(self, rows: list, n_rows: int, name: str):
    """`rows` are about to be written: none may hold a value yet; the storage
    doubles until it has `n_rows` rows."""
    import torch
    if not self.filled.isdisjoint(rows) or len(set(rows)) != len(rows):
      raise ValueError(f'{name}: a ({", ".join(self.key_dims)}) combination '
                       'came twice')
    self.filled.update(rows)
    while n_rows > self.total.shape[0]:
      self.total = torch.cat([self.total, torch.zeros_like(self.total)])
      self.count = torch.cat([self.count, torch.zeros_like(self.count)])


class RunningConcat:
  """`temporal_mean=False` (config.py:55; the `deterministic_temporal` config of
  scripts/evaluate.py:479-487): the per-chunk results are KEPT along the time
  dim instead of averaged -- what the Beam pipeline writes chunk by chunk when
  it skips `TemporalMean` (evaluation.py:735-752).  Same interface as
  `RunningMean`: add(chunk_result) files every (time, lead) slice of a chunk
  result under its labels (on the device, no host round trip), result() puts
  the slices together in order of first appearance (the dataset's order when
  the chunk list is walked in order; NaN where a combination never came).
  Chunk programs feed it through `wb2_gather_accumulate` like the mean (each
  destination gets exactly one value: 0 + v, exact)."""

  keeps_time = True

  def __init__(self, dim: str, device=None, split_dim: t.Optional[str] = None):
    self.dim, self.device, self.split_dim = dim, device, split_dim
    self.skipna = False
    self._acc: dict = {}      # var -> _KeptRows
    self._rows: dict = {}     # key dims -> {label combination: row}
    self._coords: dict = {}
    self._dtypes: dict = {}
    self._seen: dict = {}     # key dim -> labels in order of first appearance

  def _on_gpu(self) -> bool:
    import torch
    return self.device is not None and torch.device(self.device).type == 'cuda'

  def _on_gpu_or_unset(self) -> bool:
    return self.device is None or self._on_gpu()

  def key_dims(self, dims) -> tuple:
    return tuple(d for d in (self.dim, self.split_dim) if d and d in dims)

  def rows(self, chunk: xl.Dataset, key_dims) -> np.ndarray:
    """Rows of the chunk's label combinations along `key_dims` (C order), new
    combinations appended to the registry (shared by every variable with these
    key dims); the labels are remembered in first-seen order."""
    import itertools
    lists = []
    for d in key_dims:
      labels = chunk.coords.get(d)
      if labels is None or isinstance(labels, xl.DataArray):
        raise ValueError(f'chunk result has no {d!r} labels')
      labels = np.asarray(labels)
      values = labels.tolist()
      seen = self._seen.setdefault(d, {})
      for value, label in zip(values, labels):
        if value not in seen:
          seen[value] = label
      lists.append(values)
    row_of = self._rows.setdefault(tuple(key_dims), {})
    out = np.empty([len(x) for x in lists], dtype=np.int64)
    flat = out.reshape(-1)
    for i, key in enumerate(itertools.product(*lists)):
      row = row_of.get(key)
      if row is None:
        row = row_of[key] = len(row_of)
      flat[i] = row
    return out

  def storage(self, name, dims, shape, dtype) -> _KeptRows:
    acc = self._acc.get(name)
    if acc is None:
      acc = self._acc[name] = _KeptRows(dims, shape, self.key_dims(dims),
                                        self.device)
      self._dtypes[name] = dtype
    if acc.dims != tuple(dims) or acc.rest_shape != tuple(
        n for d, n in zip(dims, shape) if d not in acc.key_dims):
      raise ValueError(f'{name}: chunk layout changed {acc.dims} -> {dims}')
    return acc

  def snapshot(self):
    return ({n: (a.total.clone(), a.count.clone(), set(a.filled))
             for n, a in self._acc.items()},
            {k: dict(v) for k, v in self._rows.items()},
            {k: dict(v) for k, v in self._seen.items()})

  def restore(self, state):
    accs, rows, seen = state
    for n, (total, count, filled) in accs.items():
      a = self._acc[n]
      a.total, a.count, a.filled = total.clone(), count.clone(), set(filled)
    self._rows = {k: dict(v) for k, v in rows.items()}
    self._seen = {k: dict(v) for k, v in seen.items()}

  def add(self, chunk: xl.Dataset):
    import torch
    from weatherbench2_amd import engine
    for k, c in chunk.coords.items():
      if k not in (self.dim, self.split_dim) and not (
          isinstance(c, xl.DataArray) and any(
              d in (self.dim, self.split_dim) for d in c.dims)):
        self._coords.setdefault(k, c)
    rows_of_group: dict = {}
    for name, da in chunk.data_vars.items():
      if self.dim not in da.dims:
        raise ValueError(f'{name} has no {self.dim!r} dim: {da.dims}')
      raw = da.data
      if isinstance(raw, torch.Tensor):
        if self.device is None and raw.is_cuda:
          self.device = raw.device
        engine.order_read(raw)
        values = raw
        np_dtype = np.dtype(str(raw.dtype).replace('torch.', ''))
      else:
        values = torch.as_tensor(np.ascontiguousarray(da.values))
        np_dtype = np.asarray(da.values).dtype
      acc = self.storage(name, da.dims, da.shape, np_dtype)
      rows = rows_of_group.get(acc.key_dims)
      if rows is None:
        rows = rows_of_group[acc.key_dims] = self.rows(
            chunk, acc.key_dims).ravel()
      acc.claim(rows.tolist(), len(self._rows[acc.key_dims]), name)
      moved = values.to(acc.total.device, torch.float64).movedim(
          acc.key_pos, list(range(len(acc.key_pos))))
      moved = moved.reshape((rows.size,) + acc.rest_shape)
      index = torch.as_tensor(rows, device=acc.total.device)
      acc.total.index_copy_(0, index, moved)
      acc.count.index_fill_(0, index, 1.0)

  def result(self) -> xl.Dataset:
    import torch
    import torch.distributed as dist
    several = dist.is_available() and dist.is_initialized() and (
        dist.get_world_size() > 1)

    def final_dtype(name):
      try:
        final = torch.from_numpy(
            np.empty(0, dtype=np.dtype(self._dtypes[name]))).dtype
        return final if final.is_floating_point else None
      except TypeError:
        return None
    pieces = {}
    for name, acc in self._acc.items():
      keys = list(self._rows[acc.key_dims])
      rows = sorted(acc.filled)
      if rows:
        # rows 0 .. n - 1 are a slice; the values leave the device in the
        # dtype of the result (the same rounding as numpy's astype, half the
        # bytes for float32 results).  One process: they stay on the device
        # until they are in place (below).
        values = (acc.total[:len(rows)] if rows[-1] == len(rows) - 1
                  else acc.total[rows])
        if several or not values.is_cuda:
          final = final_dtype(name)
          if final is not None and final != values.dtype:
            values = values.to(final)
          values = values.cpu().numpy()
      else:
        values = np.zeros((0,) + acc.rest_shape)
      pieces[name] = (acc.dims, acc.key_dims, acc.rest_shape,
                      [keys[r] for r in rows], values, self._dtypes[name])
    seen = {d: list(v.values()) for d, v in self._seen.items()}
    everyone = [(pieces, seen)]
    if several:
      everyone = [None] * dist.get_world_size()
      dist.all_gather_object(everyone, (pieces, seen))
    labels: dict = {}
    for _, rank_seen in everyone:
      for d, values in rank_seen.items():
        known = labels.setdefault(d, {})
        for label in values:
          known.setdefault(np.asarray(label).tolist(), label)
    coords = dict(self._coords)
    for d, known in labels.items():
      first = next(iter(known.values()))
      coords[d] = np.array(list(known.values()), dtype=np.asarray(first).dtype)
    out = xl.Dataset(coords=coords)
    names = []
    places: dict = {}
    for rank_pieces, _ in everyone:
      names += [n for n in rank_pieces if n not in names]
    for name in names:
      ref = next(p[name] for p, _ in everyone if name in p)
      dims, key_dims, rest_shape, _, _, dtype = ref
      grid = tuple(len(labels[d]) for d in key_dims)
      rest_dims = [d for d in dims if d not in key_dims]
      order = list(key_dims) + rest_dims
      perm = [order.index(d) for d in dims]
      full = None
      for rank, (rank_pieces, _) in enumerate(everyone):
        if name not in rank_pieces:
          continue
        _, _, _, keys, values, _ = rank_pieces[name]
        if not keys:
          continue
        # where every row goes: one index array per key dim, shared by the
        # variables of a rank that were filed under the same label combinations
        memo = (rank, key_dims, len(keys), keys[0], keys[-1])
        where = places.get(memo)
        if where is None or where[0] != keys:
          pos = [{v: i for i, v in enumerate(labels[d])} for d in key_dims]
          index = tuple(
              np.fromiter((p[key[j]] for key in keys), dtype=np.int64,
                          count=len(keys)) for j, p in enumerate(pos))
          flat = np.ravel_multi_index(index, grid) if grid else None
          whole = flat is not None and flat.size == int(np.prod(grid)) and (
              np.array_equal(flat, np.arange(flat.size)))
          where = places[memo] = (keys, index, whole, {})
        if isinstance(values, torch.Tensor):
          # (one process, device rows) placed, transposed and rounded on the
          # device: one compact copy to the host per variable
          if where[2]:   # every combination, in order: the rows ARE the result
            placed = values.reshape(grid + rest_shape)
          else:
            placed = torch.full(grid + rest_shape, float('nan'),
                                dtype=values.dtype, device=values.device)
            at = where[3].get(values.device)
            if at is None:
              at = where[3][values.device] = tuple(
                  torch.as_tensor(ix, device=values.device) for ix in where[1])
            placed[at] = values
          placed = placed.permute(perm)
          final = final_dtype(name)
          if final is not None and final != placed.dtype:
            placed = placed.to(final)
          full = placed.contiguous().cpu().numpy()
          perm = None
          continue
        if full is None:
          full = np.full(grid + rest_shape, np.nan, dtype=np.result_type(
              np.float32, np.asarray(values).dtype))
        full[where[1]] = values
      if full is None:
        full = np.full(grid + rest_shape, np.nan, dtype=np.float64)
      if perm is not None:
        full = np.transpose(full, perm)
      out.data_vars[name] = xl.DataArray(
          np.ascontiguousarray(full.astype(dtype, copy=False)), dims, coords,
          name)
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


def _prefetched(chunks, lo: int, hi: int, depth, stage=None):
  """Yields chunks[lo], ..., chunks[hi - 1] in order, fetching up to `depth`
  items ahead on ONE background thread (so the fetches themselves stay in
  order -- sequences that read a file sequentially keep doing so).  `depth` is
  an int or a callable read before every fetch (the window size of
  evaluate_chunks is only known after the first chunk).  `stage` (optional) is
  applied to every fetched item ON the fetch thread (`_stage_on_device`: the
  host -> HBM copy of chunk i + 1 runs while the main thread works on chunk
  i); without a fetch thread it is not applied (the metrics upload what they
  read)."""
  depth_now = depth if callable(depth) else (lambda: depth)
  if depth_now() <= 0 or hi - lo <= 1:
    for i in range(lo, hi):
      yield chunks[i]
    return
  fetch = chunks.__getitem__ if stage is None else (
      lambda i: stage(chunks[i]))
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
        while nxt < hi and len(pending) <= depth_now():
          pending.append(pool.submit(fetch, nxt))
          nxt += 1
        yield pending.popleft().result()
    finally:
      for fut in pending:
        fut.cancel()
      if stage is not None:
        # the fetch thread's uploader (pinned ring, copy threads) goes with it
        from weatherbench2_amd import feeder
        pool.submit(feeder.close_thread_uploaders).result()


# host arrays below this size are left to the metrics (coordinates, scalars)
_STAGE_MIN_BYTES = 1 << 20


class _Staged:
  """A chunk pair whose host arrays have been sent to the device by the fetch
  thread, with the event (on the copy stream) that completes the copies."""

  def __init__(self, pair, event, tensors):
    self.pair, self.event, self.tensors = pair, event, tensors


def _stage_on_device(device):
  """stage(pair) for `_prefetched`: every NumPy-backed variable of a chunk
  pair that holds (latitude, longitude) slabs goes to HBM through the calling
  (fetch) thread's uploader (feeder.upload: a pool of copy threads stages
  slices into a pinned ring while the DMA of the previous slice runs;
  evaluation.py:693-705 hands such pageable arrays to every chunk call).
  Device-resident and lazy (gather / concat) variables pass through."""
  import torch
  from weatherbench2_amd import feeder
  dev = torch.device(device)

  def to_device(ds):
    ds = xl.as_dataset(ds)
    moved = []
    hits = [n for n, da in ds.data_vars.items()
            if isinstance(da.data, np.ndarray)
            and da.data.nbytes >= _STAGE_MIN_BYTES
            and da.data.dtype in (np.float32, np.float64)
            and da.data.dtype.isnative]
    if not hits:
      return ds, moved
    # every variable of the chunk in one foreign call (no interpreter lock
    # taken between two variables)
    moved = feeder.upload_many([ds.data_vars[n].data for n in hits], dev,
                               wait=False)
    on_device = dict(zip(hits, moved))
    out = xl.Dataset(coords=ds.coords, attrs=dict(ds.attrs))
    for name, da in ds.data_vars.items():
      out.data_vars[name] = xl.DataArray(on_device.get(name, da.data), da.dims,
                                         ds.coords, name)
    return out, moved

  def stage(pair):
    staged = [to_device(ds) for ds in pair[:2]]
    tensors = [x for _, moved in staged for x in moved]
    if not tensors:
      return pair
    event = torch.cuda.Event()
    event.record(feeder.copy_stream(dev))
    return _Staged((staged[0][0], staged[1][0]) + tuple(pair[2:]), event,
                   tensors)
  return stage


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


def _contiguous(data) -> bool:
  if isinstance(data, np.ndarray):
    return bool(data.flags.c_contiguous)
  return not isinstance(data, (xl.SlabGather, xl.SlabConcat)) and bool(
      data.is_contiguous())


def _same_values(a, b) -> bool:
  if a is b:
    return True
  a = a.values if isinstance(a, xl.DataArray) else a
  b = b.values if isinstance(b, xl.DataArray) else b
  return a is b or np.array_equal(np.asarray(a), np.asarray(b))


_CONCAT_PLANS: dict = {}   # concat_chunks: layout key -> per-variable layout


def concat_chunks(datasets: t.Sequence, time_dim: str,
                  lead_dim: t.Optional[str] = None):
  """The chunks of a (time block x lead block) rectangle as ONE Dataset over
  the concatenated `time_dim` (and `lead_dim`) labels -- without moving data:
  every variable becomes an `xarray_lite.SlabConcat`, an index over the
  chunks' own arrays, which the fused deterministic passes read slab by slab
  through device addresses.  Coordinates that follow the two dims (valid_time,
  the 2-D `time` of a by-init truth chunk) are put together the same way.

  Returns None when the chunks are not equally shaped pieces of such a
  rectangle or hold data that cannot be addressed in place (lazy gathers,
  transposed or strided arrays): the caller then evaluates them one by one."""
  datasets = [xl.as_dataset(d) for d in datasets]
  first = datasets[0]
  if len(datasets) == 1:
    return first
  if time_dim not in first.dims:
    return None
  split = (time_dim,) + ((lead_dim,) if lead_dim and lead_dim in first.dims
                         else ())
  names = list(first.data_vars)
  # position of every chunk in the rectangle, by its labels
  # (whole-list steps here and below: this runs once per window and input on
  # the thread that feeds the device, and a window of the ensemble chunks is
  # 1.2 ms of device time)
  blocks: list = [{} for _ in split]   # label bytes -> (position, labels)
  positions = []
  for j, d in enumerate(split):
    found = [ds.coords.get(d) for ds in datasets]
    if any(c is None or isinstance(c, xl.DataArray) for c in found):
      return None
    found = [np.asarray(c) for c in found]
    block, pos = blocks[j], []
    for key, labels in zip([(c.dtype.str, c.tobytes()) for c in found], found):
      hit = block.get(key)
      if hit is None:
        hit = block[key] = (len(block), labels)
      pos.append(hit[0])
    positions.append(pos)
  where = (list(zip(positions[0], positions[1])) if len(split) == 2
           else [(p, 0) for p in positions[0]])
  n_i = len(blocks[0])
  n_l = len(blocks[1]) if len(split) == 2 else 1
  if len(set(where)) != len(where) or len(where) != n_i * n_l:
    return None
  labels_of = []
  for j in range(len(split)):  # blocks neither share labels nor differ in size
    parts = [lab for _, lab in blocks[j].values()]
    if len({len(p) for p in parts}) != 1:
      return None
    joined = np.concatenate(parts)
    if len(set(joined.tolist())) != len(joined):
      return None
    labels_of.append(joined)
  size = {d: len(labels_of[j]) // (n_i, n_l)[j] for j, d in enumerate(split)}
  count = {time_dim: n_i}
  if len(split) == 2:
    count[split[1]] = n_l

  def assemble(dims, shape, cells, dtype):
    """Block matrix of equally shaped cells {(bi, bl): array} over `dims`."""
    out_shape = tuple(n * count.get(d, 1) for d, n in zip(dims, shape))
    out = np.empty(out_shape, dtype=dtype)
    ax = [dims.index(d) if d in dims else None for d in split] + [None]
    for (bi, bl), cell in cells.items():
      sl = [slice(None)] * len(dims)
      if ax[0] is not None:
        sl[ax[0]] = slice(bi * shape[ax[0]], (bi + 1) * shape[ax[0]])
      if ax[1] is not None:
        sl[ax[1]] = slice(bl * shape[ax[1]], (bl + 1) * shape[ax[1]])
      out[tuple(sl)] = cell
    return out

  def cell_of(at, dims):
    return (at[0] if time_dim in dims else 0,
            at[1] if len(split) == 2 and split[1] in dims else 0)

  # coordinates: common ones must agree, those along the split dims are joined
  coords = {}
  for k, c in first.coords.items():
    cdims = tuple(c.dims) if isinstance(c, xl.DataArray) else (k,)
    if not any(d in split for d in cdims):
      others = [ds.coords.get(k) for ds in datasets[1:]]
      if any(other is not c for other in others):  # (usually shared objects)
        for other in others:
          if other is None or not _same_values(c, other):
            return None
      coords[k] = c
      continue
    if not isinstance(c, xl.DataArray):  # the split dims' own labels
      coords[k] = labels_of[split.index(k)]
      continue
    cells = {}
    ref = np.asarray(c.values)
    for ds, at in zip(datasets, where):
      other = ds.coords.get(k)
      if not isinstance(other, xl.DataArray) or tuple(other.dims) != cdims:
        return None
      value = np.asarray(other.values)
      if value.shape != ref.shape:
        return None
      cells.setdefault(cell_of(at, cdims), value)
    coords[k] = xl.DataArray(assemble(cdims, ref.shape, cells, ref.dtype),
                             cdims)
  out = xl.Dataset(coords=coords, attrs=dict(first.attrs))
  for ds in datasets[1:]:
    if len(ds.data_vars) != len(names):
      return None
  # The layout of the variables is a function of the rectangle and of the
  # first chunk's variables: windows of one evaluation come with the same one
  # again and again.  A remembered layout leaves, per variable, one check of
  # every chunk's array and a list of them.
  plan_key = (time_dim, split, tuple(where), tuple(
      (n, v.dims, tuple(v.shape), v.dtype, type(v.data))
      for n, v in first.data_vars.items()))
  plan = _CONCAT_PLANS.get(plan_key)
  if plan is not None:
    tables = [ds.data_vars for ds in datasets]
    for name, (rdims, rshape, rdtype, kind, providers, template) in zip(
        names, plan):
      if providers is None:   # follows neither split dim: the first chunk's
        ref = first.data_vars[name]
        out.data_vars[name] = xl.DataArray(ref.data, rdims, coords, name)
        continue
      # (whole-list checks: a window of 32 chunks x 11 variables comes here
      # once per window and input -- per-element Python was 0.5 ms of it)
      try:
        das = [tables[p][name] for p in providers]
      except KeyError:
        return None
      bases = [da.data for da in das]
      if ({da.dims for da in das} != {rdims}
          or {type(x) for x in bases} != {kind}
          or {x.dtype for x in bases} != {rdtype}
          or any(x.shape != rshape for x in bases)):
        return None
      if kind is np.ndarray:
        if not all(x.flags.c_contiguous for x in bases):
          return None
      elif not all(map(kind.is_contiguous, bases)):
        return None
      out.data_vars[name] = xl.DataArray(template.with_bases(bases), rdims,
                                         coords, name)
    return out
  new_plan = []
  for name in names:
    ref = first.data_vars[name]
    rdims, rshape, rdtype, kind = ref.dims, ref.shape, ref.dtype, type(ref.data)
    if not any(d in rdims for d in split):
      out.data_vars[name] = xl.DataArray(ref.data, rdims, coords, name)
      new_plan.append((rdims, None, None, None, None, None))
      continue
    if (len(rdims) < 2 or set(rdims[-2:]) != {'latitude', 'longitude'}
        or any(size[d] != rshape[rdims.index(d)] for d in split
               if d in rdims)):
      return None
    outer = rshape[:-2]
    n = int(np.prod(outer, dtype=np.int64))
    template = np.arange(n, dtype=np.int64).reshape(outer)
    bases, cells, providers = [], {}, []
    for p, (ds, at) in enumerate(zip(datasets, where)):
      da = ds.data_vars.get(name)
      if da is None:
        return None
      data = da.data
      if (da.dims != rdims or type(data) is not kind
          or tuple(data.shape) != rshape or data.dtype != rdtype
          or not _contiguous(data)):
        return None
      at = cell_of(at, rdims)
      if at in cells:  # the variable does not follow one of the split dims
        continue
      cells[at] = len(bases) * n + template
      bases.append(data)
      providers.append(p)
    index = assemble(rdims[:-2], outer, cells, np.int64)
    joined = xl.SlabConcat(bases, index, uniform=True)
    out.data_vars[name] = xl.DataArray(joined, rdims, coords, name)
    # (the plain lazy containers only: their arrays are what `kind` says)
    rememberable = kind is np.ndarray or xl._is_torch(bases[0])
    new_plan.append((rdims, tuple(rshape), rdtype, kind, providers,
                     joined.with_bases(())) if rememberable else None)
  if all(e is not None for e in new_plan):
    if len(_CONCAT_PLANS) >= 32:
      _CONCAT_PLANS.clear()
    _CONCAT_PLANS[plan_key] = new_plan
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


class _PieceProgram:
  """What replays the chunks of one structure for every config of a call: one
  chunk program over the configs whose metrics are K1 / K3 passes (they share
  the launches), one map suite per map-metric config, the generic loop for
  whatever could not be recorded -- all inside ONE chunk scope, so a pass one
  part has run answers the others from the cache."""

  def __init__(self, parts):
    self.parts = parts   # [(runnable or None, [config index])]

  def reset(self):
    for runnable, _ in self.parts:
      if runnable is not None:
        runnable.reset()

  def replays(self) -> bool:
    return any(runnable is not None for runnable, _ in self.parts)

  def run(self, forecast, truth_chunk, configs, skipna, sinks):
    from weatherbench2_amd import metrics as gm
    with gm.chunk_scope():
      for runnable, idx in self.parts:
        if runnable is None:
          for i in idx:
            sinks[i].add(_metric_and_region_loop(
                forecast, truth_chunk, configs[i], skipna, compute_chunk=True))
        elif len(idx) == 1 and not hasattr(runnable, 'groups'):
          runnable.run(forecast, truth_chunk, sinks[idx[0]])
        else:
          runnable.run(forecast, truth_chunk, [sinks[i] for i in idx])


def _evaluate_piece(forecast, truth_chunk, configs, skipna, sinks,
                    programs: dict) -> None:
  """One (forecast, truth) piece of a window through the metric x region loop
  of every config and into the configs' sinks: the generic path for the first
  piece of every chunk STRUCTURE (recorded), the recorded program for the rest
  (program.py, map_suite.py -- same launches, same accumulation order, same
  bits)."""
  from weatherbench2_amd import map_suite, metrics as gm, program
  how = program.mode()
  sig = None
  # Derived variables are computed on (and assigned into) every chunk by the
  # loop itself (evaluation.py:402-405): a replay would look for them in a
  # chunk that does not have them yet -- such configs keep the generic path.
  derived = any(c.derived_variables for c in configs)
  if how != '0' and not derived and all(s._on_gpu_or_unset() for s in sinks):
    sig = program.signature(forecast, truth_chunk)
  prog = programs.get(sig) if sig is not None else False
  if prog:
    # A program replays the HOST decisions of the first chunk of its
    # structure: it assumes that chunks with the same structure signature take
    # the same decisions (nothing in the loop branches on the DATA of a chunk)
    # and that coordinate arrays are not edited in place between chunks.  Every
    # VERIFY_EVERY-th chunk of a structure is therefore evaluated both ways
    # from the same accumulator state and must leave the same bits.
    seen = programs[('replays', sig)] = programs.get(('replays', sig), 0) + 1
    if how == 'verify' or (VERIFY_EVERY and seen % VERIFY_EVERY == 0):
      _verify_program(prog, forecast, truth_chunk, configs, skipna, sinks)
    else:
      prog.run(forecast, truth_chunk, configs, skipna, sinks)
    return

  def loop(which=None):
    which = range(len(configs)) if which is None else which
    with gm.chunk_scope():  # one scope: the configs share the passes
      out = []
      for i in which:
        f, t_ = forecast, truth_chunk
        if derived and len(configs) > 1:
          # every pipeline branch of the reference reads the chunk itself
          # (evaluation.py:757-828): one config's derived variables must not
          # show up in the next config's inputs
          f, t_ = xl.as_dataset(f).copy(), xl.as_dataset(t_).copy()
        out.append(_metric_and_region_loop(f, t_, configs[i], skipna,
                                           compute_chunk=True))
      return out
  if prog is False:      # not replayable (or programs are off)
    for sink, result in zip(sinks, loop()):
      sink.add(result)
    return
  with program.Recorder() as rec:
    results = loop()
  for sink, result in zip(sinks, results):
    sink.add(result)
  built = None
  if all(s._on_gpu() for s in sinks):
    maps = [i for i, c in enumerate(configs) if map_suite.applies(c)]
    scalar = [i for i in range(len(configs)) if i not in maps]
    parts = []
    if scalar:
      one = program.build(rec, forecast, truth_chunk,
                          [xl.as_dataset(results[i]) for i in scalar],
                          [sinks[i] for i in scalar], lambda: loop(scalar))
      parts.append((one, scalar))
    for i in maps:
      # map metrics (`deterministic_spatial`): one fused launch per chunk
      parts.append((map_suite.build(configs[i], forecast, truth_chunk,
                                    results[i], sinks[i], skipna), [i]))
    built = _PieceProgram(parts)
    if not built.replays():
      built = None
  programs[sig] = built or False


# every n-th replayed chunk of a structure goes through the generic path as
# well and is compared bit for bit (0: never; WB2HIP_CHUNK_PROGRAM=verify:
# every chunk).  A generic pass costs ~2.5 ms of host time against ~0.3 ms per
# replayed official chunk: 256 keeps the check below 4 % of a long run.
VERIFY_EVERY = int(os.environ.get('WB2HIP_CHUNK_PROGRAM_VERIFY_EVERY', '256'))


def _evaluate_map_window(window, configs, skipna, sinks, programs,
                         lead_dim) -> None:
  """A window of chunks through map-metric configs (`deterministic_spatial`,
  scripts/evaluate.py:431-435, 471-478).  `xbeam.Mean` combines any number of
  chunks per key before it touches the output (evaluation.py:735-744): the
  chunks of the window whose map suite exists and that carry the same lead
  labels are accumulated by ONE launch per suite, in chunk order (the running
  sums cross HBM once per group instead of once per chunk -- same additions,
  same order, same bits as chunk by chunk).  A chunk that needs the generic
  path (the first of its structure) goes alone, after everything queued
  before it."""
  from weatherbench2_amd import map_suite, metrics as gm, program
  pending: dict = {}   # (structure, lead labels) -> [prog, [(forecast, truth)]]

  def drain():
    for prog, pairs in pending.values():
      with gm.chunk_scope():
        for suite, idx in prog.parts:
          suite.run_many(pairs, sinks[idx[0]])
    pending.clear()
  replayable = program.mode() != '0' and all(
      s._on_gpu_or_unset() for s in sinks)
  for forecast, truth_chunk in window:
    sig = program.signature(forecast, truth_chunk) if replayable else None
    prog = programs.get(sig) if sig is not None else None
    suites = prog and all(isinstance(r, map_suite.MapSuite) and len(idx) == 1
                          for r, idx in prog.parts)
    if not suites or program.mode() == 'verify':
      drain()
      _evaluate_piece(forecast, truth_chunk, configs, skipna, sinks, programs)
      continue
    labels = forecast.coords.get(lead_dim) if lead_dim else None
    if isinstance(labels, xl.DataArray):
      labels = labels.values
    key = (sig, None if labels is None else
           (np.asarray(labels).dtype.str, np.asarray(labels).tobytes()))
    pending.setdefault(key, [prog, []])[1].append((forecast, truth_chunk))
  drain()


def _verify_program(prog, forecast, truth_chunk, configs, skipna, sinks):
  """WB2HIP_CHUNK_PROGRAM=verify: the chunk through the program AND through the
  generic path, each from the same accumulator state -- they must leave the
  same bits behind (NaN == NaN)."""
  import torch

  def snapshot(mean):
    if getattr(mean, 'keeps_time', False):
      return mean.snapshot()
    for acc in mean._acc.values():
      acc.settle()
    return {n: (acc.total.clone(), acc.count.clone(), list(acc.labels),
                dict(acc.row_of)) for n, acc in mean._acc.items()}

  def restore(mean, state):
    if getattr(mean, 'keeps_time', False):
      return mean.restore(state)
    for n, (total, count, labels, row_of) in state.items():
      acc = mean._acc[n]
      acc.total, acc.count = total.clone(), count.clone()
      acc.labels, acc.row_of, acc.dst = list(labels), dict(row_of), {}
  before = [snapshot(m) for m in sinks]
  prog.run(forecast, truth_chunk, configs, skipna, sinks)
  replayed = [snapshot(m) for m in sinks]
  for m, state in zip(sinks, before):
    restore(m, state)
  from weatherbench2_amd import metrics as gm
  with gm.chunk_scope():
    for m, c in zip(sinks, configs):
      m.add(_metric_and_region_loop(forecast, truth_chunk, c, skipna,
                                    compute_chunk=True))
  for m, state in zip(sinks, replayed):
    if getattr(m, 'keeps_time', False):
      state = state[0]
    for n, acc in m._acc.items():
      if hasattr(acc, 'settle'):
        acc.settle()
      for a, b in zip(state[n][:2], (acc.total, acc.count)):
        same = a.shape == b.shape and bool(
            ((a == b) | (torch.isnan(a) & torch.isnan(b))).all().item())
        if not same:
          raise AssertionError(f'chunk program and generic path differ on {n}')
  prog.reset()   # (the accumulators were replaced: new addresses)


# K1 chunking of evaluate_chunks (pinned: the result must not depend on how
# many chunks share a launch).  32 rows is the measured optimum of launches of
# ~200 slabs (profiles/r01_rows_per_chunk.md); the windows of the production
# path launch 1 000+ slabs, where the per-workgroup prologue and fold weigh
# more than the tail: 24 / 32 / 40 / 48 / 64 rows give 412 / 425 / 423 / 434 /
# 434 G evals/s in the default window and 291 / 296 / 282 / 296 / 275 G chunk
# by chunk (official chunks, one box, profiles/r06_round_log.md)
EVALUATE_ROWS_PER_CHUNK = int(os.environ.get('WB2HIP_EVALUATE_ROWS', '48'))


# evaluate_chunks(batch_chunks=None): chunks per window = what holds this many
# bytes of (forecast + truth) input, at most AUTO_BATCH_MAX
AUTO_BATCH_BYTES = 16 << 30
AUTO_BATCH_MAX = 32


def _input_bytes(ds: xl.Dataset) -> int:
  total = 0
  for da in ds.data_vars.values():
    data = da.data
    itemsize = getattr(data, 'itemsize', None)
    if itemsize is None and hasattr(data, 'element_size'):
      itemsize = data.element_size()
    total += int(np.prod(da.shape, dtype=np.int64)) * int(itemsize or 4)
  return total


def evaluate_chunks(
    chunks: t.Sequence[tuple],
    eval_config: t.Union[config.Eval, t.Dict[str, config.Eval]],
    skipna: bool = False,
    device=None,
    prefetch: int = 2,
    *,
    truth=None,
    climatology=None,
    by_init: bool = True,
    batch_chunks: t.Optional[int] = None,
) -> t.Union[xl.Dataset, t.Dict[str, xl.Dataset]]:
  """Evaluates (forecast, truth) chunks and returns the temporal mean.

  `eval_config` may be a dict {name: Eval} like `evaluate_with_beam`'s
  `eval_configs` (evaluation.py:757-828; the documented 0.25-degree command
  line runs `--eval_configs=deterministic,deterministic_temporal`): the
  reference builds one pipeline branch per config, each reading the chunks
  again; here every chunk is read ONCE -- the configs' loops run inside one
  chunk scope (a fused pass one config has run answers the next from the
  cache), their recorded programs share the launches -- and the result is
  {name: Dataset}.

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

  `batch_chunks` = k (deterministic suites: MSE / RMSE / MAE / Bias / ACC /
  wind vectors / SEEPS, the scalar ensemble metrics -- CRPS, spread, skill,
  ensemble-mean MSE, variance: K3 reads the chunks of a window by address --
  and the map suite; ignored otherwise; None = as many chunks as hold
  AUTO_BATCH_BYTES of input, at most AUTO_BATCH_MAX -- 24 of the official
  0.25-degree chunks) evaluates k consecutive chunks in ONE pass of
  the metric x region loop: they are concatenated without copying (`concat_chunks`: a
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
  several = isinstance(eval_config, dict)
  names = list(eval_config) if several else [None]
  configs = [eval_config[k] for k in names] if several else [eval_config]
  if not configs:
    raise ValueError('no eval config given')
  switches = {tuple(bool(getattr(c, k, False)) for k in (
      'evaluate_climatology', 'evaluate_probabilistic_climatology',
      'evaluate_persistence')) for c in configs}
  if len(switches) > 1:
    # (the switches replace the FORECAST of a chunk: configs that disagree on
    # them do not evaluate the same data)
    raise ValueError('eval configs with different baseline switches '
                     '(evaluate_climatology / _probabilistic_climatology / '
                     '_persistence) cannot share one pass over the chunks')
  substitute = _chunk_substitution(configs[0], truth, climatology, by_init)
  auto_batch = batch_chunks is None
  batch_chunks = AUTO_BATCH_MAX if auto_batch else max(1, int(batch_chunks))
  from weatherbench2_amd import map_suite
  # map-metric configs (`deterministic_spatial`): windows too -- the chunks of
  # a window that carry the same lead labels go through ONE accumulate launch
  # (map_suite.MapSuite.run_many)
  maps_only = all(map_suite.applies(c) for c in configs)
  if not maps_only and any(c.derived_variables or not all(
      getattr(m, '_reads_slabs_in_place', False)
      for m in c.metrics.values()) for c in configs):
    # Chunks are batched only for metrics that read a concatenation in place
    # (the deterministic suite and the scalar ensemble metrics: address
    # tables); every other metric would
    # materialise the window first -- a copy of the data, slower than going
    # chunk by chunk.  Derived variables are computed on (and assigned into)
    # each chunk as the caller handed it in (evaluation.py:402-405).
    batch_chunks, auto_batch = 1, False
  sinks: list = []  # per config: RunningMean, or RunningConcat for
  window: list = []  # temporal_mean=False
  # chunk structures seen so far -> their replayable program (program.py), or
  # False where the generic path has to stay
  programs: dict = {}

  def flush():
    if not window:
      return
    first = window[0][0]
    time_dim = 'time' if first.has_dim('time') else 'init_time'
    lead_dim = _lead_dim(first)
    lead_dim = lead_dim if first.has_dim(lead_dim) else None
    if not sinks:
      for c in configs:
        if getattr(c, 'temporal_mean', True) is False:
          sinks.append(RunningConcat(time_dim, device, split_dim=lead_dim))
        else:
          sinks.append(RunningMean(time_dim, skipna, device,
                                   split_dim=lead_dim,
                                   split_order='first_seen'))
    if maps_only:
      _evaluate_map_window(window, configs, skipna, sinks, programs, lead_dim)
    else:
      for forecast, truth_chunk in _batches(window, time_dim, lead_dim):
        _evaluate_piece(forecast, truth_chunk, configs, skipna, sinks,
                        programs)
    window.clear()

  with metrics_lib.pinned_rows_per_chunk(EVALUATE_ROWS_PER_CHUNK):
    # while the window size is undecided (auto) only `prefetch` items are
    # fetched ahead; afterwards enough to fill a window
    depth = lambda: max(prefetch, (batch_chunks - 1)
                        if prefetch and not auto_batch else 0)
    stage = None
    if prefetch and os.environ.get('WB2HIP_STAGE_ON_FETCH', '1') != '0':
      import torch
      if torch.cuda.is_available() and (
          device is None or torch.device(device).type == 'cuda'):
        from weatherbench2_amd import engine
        stage = _stage_on_device(device if device is not None
                                 else engine.require_gpu())
    for item in _prefetched(chunks, lo, hi, depth, stage):
      if isinstance(item, _Staged):
        # the pass over this chunk is ordered after its copies (no host wait)
        import torch
        cur = torch.cuda.current_stream()
        cur.wait_event(item.event)
        for x in item.tensors:  # allocated on the copy stream, read on this one
          x.record_stream(cur)
        item = item.pair
      forecast, truth_chunk = item[0], item[1]
      forecast = xl.as_dataset(forecast)
      if substitute is not None:
        forecast = xl.as_dataset(substitute(forecast, truth_chunk))
      window.append((forecast, xl.as_dataset(truth_chunk)))
      if auto_batch:  # sized by the first chunk: its inputs' bytes
        nbytes = max(1, sum(_input_bytes(ds) for ds in window[0]))
        budget = AUTO_BATCH_BYTES
        try:  # a window in flight + the next one being staged must fit: at
          import torch  # most a quarter of the free device memory per window
          if torch.cuda.is_available():
            budget = min(budget, torch.cuda.mem_get_info()[0] // 4)
        except Exception:
          pass
        batch_chunks = int(min(AUTO_BATCH_MAX, max(1, budget // nbytes)))
        auto_batch = False
      if len(window) >= batch_chunks:
        flush()
    flush()
  assert sinks
  # (the sinks meet the other ranks one after the other, in config order)
  results = [sink.result() for sink in sinks]
  return dict(zip(names, results)) if several else results[0]


# ---------------------------------------------------------------------------
# Baseline substitutions (evaluation.py:165-193, 452-472, 618-675;
# utils.py:47-70): label work on the host, the data stays where it is
# ---------------------------------------------------------------------------
def _index_values(ds: xl.Dataset, name: str) -> np.ndarray:
  c = ds.coords[name]
  return np.asarray(c.values if isinstance(c, xl.DataArray) else c)


def _positions(have: np.ndarray, want: np.ndarray, what: str) -> np.ndarray:
  """Positions of the labels `want` in the index `have` (KeyError like .sel)."""
  pos = {v: i for i, v in enumerate(xl.label_list(have))}
  try:
    flat = [pos[v] for v in xl.label_list(want)]
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
  return 'lead_time' if 'lead_time' in forecast.coords or (
      forecast.has_dim('lead_time')) else 'prediction_timedelta'


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
  # `truth.sel(time=init_time)` keeps EVERY variable of the truth dataset
  # (`variables` only sizes the reference's thread pool, :661-667); variables
  # without a time dim pass through and get the lead dim like the others
  del variables
  names = [k for k in truth.keys() if 'time' in truth[k].dims]
  # expand_dims puts the new dim first: (lead_time, init_time, ...)
  out = _gather_dataset(truth, names, {'time': where}, (lead_dim, 'init_time'),
                        where.shape, coords)
  for k in truth.keys():
    if k not in names:
      wide = truth[k].expand_dims({lead_dim: lead})
      out.data_vars[k] = xl.DataArray(wide.data, wide.dims, out.coords, k)
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
