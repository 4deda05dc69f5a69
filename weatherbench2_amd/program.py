"""Chunk programs: the metric x region loop of ONE chunk structure, recorded
once and replayed with new addresses.

The reference's pipeline calls `_evaluate_chunk` once per chunk -- 2 920 init
times x 40 leads of `init_time=1,lead_time=1` chunks in the official 0.25-degree
run (/root/reference/weatherbench2/evaluation.py:583-599, 693-705, 735-744;
docs/source/official-evaluation.md:537-556) -- and every call walks the same
Python: 13 variables x 5 metrics x 16 regions of dims, labels, dtypes and
result assembly, ~14 000 function calls for 0.26 ms of GPU work.  Chunks of one
evaluation all have the SAME structure: only the addresses of their arrays and
the valid times (which climatology slabs to subtract, which lead rows to add
to) change.  So `evaluation.evaluate_chunks` lets the generic path run the
first chunk of a structure while a recorder watches, and replays the rest:

  record   every fused launch of the loop (metrics._run_group): plan, mode and,
           per input slab, WHERE its address comes from -- a variable of the
           forecast / truth chunk (new pointer, same offsets), or a resident
           array gathered by valid time (same pointer, table recomputed by the
           function the generic path used: metrics._climatology_slabs);
  probe    the same loop once more on FAKE launch results u_k = k + 1/3 (k =
           the element's position in the launches' outputs): whatever slicing,
           stacking, transposing, NaN-filling and dtype conversion the metric
           classes, `_assemble` and `merge_metrics` do to build the result
           Dataset, element e of it then says which output element it shows
           (floor) and whether it went through float32 (k + 1/3 is not a
           float32 number);
  verify   the mapping reproduces the first chunk's real result bit for bit
           (NaN == NaN), else the structure keeps the generic path;
  replay   per chunk ONE wb2_program_replay call (`_Native`, csrc/program.cpp:
           the chunk's base pointers and valid times in; address table, its
           upload, every recorded launch -- wb2_det_suite_step /
           wb2_det_wind_suite_step, or wb2_ens_partials_addr + wb2_ens_combine
           for an ensemble pass -- and ONE wb2_gather_accumulate per config
           that feeds every variable's (sum, count) accumulators on the C
           side); WB2HIP_NATIVE_REPLAY=0: the same steps from Python.

The replayed launches are the recorded ones (same plan, same chunking, same
kernels) and the accumulation adds the same values in the same order, so a run
with programs equals a run without them bit for bit
(tests/test_chunk_program_gpu.py; WB2HIP_CHUNK_PROGRAM=0 disables, =verify
runs both paths on every chunk and compares).
"""
from __future__ import annotations

import os
import threading
import typing as t
import weakref

import numpy as np
import torch

from weatherbench2_amd import _lib, engine
from weatherbench2_amd import xarray_lite as xl

_THIRD = 1.0 / 3.0
_MAX_ELEMENTS = 1 << 21   # k + 1/3 must survive float32 with floor(.) == k


class _Active(threading.local):

  def __init__(self):
    self.recorder = None


_ACTIVE = _Active()


def recorder():
  """The recorder watching the calling thread's launches, or None."""
  return _ACTIVE.recorder


def mode() -> str:
  return os.environ.get('WB2HIP_CHUNK_PROGRAM', '1')


class Recorder:
  """Collects the fused launches of one pass of the loop (`probe` = False) or
  stands in for them with index-valued results (`probe` = True)."""

  def __init__(self, probe: bool = False):
    self.probe = probe
    self.launches: list = []      # dicts, in launch order
    self.tables: dict = {}        # id(table) -> (table, recompute(forecast))
    self.offset = 0               # probe: elements handed out so far

  def __enter__(self):
    self._old = _ACTIVE.recorder
    _ACTIVE.recorder = self
    return self

  def __exit__(self, *exc):
    _ACTIVE.recorder = self._old

  def note_table(self, table, recompute, gather=None) -> None:
    """`table` (a slab table the generic path derived from the chunk's LABELS)
    is what `recompute(forecast)` returns for another chunk.  `gather`
    (metrics._climatology_gather) splits it into a structural part and the
    few per-chunk values a native replay needs."""
    if table is not None:
      self.tables[id(table)] = (table, recompute, gather)

  def fake_metrics(self, n_metric, n_region, n_outer, device,
                   extend: bool = False):
    """`extend`: a second output block of the launch noted last (the wind
    block of a launch with wind-vector pairs)."""
    n = n_metric * n_region * n_outer
    out = (torch.arange(self.offset, self.offset + n, dtype=torch.float64,
                        device=device) + _THIRD).reshape(n_metric, n_region,
                                                         n_outer)
    self.offset += n
    if extend:
      self.launches[-1]['n'] += n
    else:
      self.launches.append({'n': n})
    return out

  def record(self, **launch) -> None:
    self.launches.append(launch)


# ---------------------------------------------------------------------------
# structure signature
# ---------------------------------------------------------------------------
_VARYING = ('time', 'init_time', 'valid_time', 'lead_time',
            'prediction_timedelta')


_DIGESTS: dict = {}   # id(array) -> (array, digest): coordinates recur


def _digest(v: np.ndarray):
  """Content digest of a coordinate array, remembered per array OBJECT (kept
  alive here, so the id cannot be reused): chunks of one dataset share their
  latitude / longitude / level arrays, and nobody writes into coordinates."""
  hit = _DIGESTS.get(id(v))
  if hit is not None and hit[0] is v:
    return hit[1]
  d = engine.digest(v)
  if len(_DIGESTS) > 256:
    _DIGESTS.clear()
  _DIGESTS[id(v)] = (v, d)
  return d


def _data_sig(data):
  if isinstance(data, torch.Tensor):
    # (not the alignment of the address: the replay picks the kernel for the
    # addresses it is given, chunk by chunk)
    return ('t', data.shape, data.stride(), data.dtype, data.device)
  if isinstance(data, xl.SlabConcat) and data.on_device:
    b = data.bases[0]
    return ('c', tuple(data.shape), len(data.bases), tuple(b.shape),
            str(b.dtype), engine.digest(data.index))
  return None


def signature(forecast: xl.Dataset, truth: xl.Dataset):
  """Everything about a (forecast, truth) pair that shapes the loop's host
  work EXCEPT the addresses of its arrays and the values of its time
  coordinates; None if the pair holds anything a program cannot address
  (host arrays, lazy gathers)."""
  parts = []
  for ds in (forecast, truth):
    for name, da in ds.data_vars.items():
      sig = _data_sig(da.data)
      if sig is None:
        return None
      parts.append((name, tuple(da.dims), sig))
    for k, c in ds.coords.items():
      if isinstance(c, xl.DataArray):
        v, dims = c.values, tuple(c.dims)
      else:
        v, dims = c, None
      if not isinstance(v, np.ndarray):
        v = np.asarray(v)
      if k in _VARYING or (dims and any(d in _VARYING for d in dims)):
        parts.append((k, dims, v.shape, v.dtype.str))
      elif v.dtype == object:
        parts.append((k, dims, tuple(v.ravel().tolist())))
      else:
        parts.append((k, dims, v.shape, v.dtype.str, _digest(v)))
    parts.append(None)
  return tuple(parts)


# ---------------------------------------------------------------------------
# building
# ---------------------------------------------------------------------------
_NO_TABLE = np.zeros(0, dtype=np.int64)


def _run_launches(launches, outs, forecast, truth, device, stream) -> None:
  """One chunk through recorded launches: their address tables in ONE upload
  (a few hundred integers), then the launches, each on its part."""
  parts = [la.tables(forecast, truth) for la in launches]
  sizes = [p.size for p in parts]
  base = 0
  if sum(sizes):
    table = engine.upload_table(np.concatenate(parts), device, cache=False)
    base = table.data_ptr()
  off = 0
  for la, n, out in zip(launches, sizes, outs):
    la.launch(base + 8 * off, out, stream)
    off += n


class _Launch:
  """One recorded launch, ready to be re-addressed."""

  def __init__(self, rec: dict, fmap: dict, tmap: dict, device):
    pl, n_total = rec['plan'], rec['n_total']
    aux = rec.get('aux')
    if aux is not None and not isinstance(aux, torch.Tensor):
      raise _NotReplayable('a launch with a host auxiliary field')
    members = rec['members']   # [(geo, arrays, tables, tensors)]
    n_in = len(members[0][1])
    self.plan, self.mode, self.skipna = pl, rec['mode'], rec['skipna']
    self.dtype, self.n_total, self.n_in = rec['dtype'], n_total, n_in
    self.slot = np.zeros((n_in, n_total), dtype=np.int64)
    self.rel = np.zeros((n_in, n_total), dtype=np.int64)
    self.sources: list = []    # how to get pointer `slot` of a new chunk
    self.source_keys: list = []  # what pointer `slot` is (shared by launches)
    self.dynamic: list = []    # (input, offset, count, resident array,
    self.keep: list = []       #  recompute, gather)
    source_of: dict = {}

    def slot_for(key, getter):
      if key not in source_of:
        source_of[key] = len(self.sources)
        self.sources.append(getter)
        self.source_keys.append(key)
      return source_of[key]
    off = 0
    for geo, arrays, tables, tensors in members:
      n = geo.n_outer
      for j, (raw, tb, x) in enumerate(zip(arrays, tables, tensors)):
        role = fmap.get(id(raw)) or tmap.get(id(raw))
        if role is not None:
          which, name = role
          if x is not raw:
            # the pass read a CONVERTED copy (a cast, an upload): the chunk's
            # own array is not what the kernel saw
            raise _NotReplayable(f'{name}: the pass read a copy of the chunk')
          if isinstance(x, xl.SlabConcat):
            step = (x.slab_shape[0] * x.slab_shape[1] *
                    x.bases[0].element_size())
            index = x.index.ravel() if tb is None else x.index.ravel()[tb]
            base = np.searchsorted(x.offsets, index, side='right') - 1
            self.rel[j, off:off + n] = (index - x.offsets[base]) * step
            for k in np.unique(base):
              s = slot_for((which, name, int(k)),
                           lambda f, t_, w=which, nm=name, k=int(k): (
                               f if w == 'f' else t_)[nm].data.bases[k])
              self.slot[j, off:off + n][base == k] = s
          else:
            from weatherbench2_amd import metrics as gm
            addr, held = gm._slab_addresses(x, tb, pl.n_row, pl.n_col, n)
            _same_storage(held, x, name)
            self.rel[j, off:off + n] = addr - x.data_ptr()
            self.slot[j, off:off + n] = slot_for(
                (which, name), lambda f, t_, w=which, nm=name: (
                    f if w == 'f' else t_)[nm].data)
          continue
        # Not a variable of the chunk: it must be a RESIDENT array whose slab
        # table the generic path derived from the chunk's labels and told the
        # recorder how to recompute (the climatology of ACC / SEEPS).  Anything
        # else -- a temporary the pass made for this chunk -- has an address
        # that means nothing for the next chunk.
        hit = rec['tables'].get(id(tb)) if tb is not None else None
        if hit is None or not isinstance(x, (torch.Tensor, xl.SlabGather)):
          raise _NotReplayable(f'input {j} of a launch is neither a variable '
                               'of the chunk nor an explained resident gather')
        base_tensor = x.base if isinstance(x, xl.SlabGather) else x
        self.keep.append(base_tensor)
        self.slot[j, off:off + n] = slot_for(
            ('r', id(base_tensor)), lambda f, t_, b=base_tensor: b)
        self.dynamic.append((j, off, n, x, hit[1], hit[2]))
      off += n
    self.n_metric = _lib.GENERIC_KQ.get(self.mode, _lib.NMETRIC)
    self.n_values = self.n_metric * pl.n_region * n_total
    self.n_pair = int(rec.get('n_pair') or 0)
    if self.n_pair:
      # wind-vector pairs answered by the launch itself (metrics._run_group):
      # its output is the per-variable block followed by the wind block
      self.step = engine.PairSuiteStep(pl, self.mode, self.dtype, self.skipna,
                                       n_total, self.n_pair)
      self.n_values = self.step.n_values
      return
    # (an auxiliary field -- SEEPS' masked dry fraction -- is a property of
    # the metric's climatology: resident, the same for every chunk)
    self.step = engine.SuiteStep(pl, self.mode, self.dtype, self.skipna,
                                 n_total, by_address=True, aux=aux,
                                 scalar=float(rec.get('scalar') or 0.0))

  def addresses(self, forecast, truth) -> np.ndarray:
    ptrs = np.fromiter((g(forecast, truth).data_ptr() for g in self.sources),
                       dtype=np.int64, count=len(self.sources))
    addr = ptrs[self.slot] + self.rel
    if self.dynamic:
      from weatherbench2_amd import metrics as gm
      memo: dict = {}
      for j, off, n, x, recompute, _ in self.dynamic:
        table = recompute(forecast, memo)
        a, _ = gm._slab_addresses(x, table, self.plan.n_row, self.plan.n_col, n)
        addr[j, off:off + n] = a
    return addr


  # a chunk's replay: every launch says which addresses it needs (tables), the
  # program uploads them in ONE table, every launch runs on its part of it
  def tables(self, forecast, truth) -> np.ndarray:
    addr = self.addresses(forecast, truth)
    self.step.aligned = not (addr & 15).any()
    return addr.ravel()

  def launch(self, base: int, out, stream) -> None:
    step = 8 * self.n_total
    tables = [base + j * step for j in range(self.n_in)]
    if self.n_pair:
      self.step.run(None, tables, out=out, stream_ptr=stream)
    else:
      self.step.run(None, tables, metrics=out, stream_ptr=stream)


def _same_storage(held, raw, what: str) -> None:
  """`held` = what metrics._slab_addresses addressed for the chunk's array
  `raw`: the array itself or a view of it -- a compact copy (slabs that are not
  intact in a strided view) has addresses that mean nothing for the next
  chunk."""
  if isinstance(held, torch.Tensor) and isinstance(raw, torch.Tensor) and (
      held.untyped_storage().data_ptr() != raw.untyped_storage().data_ptr()):
    raise _NotReplayable(f'{what}: the pass read a copy of the chunk')


def _view_offset(view: torch.Tensor, raw, what: str) -> int:
  """Byte offset of `view` (what the pass read) inside the chunk's own array
  `raw`: the same for every chunk of the structure."""
  if not isinstance(raw, torch.Tensor) or view.dtype != raw.dtype or (
      view.device != raw.device):
    raise _NotReplayable(f'{what}: the pass read a copy of the chunk')
  store = raw.untyped_storage()
  lo = view.data_ptr()
  hi = lo + max(view.numel(), 1) * view.element_size()
  if view.untyped_storage().data_ptr() != store.data_ptr() or not (
      store.data_ptr() <= lo and hi <= store.data_ptr() + store.nbytes()):
    raise _NotReplayable(f'{what}: the pass read a copy of the chunk')
  return lo - raw.data_ptr()


class _EnsPass:
  """An ensemble pass (K3 + its region fold) over slabs given by ADDRESS
  (wb2_ens_partials_addr): member 0's slab and the truth slab of every outer
  index are `slot` / `rel` [2][n_total] -- pointer source and byte offset, as
  for `_Launch`.  Subclasses fill them."""

  def _setup(self, pl, skipna, dtype, n_member, member_stride, n_total,
             device):
    self.plan, self.skipna, self.dtype = pl, bool(skipna), dtype
    self.n_member, self.member_stride = int(n_member), int(member_stride)
    self.n_total = int(n_total)
    self.n_metric = _lib.NMETRIC_ENS
    self.n_values = self.n_metric * pl.n_region * self.n_total
    lib = self._lib = _lib.load()
    k = lib.wb2_ens_num_slots(int(self.skipna))
    tile = lib.wb2_ens_tile_cols(pl.n_col)
    n_ctile = -(-pl.n_col // tile)
    seg_eoff, n_ts = pl.seg_entries(tile)
    self.partials = torch.empty((self.n_total, pl.n_chunk, pl.nwf, n_ts, k),
                                dtype=torch.float64, device=device)
    self.keep = (seg_eoff, pl)
    ptr = _lib.ptr
    # (dtype, skipna, ens addresses, truth addresses, ...)
    self.partials_args = [
        engine._DTYPES[dtype], int(self.skipna), None, None, self.n_member,
        self.member_stride, self.n_total, pl.n_row, pl.n_col, ptr(pl.w_row),
        ptr(pl.w_col), ptr(pl.wfield), ptr(pl.chunk_row0), ptr(pl.chunk_nrow),
        pl.n_chunk, n_ctile, ptr(pl.seg_col0), ptr(seg_eoff), pl.n_seg, n_ts,
        ptr(self.partials)]
    self.combine_args = [
        int(self.skipna), ptr(self.partials), self.n_total, pl.n_chunk, pl.nwf,
        pl.n_seg, ptr(seg_eoff), n_ts, ptr(pl.band_chunk0), pl.n_band,
        ptr(pl.coef_band), ptr(pl.coef_seg), ptr(pl.region_wf),
        ptr(pl.region_wsum), pl.n_region, None]

  def fuse_key(self):
    return (id(self.plan), self.skipna, self.n_member, self.member_stride,
            self.dtype)

  def tables(self, forecast, truth) -> np.ndarray:
    ptrs = np.fromiter((g(forecast, truth).data_ptr() for g in self.sources),
                       dtype=np.int64, count=len(self.sources))
    return (ptrs[self.slot] + self.rel).ravel()

  def launch(self, base: int, out, stream) -> None:
    args = self.partials_args
    args[2] = base
    args[3] = base + 8 * self.n_total
    lib = self._lib
    hook = engine._LAUNCH_HOOK
    if hook is not None:
      hook('begin', 'ens_partials')
    status = lib.wb2_ens_partials_addr(*args, stream)
    if status != 0:
      _lib.check(status, 'wb2_ens_partials_addr')
    if hook is not None:
      hook('end', 'ens_partials')
    status = lib.wb2_ens_combine(*self.combine_args, out.data_ptr(), stream)
    if status != 0:
      _lib.check(status, 'wb2_ens_combine')


class _EnsLaunch(_EnsPass):
  """One recorded ensemble pass (metrics._ens_pass) of a variable of the chunk
  -- or of a window of chunks read in place (metrics._ens_pass_concat: the
  variable is an xarray_lite.SlabConcat, one pointer source per chunk) --,
  replayed with new base pointers: the offsets are functions of the shapes
  alone."""

  def __init__(self, rec: dict, fmap: dict, tmap: dict, device):
    pl = rec['plan']
    role_e, role_t = fmap.get(id(rec['ens_raw'])), tmap.get(id(rec['truth_raw']))
    if role_e is None or role_t is None:
      raise _NotReplayable('an ensemble pass over arrays that are not '
                           'variables of the chunk')
    n = int(rec['n_outer'])
    self.sources, self.source_keys = [], []
    self.slot = np.zeros((2, n), dtype=np.int64)
    self.rel = np.zeros((2, n), dtype=np.int64)
    source_of: dict = {}

    def slot_for(key, getter):
      if key not in source_of:
        source_of[key] = len(self.sources)
        self.sources.append(getter)
        self.source_keys.append(key)
      return source_of[key]

    def whole(which, name):
      return slot_for((which, name), lambda f, t_, w=which, nm=name: (
          f if w == 'f' else t_)[nm].data)

    def per_chunk(j, which, name, x, index, step):
      home = np.searchsorted(x.offsets, index, side='right') - 1
      self.rel[j] = (index - x.offsets[home]) * step
      for k in np.unique(home):
        self.slot[j][home == k] = slot_for(
            (which, name, int(k)), lambda f, t_, w=which, nm=name, k=int(k): (
                f if w == 'f' else t_)[nm].data.bases[k])
    lay = rec.get('concat')
    if lay is None:
      dtype = rec['ens'].dtype
      step = pl.n_row * pl.n_col * rec['ens'].element_size()
      host = lambda tb: (np.arange(n, dtype=np.int64) if tb is None
                         else tb.cpu().numpy().astype(np.int64))
      self.rel[0] = (_view_offset(rec['ens'], rec['ens_raw'], role_e[1]) +
                     host(rec['ens_table']) * step)
      self.rel[1] = (_view_offset(rec['truth'], rec['truth_raw'], role_t[1]) +
                     host(rec['truth_table']) * step)
      self.slot[0] = whole(*role_e)
      self.slot[1] = whole(*role_t)
    else:
      dtype = lay['dtype']
      x, tdata, tb = rec['ens_raw'], lay['truth_data'], lay['truth_table']
      step = pl.n_row * pl.n_col * x.bases[0].element_size()
      per_chunk(0, role_e[0], role_e[1], x, lay['ens_first'], step)
      if tdata is not rec['truth_raw']:
        raise _NotReplayable(f'{role_t[1]}: the pass read a copy of the chunk')
      if isinstance(tdata, xl.SlabConcat):
        index = tdata.index.ravel()
        per_chunk(1, role_t[0], role_t[1], tdata,
                  index if tb is None else index[tb], step)
      else:
        from weatherbench2_amd import metrics as gm
        addr, held = gm._slab_addresses(tdata, tb, pl.n_row, pl.n_col, n)
        _same_storage(held, tdata, role_t[1])
        self.rel[1] = addr - tdata.data_ptr()
        self.slot[1] = whole(*role_t)
    self._setup(pl, rec['skipna'], dtype, rec['n_member'],
                rec['member_stride'], n, device)


class _EnsFused(_EnsPass):
  """Several recorded ensemble passes of one chunk (its variables: same plan,
  member count and stride, dtype, NaN rule) as ONE launch over the slabs of all
  of them, one fold.  The kernels are those of the separate passes and the
  fold is per slab: the same bits.  The output block is
  [metric][region][all slabs] -- `perm` maps the separate blocks' elements
  into it."""

  def __init__(self, singles, device):
    first = singles[0]
    self.counts = [la.n_total for la in singles]
    self.sources, self.source_keys = [], []
    source_of: dict = {}
    slots = []
    for la in singles:
      remap = np.empty(len(la.sources), dtype=np.int64)
      for i, (key, getter) in enumerate(zip(la.source_keys, la.sources)):
        if key not in source_of:
          source_of[key] = len(self.sources)
          self.sources.append(getter)
          self.source_keys.append(key)
        remap[i] = source_of[key]
      slots.append(remap[la.slot])
    self.slot = np.concatenate(slots, axis=1)
    self.rel = np.concatenate([la.rel for la in singles], axis=1)
    self._setup(first.plan, first.skipna, first.dtype, first.n_member,
                first.member_stride, sum(self.counts), device)

  def perm(self) -> np.ndarray:
    """Element of the separate passes' output blocks (one after the other) ->
    element of the fused block."""
    n_mr = self.n_metric * self.plan.n_region
    out, off = [], 0
    for n in self.counts:
      mr = np.arange(n_mr, dtype=np.int64)[:, None]
      out.append((mr * self.n_total + off + np.arange(n, dtype=np.int64)
                  ).ravel())
      off += n
    return np.concatenate(out)


def _fuse_ensemble_launches(launches, device):
  """(launches with the compatible ensemble passes fused -- the fused launch
  stands where the first of its passes stood --, old -> new arena index or
  None)."""
  if os.environ.get('WB2HIP_FUSE_ENSEMBLE', '1') == '0':
    return launches, None
  key = lambda x: x.fuse_key()
  by_key: dict = {}
  for i, la in enumerate(launches):
    if isinstance(la, _EnsLaunch):
      by_key.setdefault(key(la), []).append(i)
  fused_at = {idx[0]: idx for idx in by_key.values() if len(idx) > 1}
  if not fused_at:
    return launches, None
  gone = {i for idx in fused_at.values() for i in idx[1:]}
  old_off = np.concatenate([[0], np.cumsum([la.n_values for la in launches])])
  perm = np.empty(int(old_off[-1]), dtype=np.int64)
  out, new_off = [], 0
  for i, la in enumerate(launches):
    if i in gone:
      continue
    if i in fused_at:
      idx = fused_at[i]
      fused = _EnsFused([launches[j] for j in idx], device)
      inner = fused.perm()
      at = 0
      for j in idx:
        n = launches[j].n_values
        perm[old_off[j]:old_off[j] + n] = new_off + inner[at:at + n]
        at += n
      out.append(fused)
      new_off += fused.n_values
    else:
      perm[old_off[i]:old_off[i] + la.n_values] = new_off + np.arange(
          la.n_values, dtype=np.int64)
      out.append(la)
      new_off += la.n_values
  return out, perm


class _NotReplayable(Exception):
  pass


# why the last structures were not given a program (diagnostics, tests)
REASONS: list = []
# host time per replay of the native programs that have been closed
REPLAY_STATS: list = []


def _no(reason: str):
  REASONS.append(reason)
  del REASONS[:-16]
  return None


def _nan_equal(a: torch.Tensor, b: torch.Tensor) -> bool:
  if a.shape != b.shape or a.dtype != b.dtype:
    return False
  return bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all().item())


class _SinkGroup:
  """The result variables of ONE eval config of a program and how they reach
  that config's sink (evaluation.RunningMean / RunningConcat)."""

  def __init__(self, variables, device, time_dim, split_dim):
    self.variables = variables     # [(name, dims, shape, axis, src, round32)]
    self.device = device
    self.time_dim, self.split_dim = time_dim, split_dim
    # every variable's [element][time] source table, one after the other
    src = np.concatenate([v[4].ravel() for v in variables]).astype(np.int32)
    r32 = np.concatenate([v[5] for v in variables]).astype(np.uint8)
    self.n_time = variables[0][4].shape[1]
    self.n_out = int(r32.size)
    self.src = torch.as_tensor(src).to(device)
    self.round32 = torch.as_tensor(r32).to(device)
    self._targets: dict = {}  # lead-label bytes -> (ptr stamp, sum, count)
    self._lib = _lib.load()
    self._kept = None         # RunningConcat sinks: see _kept_tables

  def kept_structure(self, sink):
    """A sink that KEEPS the time steps (evaluation.RunningConcat): every
    (element, time step) entry of the source table has a destination of its
    own, row(time label, lead label) * rest size + rest index.  Structural, on
    the device once: which row of the chunk an entry goes to (`sel`), its rest
    size and rest offset in bytes.  None for a sink that averages."""
    if not getattr(sink, 'keeps_time', False):
      return None
    if self._kept is None:
      sel, rsz8, rest8, vid, groups = [], [], [], [], {}
      for i, (name, dims, shape, axis, src, r32) in enumerate(self.variables):
        key_dims = sink.key_dims(dims)
        sizes = [shape[dims.index(d)] for d in key_dims]
        if key_dims not in groups:
          groups[key_dims] = (sum(g[1] for g in groups.values()),
                              int(np.prod(sizes)))
        first = groups[key_dims][0]
        inner = [(d, n) for d, n in zip(dims, shape) if d != self.time_dim]
        grid = np.indices([n for _, n in inner]).reshape(len(inner), -1)
        lead = np.zeros(grid.shape[1], dtype=np.int64)
        rest = np.zeros(grid.shape[1], dtype=np.int64)
        rest_size = 1
        for (d, n), idx in zip(inner, grid):
          if d == self.split_dim:
            lead = idx
          else:
            rest, rest_size = rest * n + idx, rest_size * n
        n_lead = sizes[1] if len(sizes) == 2 else 1
        steps = np.arange(self.n_time, dtype=np.int64)
        # [element][time]: row (t, lead of the element) of the chunk's group
        sel.append((first + steps[None, :] * n_lead + lead[:, None]).ravel())
        rest8.append(np.repeat(8 * rest, self.n_time))
        rsz8.append(np.full(rest.size * self.n_time, 8 * rest_size, np.int64))
        vid.append(np.full(rest.size * self.n_time, i, np.int64))
      dev = lambda parts: torch.as_tensor(np.concatenate(parts)).to(self.device)
      each = np.concatenate([np.repeat(v[5], self.n_time)
                             for v in self.variables]).astype(np.uint8)
      self._kept = {'sel': dev(sel), 'rsz8': dev(rsz8), 'rest8': dev(rest8),
                    'vid': dev(vid), 'groups': groups, 'stamp': None,
                    'round': torch.as_tensor(each).to(self.device),
                    'max_rows': sum(g[1] for g in groups.values())}
      self._kept['sel32'] = self._kept['sel'].to(torch.int32)
    return self._kept

  def kept_rows(self, sink, forecast):
    """Per chunk, for a sink that keeps the time steps: (device table of every
    entry's row-0 sum address, the same for the counts, the storage rows of
    the chunk's label combinations -- a handful of integers)."""
    k = self.kept_structure(sink)
    rows = {g: sink.rows(forecast, g).ravel() for g in k['groups']}
    accs = []
    for name, dims, shape, axis, src, _ in self.variables:
      acc = sink.storage(name, dims, shape, None)
      acc.claim(rows[acc.key_dims].tolist(), len(sink._rows[acc.key_dims]),
                name)
      accs.append(acc)
    stamp = tuple(a.total.data_ptr() for a in accs)
    if stamp != k['stamp']:   # first chunk, or a storage has grown
      sums = engine.upload_table(np.array(stamp, dtype=np.int64), self.device,
                                 cache=False)
      counts = engine.upload_table(
          np.array([a.count.data_ptr() for a in accs], dtype=np.int64),
          self.device, cache=False)
      k['sum0'] = sums[k['vid']] + k['rest8']
      k['count0'] = counts[k['vid']] + k['rest8']
      k['stamp'] = stamp
    mine = (next(iter(rows.values())) if len(rows) == 1
            else np.concatenate(list(rows.values())))
    return k['sum0'], k['count0'], np.ascontiguousarray(mine, dtype=np.int64)

  def _kept_tables(self, sink, forecast):
    """The Python replay's form of the above: the destination of every entry
    worked out with four small device ops."""
    sum0, count0, rows = self.kept_rows(sink, forecast)
    k = self._kept
    chunk_rows = engine.upload_table(rows, self.device, cache=False)
    offset = chunk_rows[k['sel']] * k['rsz8']
    return sum0 + offset, count0 + offset, k['round']

  def reset(self):
    """Forget cached accumulator addresses (the accumulators were replaced)."""
    self._targets.clear()
    if self._kept is not None:
      self._kept['stamp'] = None

  # -- accumulator addresses --------------------------------------------------
  def _accumulator_tables(self, mean, labels):
    """Device tables of the (sum, count) ADDRESS of every output element for
    chunks that carry the split-dim (lead) labels `labels`; cached per label
    set, rebuilt when an accumulator has been reallocated (grown)."""
    key = None if labels is None else (labels.dtype.str, labels.tobytes())
    accs = [mean._acc[v[0]] for v in self.variables]
    hit = self._targets.get(key)
    if hit is not None and hit[0] == tuple(a.total.data_ptr() for a in accs):
      return hit[1], hit[2]
    sums, counts = [], []
    for (name, dims, shape, axis, src, _), acc in zip(self.variables, accs):
      if acc.split is not None:
        rows = acc.rows(labels)   # (may grow the accumulator: pointers below)
        acc.shape = tuple(n for d, n in zip(dims, shape) if d != self.time_dim)
        dst = acc.destinations(rows)
      else:
        dst = np.arange(src.shape[0], dtype=np.int64)
      if dst.size != src.shape[0]:
        raise ValueError(f'{name}: accumulator layout changed')
      sums.append(acc.total.data_ptr() + 8 * dst)
      counts.append(acc.count.data_ptr() + 8 * dst)
    stamp = tuple(a.total.data_ptr() for a in accs)
    d_sum = engine.upload_table(np.concatenate(sums), self.device, cache=False)
    d_cnt = engine.upload_table(np.concatenate(counts), self.device,
                                cache=False)
    if len(self._targets) > 256:
      self._targets.clear()
    self._targets[key] = (stamp, d_sum, d_cnt)
    return d_sum, d_cnt

  # -- one chunk's values into the sink -----------------------------------------
  def accumulate(self, arena, forecast: xl.Dataset, mean, stream) -> None:
    if getattr(mean, 'keeps_time', False):
      # one destination per (element, time step): 0 + value, exact
      d_sum, d_cnt, round_each = self._kept_tables(mean, forecast)
      status = self._lib.wb2_gather_accumulate(
          arena.data_ptr(), self.src.data_ptr(), round_each.data_ptr(),
          self.n_out * self.n_time, 1, 0, d_sum.data_ptr(), d_cnt.data_ptr(),
          stream)
      if status != 0:
        _lib.check(status, 'wb2_gather_accumulate')
      return
    labels = None
    if self.split_dim is not None:
      labels = np.asarray(forecast.coords[self.split_dim])
    d_sum, d_cnt = self._accumulator_tables(mean, labels)
    status = self._lib.wb2_gather_accumulate(
        arena.data_ptr(), self.src.data_ptr(), self.round32.data_ptr(),
        self.n_out, self.n_time, int(mean.skipna), d_sum.data_ptr(),
        d_cnt.data_ptr(), stream)
    if status != 0:
      _lib.check(status, 'wb2_gather_accumulate')


class _Native:
  """The launches and sinks of a ChunkProgram inside libwb2hip.so
  (csrc/program.cpp): a chunk is replayed by ONE wb2_program_replay call -- the
  base pointers of the chunk's arrays, the climatology slab of every valid
  time, the sinks' accumulator tables go in; the address table, its upload,
  every launch (small latency-bound passes on a second stream) and the
  accumulation happen on the C side.  Same kernels, same order: the bits of the
  Python replay (`WB2HIP_NATIVE_REPLAY=0` keeps that one).

  Raises _NotReplayable for what the C side does not cover (gathers from lazy
  containers): the program then replays from Python."""

  def __init__(self, launches, groups, arena, device, means):
    import ctypes
    lib = self._lib = _lib.load()
    self.device = device
    handle = ctypes.c_void_p()
    _lib.check(lib.wb2_program_create(ctypes.byref(handle)),
               'wb2_program_create')
    self._handle = handle
    self._finalizer = weakref.finalize(self, _Native._close, lib,
                                       handle.value)
    # ---- pointer sources, shared by the launches
    self.getters: list = []     # (index, getter) of the chunk's own arrays
    index_of: dict = {}
    fixed: dict = {}
    keep = self._keep = []

    def source(key, getter):
      if key not in index_of:
        index_of[key] = len(index_of)
        if key[0] == 'r':       # resident: the same pointer for every chunk
          fixed[index_of[key]] = getter(None, None).data_ptr()
        else:
          self.getters.append((index_of[key], getter))
      return index_of[key]
    # ---- gathered inputs: one value block per (climatology, dims)
    self.values: list = []      # values(forecast, memo) per block
    block_of: dict = {}
    n_values = 0
    off = 0
    for la in launches:
      if isinstance(la, _EnsPass):
        # K3 + its fold: (member 0's slab, truth slab) of every outer index
        remap = np.array([source(k, g) for k, g in zip(la.source_keys,
                                                       la.sources)],
                         dtype=np.int32)
        slot = np.ascontiguousarray(remap[la.slot], dtype=np.int32)
        rel = np.ascontiguousarray(la.rel, dtype=np.int64)
        pl = la.plan
        tables, held = engine.plan_tables(
            pl, lib.wb2_ens_tile_cols(pl.n_col), field=pl.wfield)
        _lib.check(lib.wb2_program_add_ens_launch(
            handle, ctypes.byref(tables), engine._DTYPES[la.dtype],
            int(la.skipna), la.n_member, la.member_stride, la.n_total,
            slot.ctypes.data, rel.ctypes.data, la.partials.data_ptr(), off),
                   'wb2_program_add_ens_launch')
        keep.append((la, tables, held))
        off += la.n_values
        continue
      remap = np.array([source(k, g) for k, g in zip(la.source_keys,
                                                     la.sources)],
                       dtype=np.int32)
      slot = np.ascontiguousarray(remap[la.slot], dtype=np.int32)
      rel = np.ascontiguousarray(la.rel, dtype=np.int64)
      step = la.step
      # SEEPS passes beside the streaming launch: one slab per chunk (two
      # latency-bound launches chunk by chunk) and VALU-bound at any size --
      # next to an HBM-bound kernel they cost next to nothing
      side = int(la.mode == _lib.MODE_SEEPS)
      wind = getattr(step, 'wind_partials', None)
      _lib.check(lib.wb2_program_add_launch(
          handle, step._tables_ref, la.mode, step.code, int(la.skipna),
          la.n_in, la.n_total, la.n_pair, slot.ctypes.data, rel.ctypes.data,
          step.partials.data_ptr(), _lib.ptr(wind) or None, off, side),
                 'wb2_program_add_launch')
      keep.append(step)
      for j, first, n, x, _, gather in la.dynamic:
        if gather is None or not isinstance(x, torch.Tensor) or not (
            x.is_contiguous()):
          raise _NotReplayable('a gather the native replay does not cover')
        block = block_of.get(gather['key'])
        n_cell = int(gather['cell'].max()) + 1 if gather['cell'].size else 0
        if block is None:
          block = block_of[gather['key']] = (n_values, n_cell)
          self.values.append(gather['values'])
          n_values += n_cell
        if block[1] < n_cell:
          raise _NotReplayable('gathers of one climatology differ in cells')
        src = source(('r', id(x)), lambda f, t_, x=x: x)
        cell = np.ascontiguousarray(gather['cell'], dtype=np.int32)
        base = np.ascontiguousarray(gather['base'], dtype=np.int64)
        if cell.size != n or base.size != n:
          raise _NotReplayable('gather tables do not cover the launch slabs')
        slab_bytes = la.plan.n_row * la.plan.n_col * x.element_size()
        _lib.check(lib.wb2_program_add_gather(
            handle, j, first, n, src, slab_bytes, block[0], cell.ctypes.data,
            base.ctypes.data), 'wb2_program_add_gather')
        keep.append(x)
      off += la.n_values
    self.n_values = n_values
    # ---- sinks
    self.groups = groups
    for g, mean in zip(groups, means):
      kept = g.kept_structure(mean)
      if kept is None:
        _lib.check(lib.wb2_program_add_sink(
            handle, g.src.data_ptr(), g.round32.data_ptr(), g.n_out, g.n_time,
            int(bool(mean.skipna)), None, None, 0), 'wb2_program_add_sink')
      else:
        _lib.check(lib.wb2_program_add_sink(
            handle, g.src.data_ptr(), kept['round'].data_ptr(),
            g.n_out * g.n_time, 1, 0, kept['sel32'].data_ptr(),
            kept['rsz8'].data_ptr(), kept['max_rows']),
                   'wb2_program_add_sink')
    self.n_ptrs = len(index_of)
    _lib.check(lib.wb2_program_finalize(handle, arena.data_ptr(), self.n_ptrs,
                                        n_values), 'wb2_program_finalize')
    self.ptrs = np.zeros(max(self.n_ptrs, 1), dtype=np.int64)
    for i, p in fixed.items():
      self.ptrs[i] = p
    self._ptrs_at = self.ptrs.ctypes.data
    self.sink_args = np.zeros((max(len(groups), 1), 3), dtype=np.int64)
    self._sink_at = self.sink_args.ctypes.data
    self._replay = lib.wb2_program_replay
    self._no_values = np.zeros(1, dtype=np.int64)
    self._values_of: dict = {}

  @staticmethod
  def _stats(lib, handle) -> dict:
    import ctypes
    sec = (ctypes.c_double * 5)()
    n = ctypes.c_int64()
    _lib.check(lib.wb2_program_stats(handle, sec, ctypes.byref(n)),
               'wb2_program_stats')
    per = max(n.value, 1)
    names = ('wait_for_slot', 'fill_table', 'copy', 'launches', 'sinks')
    out = {k: 1e3 * v / per for k, v in zip(names, sec)}
    out['replays'] = n.value
    return out

  def stats(self) -> dict:
    """Host milliseconds per replay by phase (wb2_program_stats)."""
    return self._stats(self._lib, self._handle)

  @staticmethod
  def _close(lib, handle):
    try:   # what the program's replays cost the host (diagnostics, tools)
      REPLAY_STATS.append(_Native._stats(lib, handle))
      del REPLAY_STATS[:-16]
    except Exception:
      pass
    lib.wb2_program_destroy(handle)

  def run(self, forecast, truth, means, stream) -> None:
    ptrs = self.ptrs
    for i, g in self.getters:
      ptrs[i] = g(forecast, truth).data_ptr()
    if self.values:
      # the climatology slabs of the chunk's valid times: label work on a few
      # time stamps, remembered per time stamp set (a valid time comes back
      # with every lead that reaches it)
      from weatherbench2_amd import metrics as gm
      vt, dims = gm._valid_times(forecast)
      key = vt.tobytes()
      values = self._values_of.get(key)
      if values is None:
        memo: dict = {'valid_times': (forecast, vt, dims)}
        values = (self.values[0](forecast, memo) if len(self.values) == 1
                  else np.concatenate([v(forecast, memo)
                                       for v in self.values]))
        if len(self._values_of) >= 4096:
          self._values_of.clear()
        self._values_of[key] = values
    else:
      values = self._no_values
    rows, args = None, self.sink_args
    for k, (group, mean) in enumerate(zip(self.groups, means)):
      if getattr(mean, 'keeps_time', False):
        d_sum, d_cnt, mine = group.kept_rows(mean, forecast)
        rows = mine if rows is None else np.concatenate([rows, mine])
        args[k, 2] = mine.size
      else:
        labels = None
        if group.split_dim is not None:
          labels = np.asarray(forecast.coords[group.split_dim])
        d_sum, d_cnt = group._accumulator_tables(mean, labels)
        # (the program was built for this sink's NaN rule)
      args[k, 0], args[k, 1] = d_sum.data_ptr(), d_cnt.data_ptr()
    hook = engine._LAUNCH_HOOK
    if hook is not None:   # (brackets every launch of the chunk + the sinks)
      hook('begin', 'stream_partials')
    status = self._replay(
        self._handle, self._ptrs_at, self.n_ptrs, values.ctypes.data,
        self.n_values, self._sink_at,
        None if rows is None else rows.ctypes.data,
        0 if rows is None else rows.size, stream)
    if hook is not None:
      hook('end', 'stream_partials')
    if status != 0:
      _lib.check(status, 'wb2_program_replay')


class ChunkProgram:
  """The replayable form of one chunk structure (see the module docstring):
  the launches of the loop(s) over it, and one `_SinkGroup` per eval config
  that was evaluated on the chunk (several configs -- `deterministic` and
  `deterministic_temporal` of the documented command line -- share the
  launches: the chunk is read once)."""

  def __init__(self, launches, groups, arena, device, time_dim, split_dim,
               means=None):
    self.launches = launches       # [_Launch | _EnsLaunch]
    self.groups = groups           # [_SinkGroup], one per config
    self.arena = arena             # float64 device tensor: launch outputs
    self.device = device
    self.time_dim, self.split_dim = time_dim, split_dim
    off = 0
    self.slices = []
    for la in launches:   # (flat: a launch may write several output blocks)
      self.slices.append(arena[off:off + la.n_values])
      off += la.n_values
    # the whole replay behind one C-ABI call where the C side covers it
    self.native = None
    if means is not None and os.environ.get('WB2HIP_NATIVE_REPLAY',
                                            '1') != '0':
      try:
        self.native = _Native(launches, groups, arena, device, means)
      except _NotReplayable as e:
        _no(f'python replay: {e}')

  def reset(self):
    for g in self.groups:
      g.reset()

  def run(self, forecast: xl.Dataset, truth: xl.Dataset, means) -> None:
    """`means`: the sink of every config, in the order of the build (a single
    sink for a single config)."""
    if not isinstance(means, (list, tuple)):
      means = [means]
    stream = engine.current_stream_ptr(self.device)
    if self.native is not None:
      self.native.run(forecast, truth, means, stream)
      return
    _run_launches(self.launches, self.slices, forecast, truth, self.device,
                  stream)
    for group, mean in zip(self.groups, means):
      group.accumulate(self.arena, forecast, mean, stream)


def build(first: Recorder, forecast: xl.Dataset, truth: xl.Dataset, results,
          means, loop) -> t.Optional[ChunkProgram]:
  """The program of the structure of (forecast, truth), from the recorder that
  watched the generic pass over it; `results` = what that pass returned (one
  Dataset per eval config, already added to the config's sink in `means`), or
  None when the structure cannot be replayed.  `loop()` runs the generic pass
  again (under the probe recorder) and returns the same list.  A single
  Dataset / sink / loop result stands for a list of one."""
  if not isinstance(means, (list, tuple)):
    one = loop
    results, means, loop = [results], [means], (lambda: [one()])
  try:
    return _build(first, forecast, truth, results, means, loop)
  except _NotReplayable as e:
    return _no(f'not replayable: {e}')


def _build(first, forecast, truth, results, means, loop):
  launches_rec = [l for l in first.launches if 'plan' in l]
  if not launches_rec:
    return _no('no launch recorded')
  device = launches_rec[0]['plan'].device
  outputs = lambda l: [l['metrics']] + (
      [l['wind_metrics']] if l.get('wind_metrics') is not None else [])
  total = sum(int(x.numel()) for l in launches_rec for x in outputs(l))
  if total >= _MAX_ELEMENTS:
    return _no('too many output elements')
  time_dim, split_dim = means[0].dim, means[0].split_dim
  fmap = {id(v.data): ('f', k) for k, v in forecast.data_vars.items()}
  tmap = {id(v.data): ('t', k) for k, v in truth.data_vars.items()}
  if any(l.get('kind') == 'unsupported' for l in first.launches):
    return _no('a pass the programs do not cover (maps / gathered members)')
  for l in launches_rec:
    l['tables'] = first.tables
  launches = [(_EnsLaunch if l.get('kind') == 'ens' else _Launch)(
      l, fmap, tmap, device) for l in launches_rec]
  # ---- probe: which output element does every result element show?
  with Recorder(probe=True) as probe:
    shown_all = [xl.as_dataset(r) for r in loop()]
  if [l['n'] for l in probe.launches] != [la.n_values for la in launches]:
    return _no('probe saw other launches')
  flat_real = torch.cat([x.reshape(-1) for l in launches_rec
                         for x in outputs(l)])
  per_sink = []
  for shown, result, mean in zip(shown_all, results, means):
    if (mean.dim, mean.split_dim) != (time_dim, split_dim):
      return _no('the sinks disagree on the time / lead dims')
    variables = _group_variables(shown, xl.as_dataset(result), mean, forecast,
                                 flat_real, total, time_dim)
    if variables is None:
      return None
    per_sink.append(variables)
  # the ensemble passes of a chunk's variables become one launch per member
  # stride (same kernels, per-slab fold: the separate passes' bits -- checked
  # on this chunk all the same)
  singles = launches
  launches, perm = _fuse_ensemble_launches(launches, device)
  if perm is not None:
    trial = torch.empty((total,), dtype=torch.float64, device=device)
    stream = engine.current_stream_ptr(device)
    off = 0
    for la in launches:
      if isinstance(la, _EnsFused):
        _run_launches([la], [trial[off:off + la.n_values]], forecast, truth,
                      device, stream)
      off += la.n_values
    fused_at = torch.as_tensor(perm, device=device)
    mine = torch.zeros_like(trial, dtype=torch.bool)
    off = 0
    for la in launches:
      if isinstance(la, _EnsFused):
        mine[off:off + la.n_values] = True
      off += la.n_values
    pick = mine[fused_at]   # old elements that the fused launches produce
    if not _nan_equal(trial[fused_at][pick], flat_real[pick]):
      _no('fused ensemble launch differs in bits: separate passes kept')
      launches, perm = singles, None
  if perm is not None:
    per_sink = [[(name, dims, shape, axis,
                  np.where(src >= 0, perm[np.maximum(src, 0)], -1), r32)
                 for name, dims, shape, axis, src, r32 in variables]
                for variables in per_sink]
  groups = [_SinkGroup(variables, device, time_dim, split_dim)
            for variables in per_sink]
  arena = torch.empty((total,), dtype=torch.float64, device=device)
  return ChunkProgram(launches, groups, arena, device, time_dim, split_dim,
                      means)


def _group_variables(shown, result, mean, forecast, flat_real, total, time_dim):
  """[(name, dims, shape, time axis, source table, float32 flags)] of one
  config's result, or None (reason noted) if it is not a rearrangement of the
  launches' outputs."""
  variables = []
  for name, da in shown.data_vars.items():
    real = result.data_vars.get(name)
    if real is None or not isinstance(da.data, torch.Tensor) or not isinstance(
        real.data, torch.Tensor) or tuple(real.dims) != tuple(da.dims):
      return _no('result variable missing / not a device tensor / other dims')
    if time_dim not in da.dims:
      return _no('no time dim in a result variable')
    axis = da.dims.index(time_dim)
    v = da.data.to(torch.float64)
    k = torch.floor(v)
    fill = torch.isnan(v)
    u = k + _THIRD                # what launch element k holds, in float64
    exact = v == u
    # exact: the element is an output element as it is; otherwise it must be
    # that element rounded to float32 (k + 1/3 in float32, widened again)
    as32 = u.to(torch.float32).to(torch.float64)
    rounded = (~exact) & (v == as32)
    if not bool((fill | exact | rounded).all().item()):
      return _no('values were computed, not moved')
    src = torch.where(fill, torch.full_like(k, -1.0), k).to(torch.int64)
    if bool((src >= total).any().item()):
      return _no('source index out of range')
    # verify on the real first chunk
    picked = flat_real[src.clamp(min=0)]
    picked = torch.where(rounded,
                         picked.to(torch.float32).to(torch.float64), picked)
    picked = torch.where(fill, torch.full_like(picked, float('nan')), picked)
    if not _nan_equal(picked.to(real.data.dtype), real.data):
      return _no('mapping does not reproduce the first chunk')
    if real.data.dtype == torch.float32 and bool(exact.any().item()):
      return _no('float32 result without float32 rounding')
    moved = src.movedim(axis, -1)          # [..., time]
    n_time = moved.shape[-1]
    # an element's time steps come from one metric row: one flag per element
    flags = rounded.movedim(axis, -1).reshape(-1, n_time)
    if not bool((flags == flags[:, :1]).all().item()):
      return _no('time steps of an element differ in rounding')
    variables.append((name, tuple(da.dims), tuple(da.shape), axis,
                      moved.reshape(-1, n_time).cpu().numpy(),
                      flags[:, 0].cpu().numpy()))
    acc = mean._acc.get(name)
    kept = getattr(mean, 'keeps_time', False)
    if acc is None or acc.dims != tuple(
        d for d in da.dims if kept or d != time_dim):
      return _no('accumulator layout differs')
    if kept:
      # the program files a chunk under the FORECAST's labels
      for d in acc.key_dims:
        mine, theirs = forecast.coords.get(d), result.coords.get(d)
        if mine is None or theirs is None or isinstance(
            mine, xl.DataArray) or isinstance(theirs, xl.DataArray) or not (
                np.array_equal(np.asarray(mine), np.asarray(theirs))):
          return _no(f'result {d!r} labels are not the forecast chunk\'s')
  if set(result.data_vars) != {v[0] for v in variables}:
    return _no('result variables differ')
  if len({v[4].shape[1] for v in variables}) != 1:
    return _no('time lengths differ')
  return variables
