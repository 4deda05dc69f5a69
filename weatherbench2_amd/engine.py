"""Launch layer: torch tensors (device memory, streams) -> C-ABI kernels.

PyTorch is plumbing here: it owns HBM allocations and the current HIP stream;
every FLOP of the hot path runs in libwb2hip.so.
"""
from __future__ import annotations

import os
import threading
import typing as t

import numpy as np
import torch

from weatherbench2_amd import _lib
from weatherbench2_amd.plan import ReductionPlan

_DTYPES = {torch.float32: _lib.WB2_F32, torch.float64: _lib.WB2_F64}
_FIELD_F32_MODES = (_lib.MODE_DET, _lib.MODE_DET_ACC, _lib.MODE_WIND)

_STAGED_UPLOAD_MIN_BYTES = 1 << 20

# Profiling hook (bench.py, tools/): called as hook('begin' | 'end', kernel) on
# the launch stream around the dominant kernel of a pass.  None in production.
_LAUNCH_HOOK = None


def set_launch_hook(hook):
  """Installs (or clears, with None) the profiling hook; returns the old one."""
  global _LAUNCH_HOOK
  old, _LAUNCH_HOOK = _LAUNCH_HOOK, hook
  return old


def current_stream_ptr(device) -> int:
  return torch.cuda.current_stream(device).cuda_stream


class _ThreadStreams(threading.local):

  def __init__(self):
    self.streams: dict = {}
    self.disabled = False


_THREAD_STREAMS = _ThreadStreams()


def disable_thread_stream() -> None:
  """The calling thread keeps the default / caller's stream for good (IO helper
  threads -- evaluation._prefetched -- whose uploads the main thread consumes:
  a private stream there would order nothing with the consumer)."""
  _THREAD_STREAMS.disabled = True


def _own_stream(device: torch.device):
  return _THREAD_STREAMS.streams.get((device.type, device.index))


def _adopt_thread_stream(device: torch.device) -> None:
  """Worker threads (Beam's DirectRunner calls compute_chunk from several,
  evaluation.py:583-599; ctypes releases the GIL) get a HIP stream of their own
  as their torch *current* stream the first time they reach the GPU path:
  everything such a thread launches, allocates and reads back is then ordered
  on its stream, and the passes of different threads overlap on the device
  instead of queueing on the shared default stream.

  Hand-offs between threads are ordered through the default stream:
    * every ENTRY to the GPU path (`require_gpu`) makes the own stream wait for
      what the default stream has queued -- inputs the main thread produced or
      made resident at any time before, and whatever other workers published;
    * `publish_thread_stream` (the exit of the outermost chunk scope, i.e. the
      moment results leave the pass) makes the default stream wait for the own
      stream, so a result read on the main thread, or on another worker (whose
      reads wait for the default stream, `order_read`), sees finished kernels.
  The main thread keeps the caller's stream.  WB2HIP_THREAD_STREAMS=0 disables."""
  if threading.current_thread() is threading.main_thread():
    return
  st = _THREAD_STREAMS
  key = (device.type, device.index)
  if key not in st.streams:
    if st.disabled or os.environ.get('WB2HIP_THREAD_STREAMS', '1') == '0':
      st.streams[key] = None
    else:
      own = torch.cuda.Stream(device=device)
      torch.cuda.set_stream(own)
      st.streams[key] = own
  own = st.streams[key]
  if own is not None:
    own.wait_stream(torch.cuda.default_stream(device))


def publish_thread_stream() -> None:
  """Results of this (worker) thread's stream become visible to the default
  stream: called when the outermost chunk scope of a pass exits."""
  for (kind, index), own in _THREAD_STREAMS.streams.items():
    if own is not None:
      torch.cuda.default_stream(torch.device(kind, index)).wait_stream(own)


def order_read(tensor: torch.Tensor) -> None:
  """Before a device result is read (`.values`, RunningMean.add): a reader on a
  private stream waits for what has been published on the default stream, and
  the read is recorded with the allocator (cross-stream block reuse)."""
  if not tensor.is_cuda:
    return
  dev = tensor.device
  cur = torch.cuda.current_stream(dev)
  default = torch.cuda.default_stream(dev)
  if cur != default:
    cur.wait_stream(default)
  # The reader's stream is (in general) not the stream the result was
  # allocated on -- a worker's private stream -- and `wait_stream` orders
  # visibility only: the caching allocator hands a freed block back to its
  # OWN stream at once, where the worker's next pass could overwrite it while
  # this read is still queued here.  Recording the use makes the allocator
  # wait for this stream before it reuses the block (a no-op when the streams
  # are the same).
  tensor.record_stream(cur)


def require_gpu() -> torch.device:
  if not torch.cuda.is_available():
    raise _lib.Wb2HipError(
        'no HIP device visible: the MI355X path has no CPU fallback')
  device = torch.device('cuda', torch.cuda.current_device())
  _adopt_thread_stream(device)
  return device


def as_device_tensor(x, device, dtype=None) -> torch.Tensor:
  """numpy / torch / DLPack-capable array -> contiguous tensor on `device`."""
  if isinstance(x, torch.Tensor):
    ten = x
  elif isinstance(x, np.ndarray):
    if not x.dtype.isnative:
      x = x.astype(x.dtype.newbyteorder('='))
    if (device.type == 'cuda' and x.nbytes >= _STAGED_UPLOAD_MIN_BYTES
        and (dtype is None or torch.from_numpy(np.empty(0, x.dtype)).dtype
             == dtype)):
      # big host chunks cross PCIe through the pinned staging ring on the copy
      # stream (feeder.py) instead of a pageable, synchronous cudaMemcpy
      from weatherbench2_amd import feeder
      return feeder.upload(x, device)
    ten = torch.from_numpy(np.ascontiguousarray(x))
  elif hasattr(x, '__dlpack__'):
    ten = torch.from_dlpack(x)
  else:
    ten = torch.as_tensor(np.asarray(x))
  if dtype is not None and ten.dtype != dtype:
    ten = ten.to(dtype)
  if ten.device != device:
    ten = ten.to(device, non_blocking=True)
  return ten.contiguous()


def digest(buf: np.ndarray) -> bytes:
  """Content digest of a contiguous numeric array (xxh3 when importable --
  GB/s --, blake2b otherwise)."""
  buf = np.ascontiguousarray(buf)
  if buf.size == 0:  # memoryview cannot cast shapes with zeros
    return b'empty:' + repr((buf.shape, buf.dtype.str)).encode()
  raw = memoryview(buf.reshape(-1)).cast('B')
  try:
    import xxhash
    return xxhash.xxh3_128_digest(raw)
  except ImportError:
    import hashlib
    return hashlib.blake2b(raw, digest_size=16).digest()


class _TableUploads(threading.local):
  """Per-thread, per-device: content cache of uploaded int64 tables + a small
  pinned ring for the misses."""

  def __init__(self):
    self.cache: dict = {}   # (device, shape, digest) -> (device tensor, event)
    self.ring: dict = {}    # device -> [pinned slots, events, next]


_TABLES = _TableUploads()
_TABLE_SLOT_BYTES = 1 << 18
# A slot is reused when the copy out of it has run: with 8 slots the host was
# held at 8 tables ahead of the device -- a window of the map suite uploads 20
_TABLE_SLOTS = int(os.environ.get('WB2HIP_TABLE_SLOTS', 32))


def upload_table(table: np.ndarray, device, cache: bool = True) -> torch.Tensor:
  """int64 slab table -> device tensor WITHOUT stalling the queue.

  `torch.from_numpy(t).to(device)` is a synchronous pageable copy: with kernels
  queued ahead the host blocks until the GPU has drained, once per table and
  call.  Tables recur (same chunk geometry, same valid times), so they are
  cached by a digest of their content; a miss goes through a pinned slot with
  an asynchronous copy on the current stream of `device`, and the event of that
  copy stays with the cache entry: a hit under a different current stream
  (`torch.cuda.stream(...)`) waits for it before the table is read."""
  table = np.ascontiguousarray(table, dtype=np.int64)
  device = torch.device(device)
  st = _TABLES
  key = None
  if cache:  # (cache=False: a table that will not come again -- addresses)
    key = (str(device), table.shape, digest(table))
    hit = st.cache.get(key)
    if hit is not None:
      dev, ev = hit
      if ev is not None:
        torch.cuda.current_stream(device).wait_event(ev)
      return dev
  nbytes = table.nbytes
  ev = None
  if nbytes > _TABLE_SLOT_BYTES or nbytes == 0 or device.type != 'cuda':
    dev = torch.from_numpy(table).to(device)  # synchronous: complete on return
  else:
    ring = st.ring.get(str(device))
    if ring is None:
      ring = st.ring[str(device)] = [
          [torch.empty(_TABLE_SLOT_BYTES // 8, dtype=torch.int64).pin_memory()
           for _ in range(_TABLE_SLOTS)], [None] * _TABLE_SLOTS, 0]
    slots, events, nxt = ring
    ring[2] = (nxt + 1) % len(slots)
    if events[nxt] is not None:
      events[nxt].synchronize()
    slot = slots[nxt][:table.size]
    slot.numpy()[...] = table.reshape(-1)
    # (the current stream of `device`: allocation and copy are queued on it)
    dev = torch.empty(table.shape, dtype=torch.int64, device=device)
    dev.view(-1).copy_(slot, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    events[nxt] = ev
  if key is not None:
    if len(st.cache) >= 512:
      st.cache.clear()
    st.cache[key] = (dev, ev)
  return dev


def _slabs_intact(x: torch.Tensor) -> bool:
  """Contiguous, or a strided view whose 2-D slabs are contiguous and whose
  outer strides are whole slabs (addressed through a slab table)."""
  if x.is_contiguous():
    return True
  if x.dim() < 2:
    return False
  st = x.stride()
  se = x.shape[-1] * x.shape[-2]
  return st[-1] == 1 and st[-2] == x.shape[-1] and all(
      s >= 0 and s % se == 0 for s in st[:-2])


def stream_reduce(plan: ReductionPlan, mode: int,
                  inputs: t.Sequence[torch.Tensor],
                  slabs: t.Sequence[t.Optional[torch.Tensor]], n_outer: int,
                  skipna: bool, want_sums: bool = False,
                  aux: t.Optional[torch.Tensor] = None, scalar: float = 0.0):
  """Runs K1 + K2.  `inputs[i]` is [n_slab_i, n_row, n_col] on the plan's device.

  Returns (metrics[NMETRIC, n_region, n_outer], sums[n_outer, n_region, K] or
  None) as float64 device tensors; the call is asynchronous on the current
  stream.
  """
  dev = plan.device
  dtype = inputs[0].dtype
  if dtype not in _DTYPES:
    raise TypeError(f'unsupported dtype {dtype}')
  for x in inputs:
    if x.dtype != dtype or x.device != dev or not _slabs_intact(x):
      raise ValueError('inputs must share dtype/device and keep every '
                       '(n_row, n_col) slab contiguous')
    if tuple(x.shape[-2:]) != (plan.n_row, plan.n_col):
      raise ValueError(f'slab shape {tuple(x.shape[-2:])} != plan '
                       f'({plan.n_row}, {plan.n_col})')
  for s, x in zip(slabs, inputs):
    if s is None and (not x.is_contiguous() or
                      x.numel() != n_outer * plan.n_row * plan.n_col):
      raise ValueError('an input without a slab table must be n_outer '
                       'contiguous slabs')
    if s is not None and (s.dtype != torch.int64 or s.numel() != n_outer):
      raise ValueError('slab tables are int64[n_outer]')
  aligned = all(x.data_ptr() % 16 == 0 for x in inputs)
  return _stream_launch(plan, mode, dtype, n_outer, skipna, want_sums, aux,
                        scalar, aligned, inputs=inputs, slabs=slabs)


def stream_reduce_addr(plan: ReductionPlan, mode: int, dtype: torch.dtype,
                       addr: t.Sequence[torch.Tensor], aligned16: bool,
                       n_outer: int, skipna: bool, want_sums: bool = False,
                       aux: t.Optional[torch.Tensor] = None,
                       scalar: float = 0.0):
  """K1 + K2 over slabs that live in different allocations (the variables of
  a chunk, several consecutive chunks): `addr[i]` is an int64[n_outer] device
  tensor holding the byte address of input i's slab for every outer slab
  (wb2_stream_partials_addr).  The caller keeps the memory behind the
  addresses alive until the launch has been enqueued on the current stream
  (torch's allocator is stream-ordered).  Same return value as stream_reduce."""
  if dtype not in _DTYPES:
    raise TypeError(f'unsupported dtype {dtype}')
  for a in addr:
    if (a.dtype != torch.int64 or a.device != plan.device or
        a.numel() != n_outer or not a.is_contiguous()):
      raise ValueError('address tables are contiguous int64[n_outer] on the '
                       'plan device')
  return _stream_launch(plan, mode, dtype, n_outer, skipna, want_sums, aux,
                        scalar, bool(aligned16), addr=addr)


def _stream_launch(plan, mode, dtype, n_outer, skipna, want_sums, aux, scalar,
                   aligned, inputs=None, slabs=None, addr=None):
  lib = _lib.load()
  dev = plan.device
  code = _DTYPES[dtype]
  k = lib.wb2_num_slots(mode, int(skipna))
  # the 2-D weight field as float32 where its values are float32 numbers and
  # the launch has the instantiation (float32 inputs, DET / DET_ACC / WIND):
  # the same bits, half the field bytes per point
  field, field_code = plan.wfield, _lib.WB2_F64
  if (dtype == torch.float32 and mode in _FIELD_F32_MODES and
      getattr(plan, 'wfield32', None) is not None and
      os.environ.get('WB2HIP_FIELD_F32', '1') != '0'):
    field, field_code = plan.wfield32, _lib.WB2_F32
  aligned = aligned and (field is None or field.data_ptr() % 16 == 0)
  tile = lib.wb2_tile_cols_ex(mode, code, int(skipna),
                              int(plan.wfield is not None), plan.n_col,
                              int(aligned))
  n_ctile = -(-plan.n_col // tile)
  stream = current_stream_ptr(dev)
  seg_eoff, n_ts = plan.seg_entries(tile)
  partials = torch.empty((n_outer, plan.n_chunk, plan.nwf, n_ts, k),
                         dtype=torch.float64, device=dev)
  if _LAUNCH_HOOK is not None:
    _LAUNCH_HOOK('begin', 'stream_partials')
  tail = (n_outer, plan.n_row, plan.n_col, _lib.ptr(plan.w_row),
          _lib.ptr(plan.w_col), _lib.ptr(field), field_code, _lib.ptr(aux),
          float(scalar), _lib.ptr(plan.chunk_row0),
          _lib.ptr(plan.chunk_nrow), plan.n_chunk, n_ctile,
          _lib.ptr(plan.seg_col0), _lib.ptr(seg_eoff), plan.n_seg, n_ts,
          _lib.ptr(partials), stream)
  if addr is not None:
    _lib.check(lib.wb2_stream_partials_addr(
        mode, code, int(skipna), _lib.ptr_array(addr), int(aligned), *tail),
               'wb2_stream_partials_addr')
  else:
    _lib.check(lib.wb2_stream_partials_ex(
        mode, code, int(skipna), _lib.ptr_array(inputs), _lib.ptr_array(slabs),
        *tail), 'wb2_stream_partials')
  if _LAUNCH_HOOK is not None:
    _LAUNCH_HOOK('end', 'stream_partials')
  metrics = torch.empty((_lib.GENERIC_KQ.get(mode, _lib.NMETRIC),
                         plan.n_region, n_outer),
                        dtype=torch.float64, device=dev)
  sums = (torch.empty((n_outer, plan.n_region, k), dtype=torch.float64,
                      device=dev) if want_sums else None)
  _lib.check(lib.wb2_det_combine(
      mode, int(skipna), _lib.ptr(partials), n_outer, plan.n_chunk, plan.nwf,
      plan.n_seg, _lib.ptr(seg_eoff), n_ts, _lib.ptr(plan.band_chunk0),
      plan.n_band,
      _lib.ptr(plan.coef_band), _lib.ptr(plan.coef_seg),
      _lib.ptr(plan.region_wf), _lib.ptr(plan.region_wsum), plan.n_region,
      _lib.ptr(sums), _lib.ptr(metrics), stream), 'wb2_det_combine')
  return metrics, sums


def _suite_field(plan, dtype, mode):
  """(weight field, its dtype code) of a deterministic launch: float32 where
  the field's values are float32 numbers and the launch has the instantiation
  (same bits, half the field bytes per point)."""
  field, field_code = plan.wfield, _lib.WB2_F64
  if (dtype == torch.float32 and mode in _FIELD_F32_MODES and
      getattr(plan, 'wfield32', None) is not None and
      os.environ.get('WB2HIP_FIELD_F32', '1') != '0'):
    field, field_code = plan.wfield32, _lib.WB2_F32
  return field, field_code


def pairs_supported(plan: ReductionPlan, mode: int, dtype: torch.dtype,
                    skipna: bool, aligned: bool = True) -> bool:
  """Whether a launch of `mode` over `plan` can answer wind-vector pairs from
  its own read (wb2_pairs_supported; WB2HIP_WIND_PAIRS=0 switches the pair
  kernel off for A/B runs: the pairs then take a WB2_MODE_WIND launch)."""
  if os.environ.get('WB2HIP_WIND_PAIRS', '1') == '0' or dtype not in _DTYPES:
    return False
  field, _ = _suite_field(plan, dtype, mode)
  aligned = aligned and (field is None or field.data_ptr() % 16 == 0)
  return bool(_lib.load().wb2_pairs_supported(
      mode, _DTYPES[dtype], int(skipna), int(plan.wfield is not None),
      plan.n_col, int(aligned)))


def stream_reduce_pairs(plan: ReductionPlan, mode: int, dtype: torch.dtype,
                        addr: t.Sequence[torch.Tensor], aligned16: bool,
                        n_outer: int, n_pair: int, skipna: bool):
  """K1 + K1p + K2 over slabs given by address, the last 2 * n_pair of them the
  u slabs then the v slabs of n_pair wind-vector pairs: returns
  (metrics[NMETRIC, n_region, n_outer], wind[NMETRIC, n_region, n_pair]) --
  the bits of stream_reduce_addr over the slabs plus a MODE_WIND pass over the
  pairs, from one read (wb2_det_wind_suite_step)."""
  for a in addr:
    if (a.dtype != torch.int64 or a.device != plan.device or
        a.numel() != n_outer or not a.is_contiguous()):
      raise ValueError('address tables are contiguous int64[n_outer] on the '
                       'plan device')
  step = PairSuiteStep(plan, mode, dtype, skipna, n_outer, n_pair,
                       aligned=bool(aligned16))
  return step.run(None, list(addr))


class PairSuiteStep:
  """The deterministic suite over a launch whose last 2 * n_pair slabs are the
  u slabs, then the v slabs, of n_pair wind-vector pairs: per-variable metrics
  of all n_outer slabs AND the wind-vector metrics of the pairs from ONE read
  (wb2_det_wind_suite_step; metrics.py:283-301 calling :194-201 derive both
  from the same `diff`).  Bit-identical to a DET / DET_ACC launch over the
  slabs plus a WIND launch over the pairs.

  Prepared once per (plan, mode, dtype, skipna, n_outer, n_pair); `run` takes
  inputs + slab-number tables, or inputs=None + address tables, and returns
  (metrics[NMETRIC, n_region, n_outer], wind[NMETRIC, n_region, n_pair])."""

  def __init__(self, plan: ReductionPlan, mode: int, dtype: torch.dtype,
               skipna: bool, n_outer: int, n_pair: int, aligned: bool = True):
    import ctypes
    lib = _lib.load()
    if dtype not in _DTYPES:
      raise TypeError(f'unsupported dtype {dtype}')
    if mode not in (_lib.MODE_DET, _lib.MODE_DET_ACC):
      raise ValueError('wind-vector pairs ride on MODE_DET / MODE_DET_ACC')
    if not 0 <= 2 * n_pair <= n_outer:
      raise ValueError(f'{n_pair=} does not fit {n_outer=}')
    self.lib, self.plan, self.mode, self.skipna = lib, plan, mode, bool(skipna)
    self.code, self.n_outer, self.n_pair = _DTYPES[dtype], int(n_outer), int(
        n_pair)
    self.aligned = bool(aligned)
    dev = plan.device
    field, field_code = _suite_field(plan, dtype, mode)
    aligned = aligned and (field is None or field.data_ptr() % 16 == 0)
    tile = lib.wb2_tile_cols_ex(mode, self.code, int(skipna),
                                int(plan.wfield is not None), plan.n_col,
                                int(aligned))
    self.tables, self._keep = plan_tables(plan, tile, field, field_code)
    n_ts = self._keep[3]
    k = lib.wb2_num_slots(mode, int(skipna))
    kw = lib.wb2_num_slots(_lib.MODE_WIND, int(skipna))
    self.partials = torch.empty((n_outer, plan.n_chunk, plan.nwf, n_ts, k),
                                dtype=torch.float64, device=dev)
    self.wind_partials = torch.empty(
        (max(n_pair, 1), plan.n_chunk, plan.nwf, n_ts, kw),
        dtype=torch.float64, device=dev)
    self._tables_ref = ctypes.byref(self.tables)
    self._fn = lib.wb2_det_wind_suite_step
    self.n_metric = _lib.NMETRIC
    self.n_values = _lib.NMETRIC * plan.n_region * (self.n_outer + self.n_pair)

  def run(self, inputs, tables, out: t.Optional[torch.Tensor] = None,
          stream_ptr: t.Optional[int] = None):
    """`out` (optional): flat float64[n_values] that receives the per-variable
    block followed by the wind block."""
    dev = self.plan.device
    if out is None:
      out = torch.empty((self.n_values,), dtype=torch.float64, device=dev)
    n_det = _lib.NMETRIC * self.plan.n_region * self.n_outer
    hook = _LAUNCH_HOOK
    if hook is not None:
      hook('begin', 'stream_partials')
    status = self._fn(
        self._tables_ref, self.mode, self.code, int(self.skipna),
        None if inputs is None else _lib.ptr_array(inputs),
        _lib.ptr_array(tables), int(self.aligned), self.n_outer, self.n_pair,
        self.partials.data_ptr(), self.wind_partials.data_ptr(),
        out.data_ptr(), out.data_ptr() + 8 * n_det,
        current_stream_ptr(dev) if stream_ptr is None else stream_ptr)
    if hook is not None:
      hook('end', 'stream_partials')
    if status != 0:
      _lib.check(status, 'wb2_det_wind_suite_step')
    flat = out.view(-1)
    return (flat[:n_det].view(_lib.NMETRIC, self.plan.n_region, self.n_outer),
            flat[n_det:n_det + _lib.NMETRIC * self.plan.n_region * self.n_pair]
            .view(_lib.NMETRIC, self.plan.n_region, self.n_pair))


class SuiteStep:
  """One chunk of the deterministic suite per call: K1 -> K2 -> the running
  temporal mean through wb2_det_suite_step -- ONE C-ABI call per chunk instead
  of three calls with ~25 marshalled arguments each (the kernels and their
  bits are those of stream_reduce + time_accumulate).

  Prepared once per (plan, mode, dtype, skipna, n_outer): the plan's tables go
  into a `wb2_plan_tables` struct, the partials / metrics scratch is allocated
  here and reused by every call (calls are stream-ordered).  `run` takes the
  chunk's inputs and, optionally, the accumulators.

    step = SuiteStep(plan, MODE_DET_ACC, torch.float32, False, n_outer)
    step.accumulate_into(total, count, (n_lead, n_time, n_tail))
    step.run([f, t, c], [f_tab, t_tab, c_tab])     # per chunk
  """

  def __init__(self, plan: ReductionPlan, mode: int, dtype: torch.dtype,
               skipna: bool, n_outer: int, aligned: bool = True,
               aux: t.Optional[torch.Tensor] = None, scalar: float = 0.0,
               by_address: bool = False):
    import ctypes
    lib = _lib.load()
    if dtype not in _DTYPES:
      raise TypeError(f'unsupported dtype {dtype}')
    self.lib, self.plan, self.mode, self.skipna = lib, plan, mode, bool(skipna)
    self.code, self.n_outer = _DTYPES[dtype], int(n_outer)
    self.by_address, self.aligned = bool(by_address), bool(aligned)
    dev = plan.device
    k = lib.wb2_num_slots(mode, int(skipna))
    field, field_code = plan.wfield, _lib.WB2_F64
    if (dtype == torch.float32 and mode in _FIELD_F32_MODES and
        getattr(plan, 'wfield32', None) is not None and
        os.environ.get('WB2HIP_FIELD_F32', '1') != '0'):
      field, field_code = plan.wfield32, _lib.WB2_F32
    aligned = aligned and (field is None or field.data_ptr() % 16 == 0)
    tile = lib.wb2_tile_cols_ex(mode, self.code, int(skipna),
                                int(plan.wfield is not None), plan.n_col,
                                int(aligned))
    self.tables, self._keep = plan_tables(plan, tile, field, field_code, aux,
                                          scalar)
    n_ts = self._keep[3]
    self.n_metric = _lib.GENERIC_KQ.get(mode, _lib.NMETRIC)
    self.partials = torch.empty((n_outer, plan.n_chunk, plan.nwf, n_ts, k),
                                dtype=torch.float64, device=dev)
    self.metrics = torch.empty((self.n_metric, plan.n_region, n_outer),
                               dtype=torch.float64, device=dev)
    self._tables_ref = ctypes.byref(self.tables)
    self._acc = (0, 0, 0, 0, None, None, None)
    self._acc_keep = None
    self._fn = lib.wb2_det_suite_step

  def accumulate_into(self, total: t.Optional[torch.Tensor],
                      count: t.Optional[torch.Tensor] = None,
                      view: t.Optional[tuple] = None, skipna: bool = False,
                      dst: t.Optional[torch.Tensor] = None) -> None:
    """Every later run() adds its metrics, read as [lead][time][tail] = `view`,
    to total / count over `time` (None: no accumulation)."""
    if total is None:
      self._acc, self._acc_keep = (0, 0, 0, 0, None, None, None), None
      return
    n_lead, n_time, n_tail = (int(v) for v in view)
    if n_lead * n_time * n_tail != self.metrics.numel():
      raise ValueError(f'view {view} does not cover the metrics '
                       f'{tuple(self.metrics.shape)}')
    for x in (total, count):
      if x.dtype != torch.float64 or not x.is_contiguous():
        raise ValueError('accumulators are contiguous float64 tensors')
    if dst is None and total.numel() != n_lead * n_tail:
      raise ValueError('accumulator shape mismatch')
    if dst is not None and (dst.dtype != torch.int64 or
                            dst.numel() != n_lead * n_tail):
      raise ValueError('dst is int64 with one entry per result element')
    self._acc = (n_lead, n_time, n_tail, int(skipna), _lib.ptr(dst) or None,
                 total.data_ptr(), count.data_ptr())
    self._acc_keep = (total, count, dst)

  def run(self, inputs: t.Optional[t.Sequence[torch.Tensor]],
          tables: t.Sequence[t.Optional[torch.Tensor]],
          metrics: t.Optional[torch.Tensor] = None,
          stream_ptr: t.Optional[int] = None) -> torch.Tensor:
    """Enqueues the step on the current stream.  `inputs` + slab-number
    `tables` (None entries = identity), or inputs=None + address tables
    (by_address).  Returns the metrics tensor [n_metric, n_region, n_outer]
    (the step's own scratch unless `metrics` is given: valid until the next
    run)."""
    out = self.metrics if metrics is None else metrics
    if self.by_address != (inputs is None):
      raise ValueError('by_address steps take inputs=None and address tables')
    hook = _LAUNCH_HOOK
    if hook is not None:  # (brackets K1 + K2 [+ the accumulation] here)
      hook('begin', 'stream_partials')
    status = self._fn(
        self._tables_ref, self.mode, self.code, int(self.skipna),
        None if inputs is None else _lib.ptr_array(inputs),
        _lib.ptr_array(tables), int(self.aligned), self.n_outer,
        self.partials.data_ptr(), out.data_ptr(), *self._acc,
        current_stream_ptr(self.plan.device) if stream_ptr is None
        else stream_ptr)
    if hook is not None:
      hook('end', 'stream_partials')
    if status != 0:
      _lib.check(status, 'wb2_det_suite_step')
    return out

  def bind(self, inputs, tables, stream_ptr: t.Optional[int] = None):
    """The call for one fixed (inputs, tables) pair with its arguments
    marshalled ONCE (the pointer arrays are built here, not per call): a
    zero-argument callable for loops that come back to the same chunk buffers
    (a ring of staging slots, a benchmark's table sets).  The accumulators are
    the ones set when bind() is called; the stream is `stream_ptr` or the
    current stream at bind time."""
    if self.by_address != (inputs is None):
      raise ValueError('by_address steps take inputs=None and address tables')
    fn = self._fn
    keep = (inputs, tables, self._acc_keep)
    args = (self._tables_ref, self.mode, self.code, int(self.skipna),
            None if inputs is None else _lib.ptr_array(inputs),
            _lib.ptr_array(tables), int(self.aligned), self.n_outer,
            self.partials.data_ptr(), self.metrics.data_ptr()) + self._acc + (
                current_stream_ptr(self.plan.device) if stream_ptr is None
                else stream_ptr,)

    def call(_keep=keep):
      if fn(*args) != 0:
        _lib.check(-1, 'wb2_det_suite_step')
    return call


GATHER_MAX_MEMBERS = {torch.float32: 128, torch.float64: 64}  # register sort
_NAN_SLABS: dict = {}
_NAN_SLABS_LOCK = threading.Lock()


def nan_slab(device: torch.device, dtype: torch.dtype, n_elems: int):
  """One resident slab of NaNs per (device, dtype, size): what the holes of a
  gathered ensemble point at."""
  key = (str(device), dtype, int(n_elems))
  slab = _NAN_SLABS.get(key)
  if slab is None:
    with _NAN_SLABS_LOCK:
      slab = _NAN_SLABS.get(key)
      if slab is None:
        slab = torch.full((int(n_elems),), float('nan'), dtype=dtype,
                          device=device)
        # filled before any other thread's stream can read it (once per size)
        if torch.device(device).type == 'cuda':
          torch.cuda.current_stream(device).synchronize()
        _NAN_SLABS[key] = slab
  return slab


def gather_pointers(base: torch.Tensor, index: np.ndarray, slab_elems: int):
  """Device addresses of the slabs `index` (any shape, -1 = hole) of the
  contiguous device tensor `base` ([..., slab]): int64 array of index.shape
  (host).  Holes point at the resident NaN slab."""
  index = np.asarray(index, dtype=np.int64)
  ptrs = base.data_ptr() + index * (slab_elems * base.element_size())
  if (index < 0).any():
    hole = nan_slab(base.device, base.dtype, slab_elems).data_ptr()
    ptrs = np.where(index < 0, np.int64(hole), ptrs)
  return ptrs


def ensemble_reduce(plan: ReductionPlan, ens: torch.Tensor,
                    member_stride: int, n_member: int,
                    ens_slab: t.Optional[torch.Tensor], truth: torch.Tensor,
                    truth_slab: t.Optional[torch.Tensor], n_outer: int,
                    skipna: bool, want_sums: bool = False,
                    maps: t.Optional[torch.Tensor] = None,
                    member_ptrs: t.Optional[torch.Tensor] = None,
                    addresses: t.Optional[torch.Tensor] = None):
  """Runs K3 + the region fold.  `ens` holds the members member-major with
  `member_stride` elements between members; `truth` is [n_slab, n_row, n_col].
  With `member_ptrs` (int64[n_outer, n_member] device ADDRESSES of the member
  slabs, `gather_pointers`) the ensemble is read in place from wherever its
  slabs live: `ens` is then only the tensor those addresses point into (kept
  alive, dtype), `member_stride` and `ens_slab` are ignored.
  With `addresses` (int64[2, n_outer] on the device: the byte address of member
  0's slab and of the truth slab of every outer index; member m follows
  `member_stride` elements behind member 0 -- wb2_ens_partials_addr) `ens` and
  `truth` are tensors of the right dtype that stay alive, nothing more: the
  chunks of a window are read where they lie (xarray_lite.SlabConcat).

  Returns (metrics[NMETRIC_ENS, n_region, n_outer], sums or None).
  """
  lib = _lib.load()
  dev = plan.device
  dtype = ens.dtype
  if dtype not in _DTYPES or truth.dtype != dtype:
    raise TypeError(f'unsupported / mismatched dtypes {ens.dtype} {truth.dtype}')
  # (a non-contiguous `ens` is a view with intact slabs that the caller
  # addresses through member_stride + ens_slab: metrics._ens_layout)
  if addresses is not None:
    if (addresses.dtype != torch.int64 or addresses.device != dev or
        not addresses.is_contiguous() or addresses.numel() != 2 * n_outer or
        maps is not None or member_ptrs is not None):
      raise ValueError('addresses is a contiguous int64[2, n_outer] on the '
                       'plan device (no maps, no member_ptrs)')
  elif ens.device != dev or truth.device != dev or not truth.is_contiguous() or (
      not ens.is_contiguous() and ens_slab is None and member_ptrs is None):
    raise ValueError('inputs must be contiguous on the plan device')
  for s in (ens_slab, truth_slab):
    if s is not None and (s.dtype != torch.int64 or s.numel() != n_outer):
      raise ValueError('slab tables are int64[n_outer]')
  if member_ptrs is not None and (
      member_ptrs.dtype != torch.int64 or member_ptrs.device != dev or
      not member_ptrs.is_contiguous() or
      member_ptrs.numel() != n_outer * n_member):
    raise ValueError('member_ptrs is a contiguous int64[n_outer, n_member] '
                     'on the plan device')
  k = lib.wb2_ens_num_slots(int(skipna))
  tile = lib.wb2_ens_tile_cols(plan.n_col)
  n_ctile = -(-plan.n_col // tile)
  seg_eoff, n_ts = plan.seg_entries(tile)
  stream = current_stream_ptr(dev)
  partials = torch.empty((n_outer, plan.n_chunk, plan.nwf, n_ts, k),
                         dtype=torch.float64, device=dev)
  if _LAUNCH_HOOK is not None:
    _LAUNCH_HOOK('begin', 'ens_partials')
  if maps is not None and (maps.dtype != torch.float64 or maps.numel() !=
                           6 * n_outer * plan.n_row * plan.n_col):
    raise ValueError('maps must be float64[6, n_outer, n_row * n_col]')
  if addresses is not None:
    base = addresses.data_ptr()
    _lib.check(lib.wb2_ens_partials_addr(
        _DTYPES[dtype], int(skipna), base, base + 8 * n_outer, n_member,
        member_stride, n_outer, plan.n_row, plan.n_col, _lib.ptr(plan.w_row),
        _lib.ptr(plan.w_col), _lib.ptr(plan.wfield), _lib.ptr(plan.chunk_row0),
        _lib.ptr(plan.chunk_nrow), plan.n_chunk, n_ctile,
        _lib.ptr(plan.seg_col0), _lib.ptr(seg_eoff), plan.n_seg, n_ts,
        _lib.ptr(partials), stream), 'wb2_ens_partials_addr')
  elif member_ptrs is not None:
    _lib.check(lib.wb2_ens_partials_gather(
        _DTYPES[dtype], int(skipna), _lib.ptr(member_ptrs), _lib.ptr(truth),
        _lib.ptr(truth_slab), n_member, n_outer, plan.n_row, plan.n_col,
        _lib.ptr(plan.w_row), _lib.ptr(plan.w_col), _lib.ptr(plan.wfield),
        _lib.ptr(plan.chunk_row0), _lib.ptr(plan.chunk_nrow), plan.n_chunk,
        n_ctile, _lib.ptr(plan.seg_col0), _lib.ptr(seg_eoff), plan.n_seg, n_ts,
        _lib.ptr(partials), _lib.ptr(maps), stream), 'wb2_ens_partials_gather')
  else:
    _lib.check(lib.wb2_ens_partials_maps(
        _DTYPES[dtype], int(skipna), _lib.ptr(ens), _lib.ptr(ens_slab),
        _lib.ptr(truth), _lib.ptr(truth_slab), n_member, member_stride,
        n_outer, plan.n_row, plan.n_col, _lib.ptr(plan.w_row),
        _lib.ptr(plan.w_col), _lib.ptr(plan.wfield), _lib.ptr(plan.chunk_row0),
        _lib.ptr(plan.chunk_nrow), plan.n_chunk, n_ctile,
        _lib.ptr(plan.seg_col0), _lib.ptr(seg_eoff), plan.n_seg, n_ts,
        _lib.ptr(partials), _lib.ptr(maps), stream), 'wb2_ens_partials_maps')
  if _LAUNCH_HOOK is not None:
    _LAUNCH_HOOK('end', 'ens_partials')
  metrics = torch.empty((_lib.NMETRIC_ENS, plan.n_region, n_outer),
                        dtype=torch.float64, device=dev)
  sums = (torch.empty((n_outer, plan.n_region, k), dtype=torch.float64,
                      device=dev) if want_sums else None)
  _lib.check(lib.wb2_ens_combine(
      int(skipna), _lib.ptr(partials), n_outer, plan.n_chunk, plan.nwf,
      plan.n_seg, _lib.ptr(seg_eoff), n_ts, _lib.ptr(plan.band_chunk0),
      plan.n_band, _lib.ptr(plan.coef_band), _lib.ptr(plan.coef_seg),
      _lib.ptr(plan.region_wf), _lib.ptr(plan.region_wsum), plan.n_region,
      _lib.ptr(sums), _lib.ptr(metrics), stream), 'wb2_ens_combine')
  return metrics, sums


def plan_tables(plan: ReductionPlan, tile_cols: int, field=None,
                field_code: int = _lib.WB2_F64, aux=None, scalar: float = 0.0):
  """(`wb2_plan_tables` struct of `plan` for a column-tile width, what it
  points at -- keep both alive until the call has been enqueued)."""
  seg_eoff, n_ts = plan.seg_entries(tile_cols)
  tables = _lib.PlanTables(
      n_row=plan.n_row, n_col=plan.n_col, n_chunk=plan.n_chunk,
      n_ctile=-(-plan.n_col // tile_cols), n_seg=plan.n_seg, n_ts=n_ts,
      n_band=plan.n_band, n_region=plan.n_region,
      w_row=_lib.ptr(plan.w_row) or None, w_col=_lib.ptr(plan.w_col) or None,
      wfield=_lib.ptr(field) or None, wfield_dtype=field_code, reserved=0,
      aux=_lib.ptr(aux) or None, scalar=float(scalar),
      chunk_row0=_lib.ptr(plan.chunk_row0), chunk_nrow=_lib.ptr(plan.chunk_nrow),
      seg_col0=_lib.ptr(plan.seg_col0), seg_eoff=_lib.ptr(seg_eoff),
      band_chunk0=_lib.ptr(plan.band_chunk0),
      coef_band=_lib.ptr(plan.coef_band), coef_seg=_lib.ptr(plan.coef_seg),
      region_wf=_lib.ptr(plan.region_wf),
      region_wsum=_lib.ptr(plan.region_wsum))
  return tables, (field, aux, seg_eoff, n_ts)


def energy_score(plan: ReductionPlan, ens: torch.Tensor, member_stride: int,
                 n_member: int, ens_slab: t.Optional[torch.Tensor],
                 truth: torch.Tensor, truth_slab: t.Optional[torch.Tensor],
                 n_outer: int, skipna: bool) -> torch.Tensor:
  """wb2_energy_score: (score, spread, skill)[3, n_region, n_outer] float64 of
  a member-major ensemble (`member_stride` elements between members) in one
  read of the members.  Asynchronous on the current stream."""
  import ctypes
  lib = _lib.load()
  dev = plan.device
  dtype = ens.dtype
  if dtype not in _DTYPES or truth.dtype != dtype:
    raise TypeError(f'unsupported / mismatched dtypes {ens.dtype} {truth.dtype}')
  # (a non-contiguous `ens` is a view with intact slabs that the caller
  # addresses through member_stride + ens_slab: metrics._ens_layout)
  if ens.device != dev or truth.device != dev or not truth.is_contiguous() or (
      not ens.is_contiguous() and ens_slab is None):
    raise ValueError('inputs must be contiguous on the plan device')
  for s in (ens_slab, truth_slab):
    if s is not None and (s.dtype != torch.int64 or s.numel() != n_outer):
      raise ValueError('slab tables are int64[n_outer]')
  block, n_block, k = (ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32())
  _lib.check(lib.wb2_energy_layout(
      n_member, int(skipna), int(plan.wfield is not None), ctypes.byref(block),
      ctypes.byref(n_block), ctypes.byref(k)), 'wb2_energy_layout')
  tables, keep = plan_tables(plan, lib.wb2_ens_tile_cols(plan.n_col),
                             plan.wfield)
  n_ts = keep[3]
  n_virtual = n_outer * n_block.value
  partials = torch.empty((n_virtual, plan.n_chunk, plan.nwf, n_ts, k.value),
                         dtype=torch.float64, device=dev)
  means = torch.empty((2 * block.value, plan.n_region, n_virtual),
                      dtype=torch.float64, device=dev)
  out = torch.empty((3, plan.n_region, n_outer), dtype=torch.float64,
                    device=dev)
  if _LAUNCH_HOOK is not None:
    _LAUNCH_HOOK('begin', 'energy_score')
  _lib.check(lib.wb2_energy_score(
      _DTYPES[dtype], int(skipna), _lib.ptr(ens), _lib.ptr(ens_slab),
      _lib.ptr(truth), _lib.ptr(truth_slab), n_member, member_stride, n_outer,
      ctypes.byref(tables), _lib.ptr(partials), _lib.ptr(means), _lib.ptr(out),
      current_stream_ptr(dev)), 'wb2_energy_score')
  if _LAUNCH_HOOK is not None:
    _LAUNCH_HOOK('end', 'energy_score')
  del keep
  return out


def time_accumulate(values: torch.Tensor, time_axis: int, skipna: bool,
                    total: torch.Tensor, count: torch.Tensor,
                    dst: t.Optional[torch.Tensor] = None, run: int = 1):
  """total/count += sum/notnull-count of `values` (float32 or float64) over
  `time_axis` (device); the sums continue from the accumulators value by value.  `dst` (int64 device
  tensor, one entry per RUN of `run` consecutive elements of `values` without
  its time axis): where in `total` / `count` the run goes (identity without
  it)."""
  lib = _lib.load()
  if values.dtype not in _DTYPES:
    values = values.to(torch.float64)
  values = values.contiguous()
  shape = tuple(values.shape)
  n_lead = int(np.prod(shape[:time_axis], dtype=np.int64))
  n_time = shape[time_axis]
  n_tail = int(np.prod(shape[time_axis + 1:], dtype=np.int64))
  if count is None and skipna:
    raise ValueError('skipna needs the count accumulator')
  if dst is None:
    if total.numel() != n_lead * n_tail or (
        count is not None and count.numel() != n_lead * n_tail):
      raise ValueError('accumulator shape mismatch')
  elif (dst.dtype != torch.int64 or run < 1 or (n_lead * n_tail) % run
        or dst.numel() != n_lead * n_tail // run
        or (count is not None and total.numel() != count.numel())):
    raise ValueError('dst is int64 with one entry per run of result elements')
  # (the entries of `dst` must be distinct and < total.numel(): the kernel
  # reads, adds and writes each destination without atomics.  RunningMean
  # builds them on the host -- _Accumulator.destinations -- and checks there.)
  _lib.check(lib.wb2_time_accumulate_runs(
      _DTYPES[values.dtype], _lib.ptr(values), n_lead, n_time, n_tail,
      int(skipna), _lib.ptr(dst), int(run),
      _lib.ptr(total), _lib.ptr(count), current_stream_ptr(values.device)),
             'wb2_time_accumulate')


class _SpectrumPlans(threading.local):
  """hipFFT plans are expensive (rocFFT compiles kernels): cache by shape.  One
  cache per thread: a plan's hipFFT handle is bound to a stream while it runs,
  and a thread only ever evicts plans of its own."""

  def __init__(self):
    self.plans = {}

  def get(self, dtype_code: int, n_lon: int, n_rows: int):
    import ctypes
    key = (dtype_code, n_lon, n_rows, torch.cuda.current_device())
    hit = self.plans.get(key)
    if hit is None:
      lib = _lib.load()
      handle = ctypes.c_void_p()
      _lib.check(lib.wb2_spectrum_plan_create(dtype_code, n_lon, n_rows,
                                              ctypes.byref(handle)),
                 'wb2_spectrum_plan_create')
      nbytes = lib.wb2_spectrum_plan_workspace(handle)
      if nbytes < 0:
        _lib.check(-1, 'wb2_spectrum_plan_workspace')
      if len(self.plans) >= 8:  # bounded: drop the oldest plan
        old_key = next(iter(self.plans))
        lib.wb2_spectrum_plan_destroy(self.plans.pop(old_key)[0])
      hit = (handle, int(nbytes))
      self.plans[key] = hit
    return hit


_SPECTRUM_PLANS = _SpectrumPlans()


def zonal_spectrum(x: torch.Tensor, circumference: torch.Tensor, n_lat: int,
                   n_time: int = 0, skipna: bool = True) -> torch.Tensor:
  """x: [..., n_lat, n_lon] contiguous device tensor (rows = everything but lon).

  Returns float64 [..., n_lat, n_lon//2+1], or with n_time > 0 (x's LEADING
  dim is time) the time mean [rest..., n_lat, n_lon//2+1].
  """
  lib = _lib.load()
  if x.dtype not in _DTYPES or not x.is_contiguous():
    raise ValueError('x must be a contiguous float32/float64 tensor')
  n_lon = x.shape[-1]
  n_rows = x.numel() // n_lon
  n_bins = n_lon // 2 + 1
  if n_time > 0:
    if x.shape[0] != n_time:
      raise ValueError('time must be the leading dim for the fused time mean')
    out_shape = tuple(x.shape[1:-1]) + (n_bins,)
  else:
    out_shape = tuple(x.shape[:-1]) + (n_bins,)
  out = torch.empty(out_shape, dtype=torch.float64, device=x.device)
  if n_rows == 0:  # an empty chunk: nothing to transform (a plan needs rows)
    return out
  handle, nbytes = _SPECTRUM_PLANS.get(_DTYPES[x.dtype], n_lon, n_rows)
  work = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=x.device)
  _lib.check(lib.wb2_zonal_spectrum(
      handle, _lib.ptr(x), _lib.ptr(circumference), n_lat, n_time, int(skipna),
      _lib.ptr(out), _lib.ptr(work), current_stream_ptr(x.device)),
             'wb2_zonal_spectrum')
  return out


def zonal_spectrum_lat_mean(x: torch.Tensor, circumference: torch.Tensor,
                            lat_weights: torch.Tensor, n_lat: int,
                            weight_sum: t.Optional[float] = None,
                            row_weight: t.Optional[torch.Tensor] = None
                            ) -> torch.Tensor:
  """Area-weighted latitude mean of the zonal energy spectrum (BASELINE
  configs[3]) of x: [..., n_lat, n_lon] -> float64 [..., n_lon//2+1]
  = sum_lat w[lat] S[..., lat, :] / sum_lat w[lat].

  One fused kernel when the plan has the LDS FFT (float32 rows of an
  instantiated length): the per-latitude spectra are never written.  Otherwise
  the spectrum is materialised and reduced with the axis kernel.  `weight_sum`
  (= float(sum(w)), known on the host) and `row_weight` (= w * circumference,
  float64 on the device) may be passed by callers that evaluate many chunks on
  one grid: the call is then exactly two kernel launches."""
  lib = _lib.load()
  if x.dtype not in _DTYPES or not x.is_contiguous():
    raise ValueError('x must be a contiguous float32/float64 tensor')
  n_lon = x.shape[-1]
  if x.shape[-2] != n_lat:
    raise ValueError('latitude must be the second-to-last dim')
  n_rows = x.numel() // n_lon
  n_field = n_rows // n_lat
  n_bins = n_lon // 2 + 1
  if n_rows == 0:
    return torch.empty(tuple(x.shape[:-2]) + (n_bins,), dtype=torch.float64,
                       device=x.device)
  handle, nbytes = _SPECTRUM_PLANS.get(_DTYPES[x.dtype], n_lon, n_rows)
  w = lat_weights.to(torch.float64)
  n_seg = lib.wb2_zonal_spectrum_latmean_segments(handle, n_lat)
  if n_seg < 0:
    _lib.check(n_seg, 'wb2_zonal_spectrum_latmean_segments')
  out_shape = tuple(x.shape[:-2]) + (n_bins,)
  if n_seg == 0 or x.data_ptr() % 16:
    spec = zonal_spectrum(x, circumference, n_lat)
    total, _, count = axis_moments(spec.reshape(n_field, n_lat, n_bins), n_field,
                                   n_lat, n_bins, w, False)
    return (total / count).reshape(out_shape)
  if row_weight is None:
    row_weight = (w * circumference.to(torch.float64)).contiguous()
  # without a host-side sum the normalisation stays on the device (no sync):
  # scale = 1 in the kernel, one division afterwards
  scale = 1.0 if weight_sum is None else 1.0 / float(weight_sum)
  partial = torch.empty((n_field, n_seg, n_bins), dtype=torch.float64,
                        device=x.device)
  out = torch.empty((n_field, n_bins), dtype=torch.float64, device=x.device)
  _lib.check(lib.wb2_zonal_spectrum_latmean(
      handle, _lib.ptr(x), _lib.ptr(row_weight), n_lat, n_seg, scale,
      _lib.ptr(partial), _lib.ptr(out), current_stream_ptr(x.device)),
             'wb2_zonal_spectrum_latmean')
  if weight_sum is None:
    out = out / w.sum()
  return out.reshape(out_shape)


def spatial_maps(forecast: torch.Tensor, f_slab, truth: torch.Tensor, t_slab,
                 n_outer: int, n_point: int, want=('bias', 'mse', 'mae')):
  """K5 per-time maps: returns {name: tensor[n_outer, n_point]} (input dtype)."""
  lib = _lib.load()
  dev = forecast.device
  if forecast.dtype not in _DTYPES or truth.dtype != forecast.dtype:
    raise TypeError('forecast/truth must share a float32/float64 dtype')
  outs = {k: torch.empty((n_outer, n_point), dtype=forecast.dtype, device=dev)
          for k in want}
  _lib.check(lib.wb2_spatial_maps(
      _DTYPES[forecast.dtype], _lib.ptr(forecast), _lib.ptr(f_slab),
      _lib.ptr(truth), _lib.ptr(t_slab), n_outer, n_point,
      _lib.ptr(outs.get('bias')), _lib.ptr(outs.get('mse')),
      _lib.ptr(outs.get('mae')), current_stream_ptr(dev)), 'wb2_spatial_maps')
  return outs


def spatial_accumulate(forecast: torch.Tensor, f_slab, truth: torch.Tensor,
                       t_slab, n_time: int, n_rest: int, n_point: int,
                       skipna: bool, total: torch.Tensor,
                       count: t.Optional[torch.Tensor]):
  """K5 temporal accumulation into total/count [3, n_rest, n_point] (fp64)."""
  lib = _lib.load()
  if forecast.dtype not in _DTYPES or truth.dtype != forecast.dtype:
    raise TypeError('forecast/truth must share a float32/float64 dtype')
  _lib.check(lib.wb2_spatial_accumulate(
      _DTYPES[forecast.dtype], int(skipna), _lib.ptr(forecast),
      _lib.ptr(f_slab), _lib.ptr(truth), _lib.ptr(t_slab), n_time, n_rest,
      n_point, _lib.ptr(total), _lib.ptr(count),
      current_stream_ptr(forecast.device)), 'wb2_spatial_accumulate')


def ensemble_threshold_reduce(plan: ReductionPlan, ens: torch.Tensor,
                              member_stride: int, n_member: int, ens_slab,
                              truth: torch.Tensor, truth_slab,
                              thr: torch.Tensor, thr_slab, n_outer: int,
                              skipna: bool):
  """Exceedance-count kernel + region fold: metrics[4, n_region, n_outer] =
  (Brier, debiased Brier, ignorance, RPS part)."""
  lib = _lib.load()
  dev = plan.device
  dtype = ens.dtype
  if dtype not in _DTYPES or truth.dtype != dtype or thr.dtype != dtype:
    raise TypeError('members, truth and threshold must share one dtype')
  for x in (ens, truth, thr):
    if x.device != dev or not x.is_contiguous():
      raise ValueError('inputs must be contiguous on the plan device')
  mode = _lib.MODE_ENS_THR
  k = lib.wb2_num_slots(mode, int(skipna))
  tile = lib.wb2_ens_tile_cols(plan.n_col)
  n_ctile = -(-plan.n_col // tile)
  seg_eoff, n_ts = plan.seg_entries(tile)
  stream = current_stream_ptr(dev)
  partials = torch.empty((n_outer, plan.n_chunk, plan.nwf, n_ts, k),
                         dtype=torch.float64, device=dev)
  _lib.check(lib.wb2_ens_threshold_partials(
      _DTYPES[dtype], int(skipna), _lib.ptr(ens), _lib.ptr(ens_slab),
      _lib.ptr(truth), _lib.ptr(truth_slab), _lib.ptr(thr), _lib.ptr(thr_slab),
      n_member, member_stride, n_outer, plan.n_row, plan.n_col,
      _lib.ptr(plan.w_row), _lib.ptr(plan.w_col), _lib.ptr(plan.wfield),
      _lib.ptr(plan.chunk_row0), _lib.ptr(plan.chunk_nrow), plan.n_chunk,
      n_ctile, _lib.ptr(plan.seg_col0), _lib.ptr(seg_eoff), plan.n_seg, n_ts,
      _lib.ptr(partials), stream), 'wb2_ens_threshold_partials')
  metrics = torch.empty((_lib.GENERIC_KQ[mode], plan.n_region, n_outer),
                        dtype=torch.float64, device=dev)
  _lib.check(lib.wb2_det_combine(
      mode, int(skipna), _lib.ptr(partials), n_outer, plan.n_chunk, plan.nwf,
      plan.n_seg, _lib.ptr(seg_eoff), n_ts, _lib.ptr(plan.band_chunk0),
      plan.n_band, _lib.ptr(plan.coef_band), _lib.ptr(plan.coef_seg),
      _lib.ptr(plan.region_wf), _lib.ptr(plan.region_wsum), plan.n_region,
      None, _lib.ptr(metrics), stream), 'wb2_det_combine')
  return metrics


RANK_MEAN_MAX_BINS = 256  # wb2_rank_histogram_mean: 64 x n_bins counts in LDS


def rank_histogram(ens: torch.Tensor, member_stride: int, n_member: int,
                   ens_slab, truth: torch.Tensor, truth_slab, n_outer: int,
                   n_point: int, n_bins: int, break_ties: bool, seed: int,
                   acc_row=None, n_acc: int = 0, numpy_stream=None,
                   mean_over=None) -> torch.Tensor:
  """wb2_rank_histogram: one-hot [n_outer, n_point, n_bins] (float64), or with
  `acc_row` the per-row counts [n_acc, n_point, n_bins], or with `mean_over` =
  (n_lead, n_time, n_tail), n_outer = their product: the MEAN of the one-hots
  over the middle axis [n_lead * n_tail, n_point, n_bins]
  (wb2_rank_histogram_mean: no one-hots, no atomics).

  `numpy_stream` = (pcg_state, pcg_inc, ref_outer_off[n_outer] int64 device
  tensor, (row, col, member) strides, n_col) switches to
  wb2_rank_histogram_seeded: ties are broken with the very perturbations
  np.random.default_rng(seed).uniform hands the reference."""
  lib = _lib.load()
  dev = ens.device
  if ens.dtype not in _DTYPES or truth.dtype != ens.dtype:
    raise TypeError('members and truth must share a float32/float64 dtype')
  for x in (ens, truth):
    if x.device != dev or not x.is_contiguous():
      raise ValueError('inputs must be contiguous on one device')
  if mean_over is not None:
    import ctypes
    n_lead, n_time, n_tail = (int(v) for v in mean_over)
    if n_lead * n_time * n_tail != n_outer or n_time < 1:
      raise ValueError(f'mean_over={mean_over} does not factor {n_outer}')
    out = torch.empty((n_lead * n_tail, n_point, n_bins), dtype=torch.float64,
                      device=dev)
    pcg = st = ref_off = None
    n_col = 1
    if numpy_stream is not None and break_ties:
      state, inc, ref_off, strides, n_col = numpy_stream
      mask = (1 << 64) - 1
      pcg = (ctypes.c_uint64 * 4)(state >> 64, state & mask, inc >> 64,
                                  inc & mask)
      st = (ctypes.c_int64 * 3)(*[int(v) for v in strides])
      if ref_off.dtype != torch.int64 or ref_off.numel() != n_outer:
        raise ValueError('ref_outer_off must be int64[n_outer]')
    _lib.check(lib.wb2_rank_histogram_mean(
        _DTYPES[ens.dtype], _lib.ptr(ens), _lib.ptr(ens_slab), _lib.ptr(truth),
        _lib.ptr(truth_slab), n_member, member_stride, n_lead, n_time, n_tail,
        n_point, int(n_col), n_bins, int(break_ties),
        seed & 0xFFFFFFFFFFFFFFFF, pcg, _lib.ptr(ref_off), st, 1,
        _lib.ptr(out), current_stream_ptr(dev)), 'wb2_rank_histogram_mean')
    return out
  if acc_row is None:
    out = torch.empty((n_outer, n_point, n_bins), dtype=torch.float64,
                      device=dev)
  else:
    out = torch.zeros((n_acc, n_point, n_bins), dtype=torch.float64,
                      device=dev)
  if numpy_stream is not None and break_ties:
    import ctypes
    state, inc, ref_off, strides, n_col = numpy_stream
    mask = (1 << 64) - 1
    pcg = (ctypes.c_uint64 * 4)(state >> 64, state & mask, inc >> 64, inc & mask)
    st = (ctypes.c_int64 * 3)(*[int(v) for v in strides])
    if ref_off.dtype != torch.int64 or ref_off.numel() != n_outer:
      raise ValueError('ref_outer_off must be int64[n_outer]')
    _lib.check(lib.wb2_rank_histogram_seeded(
        _DTYPES[ens.dtype], _lib.ptr(ens), _lib.ptr(ens_slab), _lib.ptr(truth),
        _lib.ptr(truth_slab), n_member, member_stride, n_outer, n_point,
        int(n_col), n_bins, pcg, _lib.ptr(ref_off), st, _lib.ptr(acc_row),
        _lib.ptr(out), current_stream_ptr(dev)), 'wb2_rank_histogram_seeded')
    return out
  _lib.check(lib.wb2_rank_histogram(
      _DTYPES[ens.dtype], _lib.ptr(ens), _lib.ptr(ens_slab), _lib.ptr(truth),
      _lib.ptr(truth_slab), n_member, member_stride, n_outer, n_point, n_bins,
      int(break_ties), seed & 0xFFFFFFFFFFFFFFFF, _lib.ptr(acc_row),
      _lib.ptr(out), current_stream_ptr(dev)), 'wb2_rank_histogram')
  return out


def ensemble_threshold_maps(ens: torch.Tensor, member_stride: int,
                            n_member: int, ens_slab, truth: torch.Tensor,
                            truth_slab, thr: torch.Tensor, thr_slab,
                            n_outer: int, n_point: int,
                            skipna: bool) -> torch.Tensor:
  """wb2_ens_threshold_maps: [4, n_outer, n_point] float64 (Brier, debiased
  Brier, ignorance, RPS part), unreduced."""
  lib = _lib.load()
  dev = ens.device
  if ens.dtype not in _DTYPES or truth.dtype != ens.dtype or (
      thr.dtype != ens.dtype):
    raise TypeError('members, truth and threshold must share one dtype')
  for x in (ens, truth, thr):
    if x.device != dev or not x.is_contiguous():
      raise ValueError('inputs must be contiguous on one device')
  maps = torch.empty((4, n_outer, n_point), dtype=torch.float64, device=dev)
  _lib.check(lib.wb2_ens_threshold_maps(
      _DTYPES[ens.dtype], int(skipna), _lib.ptr(ens), _lib.ptr(ens_slab),
      _lib.ptr(truth), _lib.ptr(truth_slab), _lib.ptr(thr), _lib.ptr(thr_slab),
      n_member, member_stride, n_outer, n_point, _lib.ptr(maps),
      current_stream_ptr(dev)), 'wb2_ens_threshold_maps')
  return maps


def seeps_map(inputs: t.Sequence[torch.Tensor],
              slabs: t.Sequence[t.Optional[torch.Tensor]], n_outer: int,
              n_point: int, aux: torch.Tensor, scalar: float) -> torch.Tensor:
  """wb2_seeps_map: [n_outer, n_point] float64 per-point SEEPS."""
  lib = _lib.load()
  dev = inputs[0].device
  dtype = inputs[0].dtype
  if dtype not in _DTYPES:
    raise TypeError(f'unsupported dtype {dtype}')
  for x in inputs:
    if x.dtype != dtype or x.device != dev or not x.is_contiguous():
      raise ValueError('inputs must share dtype/device and be contiguous')
  if aux.dtype != torch.float64 or aux.numel() != n_point:
    raise ValueError('aux must be float64[n_point]')
  out = torch.empty((n_outer, n_point), dtype=torch.float64, device=dev)
  _lib.check(lib.wb2_seeps_map(
      _DTYPES[dtype], _lib.ptr_array(inputs), _lib.ptr_array(slabs), n_outer,
      n_point, _lib.ptr(aux), float(scalar), _lib.ptr(out),
      current_stream_ptr(dev)), 'wb2_seeps_map')
  return out


def axis_moments(x: torch.Tensor, n_lead: int, n_red: int, n_tail: int,
                 w_red: t.Optional[torch.Tensor], skipna: bool,
                 want_sq: bool = False, w_repeat: int = 1):
  """wb2_axis_moments on a contiguous [n_lead, n_red, n_tail] view of `x`:
  (sum, sumsq or None, count) as float64 tensors of n_lead * n_tail.  `w_red`
  holds n_red / w_repeat weights, each shared by w_repeat consecutive r."""
  lib = _lib.load()
  dev = x.device
  if x.dtype not in _DTYPES or not x.is_contiguous():
    raise ValueError('x must be a contiguous float32/float64 device tensor')
  if x.numel() != n_lead * n_red * n_tail:
    raise ValueError('shape mismatch')
  if w_red is not None and (w_red.dtype != torch.float64
                            or w_red.numel() * w_repeat != n_red):
    raise ValueError('w_red must be float64[n_red / w_repeat]')
  n_out = n_lead * n_tail
  n_split = lib.wb2_axis_moments_splits(n_lead, n_red, n_tail,
                                        w_repeat if w_red is not None else 1)
  work = torch.empty((3 * n_split * max(n_out, 1),), dtype=torch.float64,
                     device=dev)
  total = torch.empty((n_out,), dtype=torch.float64, device=dev)
  count = torch.empty_like(total)
  sq = torch.empty_like(total) if want_sq else None
  _lib.check(lib.wb2_axis_moments(
      _DTYPES[x.dtype], _lib.ptr(x), n_lead, n_red, n_tail, _lib.ptr(w_red),
      w_repeat, int(skipna), n_split, _lib.ptr(work), _lib.ptr(total),
      _lib.ptr(sq), _lib.ptr(count), current_stream_ptr(dev)),
             'wb2_axis_moments')
  return total, sq, count


# ---------------------------------------------------------------------------
# The path's one exchange step through the C ABI (RCCL directly, no
# torch.distributed): for callers that bring their own rendezvous.
# ---------------------------------------------------------------------------
def comm_unique_id() -> bytes:
  """128-byte RCCL id made by rank 0, to be handed to every rank."""
  import ctypes
  buf = ctypes.create_string_buffer(128)
  _lib.check(_lib.load().wb2_comm_unique_id(buf), 'wb2_comm_unique_id')
  return buf.raw


def comm_init_rank(unique_id: bytes, n_ranks: int, rank: int):
  """ncclComm_t (opaque handle) of this rank on the current device."""
  import ctypes
  if len(unique_id) != 128:
    raise ValueError('unique_id must be the 128 bytes of comm_unique_id()')
  handle = ctypes.c_void_p()
  _lib.check(_lib.load().wb2_comm_init_rank(
      ctypes.create_string_buffer(unique_id, 128), n_ranks, rank,
      ctypes.byref(handle)), 'wb2_comm_init_rank')
  return handle


def comm_destroy(comm) -> None:
  _lib.check(_lib.load().wb2_comm_destroy(comm), 'wb2_comm_destroy')


def time_mean_allreduce(total: torch.Tensor, count: torch.Tensor, comm) -> None:
  """In-place sum of the (sum, count) accumulators over the ranks of `comm`
  (xbeam.Mean's combiner across init-time shards, evaluation.py:735-744), one
  grouped RCCL operation on the current stream."""
  for x in (total, count):
    if x.dtype != torch.float64 or not x.is_contiguous() or not x.is_cuda:
      raise ValueError('accumulators must be contiguous float64 device tensors')
  if total.numel() != count.numel():
    raise ValueError('sum / count size mismatch')
  _lib.check(_lib.load().wb2_time_mean_allreduce(
      _lib.ptr(total), _lib.ptr(count), total.numel(), comm,
      current_stream_ptr(total.device)), 'wb2_time_mean_allreduce')
