"""Host -> HBM chunk feeder: pinned staging ring + copy stream.

The reference loads every chunk on the host (`_sel_corresponding_truth_chunk` /
`_climatology_like_forecast_chunk`, /root/reference/weatherbench2/evaluation.py:
601-649) and hands NumPy arrays to the metrics.  Here host arrays cross PCIe
through page-locked staging buffers on a dedicated copy stream, so that the
transfer of chunk i + 1 overlaps the fused pass over chunk i:

  * `upload(array, device)`: one pageable NumPy array -> device tensor, moved in
    slices through a pinned ring by a pool of copy threads (csrc/staging.cpp:
    the staging copy of slice k + 1 overlaps the DMA of slice k; one thread's
    memcpy is a sixth of the link); used by engine.as_device_tensor.
  * `ChunkFeeder`: a depth-N ring of device buffers for a stream of equally
    shaped chunks; `submit()` starts the copy of the next chunk on the copy
    stream, `acquire()` makes the compute stream wait for it (no host sync).

Truth and climatology should NOT travel per chunk: they stay resident in HBM
and are gathered by valid time through slab tables; the
feeder is for the forecast stream (4 of the 12 B per grid point).
"""
from __future__ import annotations

import threading
import typing as t
import weakref

import numpy as np
import torch

# the pinned ring of an uploader: slices small enough that the first DMA starts
# early, enough of them that the copy pool never waits for a free slot
import os as _os

_SLICE_BYTES = int(_os.environ.get('WB2HIP_STAGE_SLICE_MIB', 32)) << 20
# (8 slots: with 4 the copy pool waits for the DMA that frees its next slot --
# 44.8 GB/s against 53.6 on the same box, profiles/r06_upload_sweep.txt)
_RING_SLOTS = int(_os.environ.get('WB2HIP_STAGE_SLOTS', 8))


def copy_threads() -> int:
  """Threads of the staging copy (WB2HIP_COPY_THREADS; default 8, fewer on
  small hosts): one thread's streaming copy moves ~8-10 GB/s, the link ~57;
  measured 53.7 / 52.9 / 52.8 / 48.8 GB/s with 8 / 16 / 32 / 64 threads against
  57.2 GB/s from pinned memory (profiles/r05_upload_sweep.txt)."""
  import os
  env = os.environ.get('WB2HIP_COPY_THREADS')
  if env:
    return max(1, int(env))
  return int(min(8, max(2, (os.cpu_count() or 8) // 2)))


class _Ring(dict):
  """One uploader (pinned ring + copy pool, csrc/staging.cpp) + its copy
  stream.  The native side is destroyed with the object: the thread-local
  table below dies with its thread (evaluate_chunks starts a fetch thread per
  call), and an uploader that outlived it would keep its pinned slots (4 x 32
  MiB), its copy threads, events and stream for the life of the process."""

  def __init__(self, lib, handle, stream):
    super().__init__(uploader=handle, lib=lib, stream=stream)
    self._finalizer = weakref.finalize(self, _Ring._destroy, lib, handle.value)

  @staticmethod
  def _destroy(lib, handle):
    try:
      lib.wb2_uploader_destroy(handle)  # waits for the DMAs out of its slots
    except Exception:  # interpreter shutdown: the process frees it anyway
      pass

  def close(self):
    self._finalizer()


UPLOADERS_ALIVE = lambda: sum(1 for r in _RINGS if r() is not None
                              and r()._finalizer.alive)
_RINGS: list = []   # weak references (diagnostics, tests)


class _Staging(threading.local):
  """Per-thread uploader and copy stream per device."""

  def __init__(self):
    self.rings: dict = {}

  def get(self, device: torch.device):
    import ctypes
    from weatherbench2_amd import _lib
    key = (device.type, device.index)
    ring = self.rings.get(key)
    if ring is None:
      lib = _lib.load()
      handle = ctypes.c_void_p()
      with torch.cuda.device(device):
        _lib.check(lib.wb2_uploader_create(copy_threads(), _SLICE_BYTES,
                                           _RING_SLOTS, ctypes.byref(handle)),
                   'wb2_uploader_create')
      ring = _Ring(lib, handle, torch.cuda.Stream(device=device))
      self.rings[key] = ring
      _RINGS[:] = [r for r in _RINGS if r() is not None] + [weakref.ref(ring)]
    return ring

  def close(self):
    """Destroys the calling thread's uploaders now (a thread that is about to
    exit; otherwise the finalizers do it when the thread's table is
    collected)."""
    for ring in self.rings.values():
      ring.close()
    self.rings.clear()


def close_thread_uploaders() -> None:
  _STAGING.close()


_STAGING = _Staging()


def copy_stream(device: torch.device) -> torch.cuda.Stream:
  return _STAGING.get(device)['stream']


def upload(array: np.ndarray, device: torch.device,
           wait: bool = True) -> torch.Tensor:
  """Contiguous (pageable) NumPy array -> new device tensor of the same shape /
  dtype through the calling thread's uploader: the copy pool stages slice
  k + 1 into the pinned ring while the DMA of slice k runs on the copy stream
  (wb2_uploader_upload; ctypes releases the GIL for the whole call).

  Returns once the last slice has been staged (the source may be reused); the
  caller's current stream waits for the copy stream before it may read the
  result (stream-ordered, no device synchronisation) unless `wait` is False --
  then the caller orders the read itself (`copy_stream(device)`)."""
  from weatherbench2_amd import _lib
  src = np.ascontiguousarray(array)
  ring = _STAGING.get(device)
  stream = ring['stream']
  cur = torch.cuda.current_stream(device)
  # The destination belongs to the COPY stream's pool: a recycled block is safe
  # to overwrite in copy-stream order, so the copy never waits for kernels
  # queued on the compute stream (the copy of chunk i + 1 overlaps the pass
  # over chunk i); the consumer's use is recorded so that the block is not
  # handed out again before that stream is done with it.
  with torch.cuda.stream(stream):
    dst = torch.empty(src.shape,
                      dtype=torch.from_numpy(np.empty(0, src.dtype)).dtype,
                      device=device)
  dst.record_stream(cur)
  nbytes = src.nbytes
  if nbytes == 0:
    return dst
  _lib.check(ring['lib'].wb2_uploader_upload(
      ring['uploader'], dst.data_ptr(), src.ctypes.data, nbytes,
      stream.cuda_stream), 'wb2_uploader_upload')
  if wait:
    cur.wait_stream(stream)
  return dst


def upload_many(arrays: t.Sequence[np.ndarray], device: torch.device,
                wait: bool = True) -> list:
  """`upload` for the variables of one chunk in ONE C-ABI call
  (wb2_uploader_upload_many): the calling thread does not come back to the
  interpreter -- and does not queue for its lock behind a busy main thread --
  between two variables."""
  import ctypes
  from weatherbench2_amd import _lib
  srcs = [np.ascontiguousarray(a) for a in arrays]
  ring = _STAGING.get(device)
  stream = ring['stream']
  cur = torch.cuda.current_stream(device)
  with torch.cuda.stream(stream):
    dsts = [torch.empty(a.shape,
                        dtype=torch.from_numpy(np.empty(0, a.dtype)).dtype,
                        device=device) for a in srcs]
  for d in dsts:
    d.record_stream(cur)
  n = len(srcs)
  if n:
    _lib.check(ring['lib'].wb2_uploader_upload_many(
        ring['uploader'], n, (ctypes.c_void_p * n)(*[d.data_ptr() or None
                                                     for d in dsts]),
        (ctypes.c_void_p * n)(*[a.ctypes.data if a.nbytes else None
                                for a in srcs]),
        (ctypes.c_int64 * n)(*[a.nbytes for a in srcs]), stream.cuda_stream),
               'wb2_uploader_upload_many')
  if wait:
    cur.wait_stream(stream)
  return dsts


# results below this size leave through torch's own copy (one staged DMA)
_DOWNLOAD_MIN_BYTES = 16 << 20


def download(tensor: torch.Tensor) -> np.ndarray:
  """Device tensor -> new (pageable) NumPy array of the same shape / dtype
  through the calling thread's ring (wb2_uploader_download): the DMAs of the
  next slices run behind the tensor's producer on the CURRENT stream while the
  copy pool moves the slice that has arrived out of its pinned slot.
  `tensor.cpu()` stages a pageable destination through ONE bounce buffer
  (~7 GB/s on the boxes measured); the float64 maps of the Spatial* metrics are
  gigabytes per variable (RunningMean.result)."""
  from weatherbench2_amd import _lib
  if tensor.device.type != 'cuda' or (
      tensor.numel() * tensor.element_size() < _DOWNLOAD_MIN_BYTES):
    return tensor.cpu().numpy()
  src = tensor.contiguous()
  out = np.empty(tuple(src.shape),
                 dtype=torch.empty(0, dtype=src.dtype).numpy().dtype)
  ring = _STAGING.get(src.device)
  with torch.cuda.device(src.device):
    _lib.check(ring['lib'].wb2_uploader_download(
        ring['uploader'], out.ctypes.data, src.data_ptr(), out.nbytes,
        torch.cuda.current_stream(src.device).cuda_stream),
               'wb2_uploader_download')
  return out


class ChunkFeeder:
  """Ring of `depth` device buffers fed from host memory on a copy stream.

      feeder = ChunkFeeder(shape, torch.float32, device, depth=2)
      feeder.submit(host_chunk_0)
      for i in range(n):
        if i + 1 < n: feeder.submit(host_chunk[i + 1])   # overlaps the pass on i
        x = feeder.acquire()        # compute stream waits for chunk i's copy
        ... launch the fused pass on x ...
        feeder.release()            # the copy stream may overwrite x afterwards

  `submit` accepts pinned torch tensors (copied straight from where they are) or
  NumPy arrays (staged through a pinned buffer of the ring).
  """

  def __init__(self, shape: t.Sequence[int], dtype: torch.dtype,
               device: torch.device, depth: int = 2):
    self.device = device
    self.stream = torch.cuda.Stream(device=device)
    self.buffers = [torch.empty(tuple(shape), dtype=dtype, device=device)
                    for _ in range(depth)]
    self.pinned = [None] * depth
    self.ready = [None] * depth   # copy finished (recorded on the copy stream)
    self.free = [None] * depth    # pass finished (recorded on the compute stream)
    self.head = 0                 # next slot to fill
    self.tail = 0                 # next slot to consume
    self.depth = depth

  def submit(self, host) -> None:
    i = self.head % self.depth
    self.head += 1
    if self.free[i] is not None:
      self.stream.wait_event(self.free[i])
    if isinstance(host, torch.Tensor):
      if not host.is_pinned():
        raise ValueError('host tensors must be pinned (page-locked)')
      src = host
    else:
      arr = np.ascontiguousarray(host)
      if self.pinned[i] is None:
        self.pinned[i] = torch.empty(self.buffers[i].shape,
                                     dtype=self.buffers[i].dtype).pin_memory()
      if self.ready[i] is not None:
        self.ready[i].synchronize()  # the previous DMA out of this buffer
      from weatherbench2_amd import _lib
      _lib.check(_lib.load().wb2_host_copy(
          self.pinned[i].data_ptr(), arr.ctypes.data, arr.nbytes,
          copy_threads()), 'wb2_host_copy')
      src = self.pinned[i]
    with torch.cuda.stream(self.stream):
      self.buffers[i].copy_(src.reshape(self.buffers[i].shape),
                            non_blocking=True)
      ev = torch.cuda.Event()
      ev.record(self.stream)
    self.ready[i] = ev

  def acquire(self) -> torch.Tensor:
    i = self.tail % self.depth
    if self.tail >= self.head:
      raise RuntimeError('acquire() without a submitted chunk')
    torch.cuda.current_stream(self.device).wait_event(self.ready[i])
    return self.buffers[i]

  def release(self) -> None:
    i = self.tail % self.depth
    self.tail += 1
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(self.device))
    self.free[i] = ev
