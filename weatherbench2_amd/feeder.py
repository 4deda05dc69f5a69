"""Host -> HBM chunk feeder: pinned staging ring + copy stream.

The reference loads every chunk on the host (`_sel_corresponding_truth_chunk` /
`_climatology_like_forecast_chunk`, /root/reference/weatherbench2/evaluation.py:
601-649) and hands NumPy arrays to the metrics.  Here host arrays cross PCIe
through page-locked staging buffers on a dedicated copy stream, so that the
transfer of chunk i + 1 overlaps the fused pass over chunk i:

  * `upload(array, device)`: one pageable NumPy array -> device tensor, moved in
    slices through a two-slot pinned ring (the host memcpy of slice k + 1
    overlaps the DMA of slice k); used by engine.as_device_tensor.
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

import numpy as np
import torch

_SLICE_BYTES = 64 << 20


class _Staging(threading.local):
  """Per-thread pinned ring (two slots) and copy stream per device."""

  def __init__(self):
    self.rings: dict = {}

  def get(self, device: torch.device):
    key = (device.type, device.index)
    ring = self.rings.get(key)
    if ring is None:
      slots = [torch.empty(_SLICE_BYTES, dtype=torch.uint8).pin_memory()
               for _ in range(2)]
      ring = {'slots': slots, 'events': [None, None],
              'stream': torch.cuda.Stream(device=device), 'next': 0}
      self.rings[key] = ring
    return ring


_STAGING = _Staging()


def copy_stream(device: torch.device) -> torch.cuda.Stream:
  return _STAGING.get(device)['stream']


def upload(array: np.ndarray, device: torch.device) -> torch.Tensor:
  """Contiguous NumPy array -> new device tensor of the same shape / dtype.

  Asynchronous with respect to the host after the last slice has been staged;
  the caller's current stream waits for the copy stream before it may read the
  result (stream-ordered, no device synchronisation)."""
  src = np.ascontiguousarray(array)
  dst = torch.empty(src.shape, dtype=torch.from_numpy(np.empty(0, src.dtype)).dtype,
                    device=device)
  nbytes = src.nbytes
  if nbytes == 0:
    return dst
  ring = _STAGING.get(device)
  flat_src = src.reshape(-1).view(np.uint8)
  flat_dst = dst.reshape(-1).view(torch.uint8)
  stream = ring['stream']
  # the destination was allocated on the current stream: the copy stream must
  # not write before that allocation is safe to use
  stream.wait_stream(torch.cuda.current_stream(device))
  dst.record_stream(stream)
  off = 0
  while off < nbytes:
    n = min(_SLICE_BYTES, nbytes - off)
    i = ring['next']
    ring['next'] = 1 - i
    ev = ring['events'][i]
    if ev is not None:
      ev.synchronize()  # the DMA that last read this slot has finished
    slot = ring['slots'][i]
    np.copyto(slot.numpy()[:n], flat_src[off:off + n])
    with torch.cuda.stream(stream):
      flat_dst[off:off + n].copy_(slot[:n], non_blocking=True)
      ev = torch.cuda.Event()
      ev.record(stream)
    ring['events'][i] = ev
    off += n
  torch.cuda.current_stream(device).wait_stream(stream)
  return dst


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
      np.copyto(self.pinned[i].numpy(), arr.reshape(self.pinned[i].shape))
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
