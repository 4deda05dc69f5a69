"""Rate of the pageable-host -> HBM uploader alone (csrc/staging.cpp) for the
calling environment's WB2HIP_COPY_THREADS / WB2HIP_STAGE_MEMCPY: one official
chunk's 13 variables (353 MB) uploaded back to back.

  WB2HIP_COPY_THREADS=16 python tools/upload_sweep.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  import torch
  from weatherbench2_amd import feeder
  dev = torch.device('cuda', 0)
  rs = np.random.default_rng(0)
  arrays = [rs.standard_normal((13, 721, 1440), dtype=np.float32)
            for _ in range(6)] + [
                rs.standard_normal((721, 1440), dtype=np.float32)
                for _ in range(7)]
  nbytes = sum(a.nbytes for a in arrays)
  for _ in range(2):
    keep = feeder.upload_many(arrays, dev)
  torch.cuda.synchronize()
  reps = 6
  t0 = time.perf_counter()
  for _ in range(reps):
    keep = feeder.upload_many(arrays, dev)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  pinned = torch.empty(nbytes // 4, dtype=torch.float32).pin_memory()
  dst = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
  dst.copy_(pinned, non_blocking=True)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps):
    dst.copy_(pinned, non_blocking=True)
  torch.cuda.synchronize()
  dp = time.perf_counter() - t0
  print(json.dumps({
      'copy_threads': feeder.copy_threads(),
      'plain_memcpy': os.environ.get('WB2HIP_STAGE_MEMCPY') == '1',
      'slice_MiB': feeder._SLICE_BYTES >> 20, 'slots': feeder._RING_SLOTS,
      'pageable_GBps': reps * nbytes / dt / 1e9,
      'pinned_GBps': reps * nbytes / dp / 1e9}))


if __name__ == '__main__':
  main()
