"""Result download rates: tensor.cpu() against feeder.download (the pinned
ring the other way round, wb2_uploader_download) for one float64 map variable
of `deterministic_spatial` (3 metrics x 4 leads x 13 levels x 721 x 1440 =
1.3 GB), and the same into a destination whose pages already exist."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from weatherbench2_amd import feeder, _lib

dev = torch.device('cuda', 0)
x = torch.randn((3, 4, 13, 721, 1440), dtype=torch.float64, device=dev)
gb = x.numel() * 8 / 1e9
out = {'bytes': x.numel() * 8, 'copy_threads': feeder.copy_threads(),
       'slots': feeder._RING_SLOTS, 'slice_MiB': feeder._SLICE_BYTES >> 20}
feeder.download(x[:1])   # creates the ring
torch.cuda.synchronize()
for name, fn in (('tensor_cpu', lambda: x.cpu().numpy()),
                 ('ring', lambda: feeder.download(x))):
  rates = []
  for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = fn()
    rates.append(gb / (time.perf_counter() - t0))
    assert y.shape == tuple(x.shape)
    del y
  out[name + '_GBps'] = [round(r, 2) for r in rates]
# destination pages that exist already (what is left is the two copies)
ring = feeder._STAGING.get(dev)
dst = np.empty(tuple(x.shape))
dst.fill(0.0)
rates = []
for _ in range(3):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  _lib.check(ring['lib'].wb2_uploader_download(
      ring['uploader'], dst.ctypes.data, x.data_ptr(), dst.nbytes,
      torch.cuda.current_stream(dev).cuda_stream), 'download')
  rates.append(gb / (time.perf_counter() - t0))
out['ring_touched_destination_GBps'] = [round(r, 2) for r in rates]
assert np.array_equal(dst, x.cpu().numpy())
print(json.dumps(out))
