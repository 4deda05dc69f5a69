"""The host-fed leg of tools/official_chunk.py alone, beside this box's pinned
copy rate (864 MB transfers on a copy stream, what bench.py's pcie_inclusive
leg moves) and the uploader alone."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, ROOT)
import torch
import official_chunk as oc

dev = torch.device('cuda', 0)
n = 16 * 13 * 721 * 1440
pinned = torch.empty((n,), dtype=torch.float32).pin_memory()
dst = torch.empty((n,), dtype=torch.float32, device=dev)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
  dst.copy_(pinned, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(st):
  for _ in range(6):
    dst.copy_(pinned, non_blocking=True)
torch.cuda.synchronize()
rate = 6 * n * 4 / (time.perf_counter() - t0) / 1e9
del pinned, dst
chunks, cfg = oc.build(dev, 160, 32)
r = oc.measure_host_fed(chunks, cfg)
print(json.dumps({'pinned_864MB_GBps': round(rate, 1),
                  'host_fed_GBps': {k: round(v['h2d_GBps'], 1)
                                    for k, v in r['by_window'].items()},
                  'uploader_alone_GBps': round(r['uploader_alone_GBps'], 1),
                  'ratio': round(r['h2d_GBps'] / rate, 3),
                  'slots': os.environ.get('WB2HIP_STAGE_SLOTS', 'default')}))
