import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from weatherbench2_amd import _lib, engine, plan as plan_lib
dev = torch.device('cuda', 0)
lat = np.linspace(-90, 90, 721); lon = np.linspace(0, 360, 1440, endpoint=False)
pl = plan_lib.build_plan(lat, lon, plan_lib.LATLON, bench.predefined_regions(), dev)
units = 16
f = torch.randn((units * 13, 721, 1440), device=dev); t = torch.randn_like(f); c = torch.randn_like(f)
lib = _lib.load()
for it in range(3):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  m, _ = engine.stream_reduce(pl, _lib.MODE_DET_ACC, [f, t, c], [None, None, None], units * 13, False)
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  print(f'iter {it}: host enqueue {1e3*(t1-t0):.3f} ms, total {1e3*(t2-t0):.3f} ms')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
  engine.stream_reduce(pl, _lib.MODE_DET_ACC, [f, t, c], [None, None, None], units * 13, False)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
