"""K7 (wb2_axis_moments) throughput: ensemble mean of 50 x 13 x 721 x 1440
float32 (strided kernel) and area-weighted global means of 208 slabs
(contiguous kernel).  HIP events around the calls, GB/s of algorithmic reads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from weatherbench2_amd import engine, plan

dev = torch.device('cuda', 0)


def timed(fn, reps=10):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps


pool = [torch.randn((50, 13 * 721 * 1440), device=dev) for _ in range(3)]
i = [0]
def ens():
  i[0] += 1
  return engine.axis_moments(pool[i[0] % 3], 1, 50, 13 * 721 * 1440, None, False)
ms = timed(ens)
print(f'ensemble mean 50 x 13 x 721 x 1440 f32: {ms:.3f} ms, '
      f'{50 * 13 * 721 * 1440 * 4 / ms / 1e6:.0f} GB/s')
ens_sk = lambda: engine.axis_moments(pool[0], 1, 50, 13 * 721 * 1440, None, True)
ms = timed(ens_sk)
print(f'  skipna: {ms:.3f} ms, {50 * 13 * 721 * 1440 * 4 / ms / 1e6:.0f} GB/s (same array: cache-assisted)')
x = [torch.randn((208 * 3, 721 * 1440), device=dev) for _ in range(2)]
w = torch.as_tensor(plan.get_lat_weights(np.linspace(-90, 90, 721)), device=dev)
def glob():
  i[0] += 1
  return engine.axis_moments(x[i[0] % 2], 208 * 3, 721 * 1440, 1, w, True, True, 1440)
ms = timed(glob)
print(f'weighted global moments of 624 slabs 721 x 1440 f32: {ms:.3f} ms, '
      f'{624 * 721 * 1440 * 4 / ms / 1e6:.0f} GB/s')
