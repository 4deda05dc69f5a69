"""K3's other production instantiations at the BASELINE configs[2] launch size
(13 slabs x 50 members of 721 x 1440 float32): the official `probabilistic`
config's 16 regions incl. three land-sea-mask regions (scripts/evaluate.py:
345-395, 496-520: the WF = true instantiation), skipna, a global-only region
set (`regions=None`), 51 and 30 members (runtime-M kernels), float64.

  python tools/k3_variants.py         -> one JSON line
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from weatherbench2_amd import engine, plan as plan_lib  # noqa: E402


def main():
  dev = torch.device('cuda', 0)
  n_lat, n_lon, n_slab = bench.N_LAT, bench.N_LON, 13
  lat = np.linspace(-90, 90, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  rows = plan_lib.ENSEMBLE_ROWS_PER_CHUNK
  plans = {
      'slice13': plan_lib.build_plan(lat, lon, plan_lib.LATLON,
                                     bench.predefined_regions(), dev,
                                     rows_per_chunk=rows),
      'official16': plan_lib.build_plan(lat, lon, plan_lib.LATLON,
                                        bench.official_regions(), dev,
                                        rows_per_chunk=rows),
      'global': plan_lib.build_plan(lat, lon, plan_lib.LATLON, {'global': None},
                                    dev, rows_per_chunk=rows),
  }
  gen = torch.Generator(device=dev).manual_seed(5)
  pool = 4
  slab = n_lat * n_lon
  out = {}

  def run(name, plan, m, dtype, skipna):
    ens = torch.randn((m, pool * n_slab, n_lat, n_lon), generator=gen,
                      device=dev, dtype=dtype)
    truth = torch.randn((pool * n_slab, n_lat, n_lon), generator=gen,
                        device=dev, dtype=dtype)
    tabs = [torch.arange(n_slab, device=dev) + k * n_slab for k in range(pool)]
    stride = pool * n_slab * slab
    timer = bench.KernelTimer()
    it = [0]

    def step():
      it[0] += 1
      tab = tabs[it[0] % pool]
      engine.ensemble_reduce(plans[plan], ens, stride, m, tab, truth, tab,
                             n_slab, skipna)
    for _ in range(3):
      step()
    bench.ramp(step, 20.0)
    engine.set_launch_hook(timer)
    for _ in range(30):
      step()
    engine.set_launch_hook(None)
    torch.cuda.synchronize()
    ms = timer.mean_ms()
    nbytes = n_slab * slab * (m + 1) * ens.element_size()
    out[name] = {'kernel_ms': ms, 'frac': nbytes / ms / 1e6 / bench.HBM_PEAK_GBPS,
                 'members': m, 'dtype': str(dtype).split('.')[-1],
                 'skipna': skipna, 'regions': plans[plan].n_region}
    del ens, truth

  run('headline_slice13', 'slice13', 50, torch.float32, False)
  run('official16_landmask', 'official16', 50, torch.float32, False)
  run('global_only', 'global', 50, torch.float32, False)
  run('skipna', 'slice13', 50, torch.float32, True)
  run('members51', 'slice13', 51, torch.float32, False)
  run('members30', 'slice13', 30, torch.float32, False)
  run('members10', 'slice13', 10, torch.float32, False)
  run('f64_members50', 'slice13', 50, torch.float64, False)
  print(json.dumps(out))


if __name__ == '__main__':
  main()
