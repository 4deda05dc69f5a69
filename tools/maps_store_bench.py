"""Map-writing kernels timed (K5 spatial_maps, K3 with the six pointwise maps):
for A/B runs of the store cache policy.  python tools/maps_store_bench.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from weatherbench2_amd import engine, plan as plan_lib  # noqa: E402


def timed(fn, steps=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
  ev[0].record()
  for _ in range(steps):
    fn()
  ev[1].record()
  torch.cuda.synchronize()
  return ev[0].elapsed_time(ev[1]) / steps


def main():
  dev = torch.device('cuda', 0)
  n_lat, n_lon, n_lev = 721, 1440, 13
  n_point = n_lat * n_lon
  gen = torch.Generator(device=dev).manual_seed(1)
  n_outer = 16 * n_lev
  pool = 3
  f = torch.randn((pool * n_outer, n_point), generator=gen, device=dev)
  t = torch.randn((pool * n_outer, n_point), generator=gen, device=dev)
  it = [0]

  def maps():
    i = it[0] = (it[0] + 1) % pool
    engine.spatial_maps(f[i * n_outer:(i + 1) * n_outer], None,
                        t[i * n_outer:(i + 1) * n_outer], None, n_outer,
                        n_point)

  out = {'spatial_maps_ms': timed(maps)}
  out['spatial_maps_TBps'] = n_outer * n_point * 4 * 5 / out['spatial_maps_ms'] / 1e9
  del f, t
  m, n_slab = 50, 13
  pl = plan_lib.build_plan(
      np.linspace(-90, 90, n_lat), np.linspace(0, 360, n_lon, endpoint=False),
      plan_lib.LATLON, {'global': None}, dev,
      rows_per_chunk=plan_lib.ENSEMBLE_ROWS_PER_CHUNK)
  ens = torch.randn((m, 2 * n_slab, n_lat, n_lon), generator=gen, device=dev)
  truth = torch.randn((2 * n_slab, n_lat, n_lon), generator=gen, device=dev)
  maps_out = torch.empty((6, n_slab, n_point), dtype=torch.float64, device=dev)
  tabs = [torch.arange(n_slab, device=dev) + k * n_slab for k in range(2)]

  def ens_maps():
    i = it[0] = (it[0] + 1) % 2
    engine.ensemble_reduce(pl, ens, 2 * n_slab * n_point, m, tabs[i], truth,
                           tabs[i], n_slab, False, maps=maps_out)

  out['ens_maps_ms'] = timed(ens_maps)
  out['ens_maps_TBps'] = n_slab * n_point * (51 * 4 + 48) / out['ens_maps_ms'] / 1e9
  print(json.dumps(out))


if __name__ == '__main__':
  main()
