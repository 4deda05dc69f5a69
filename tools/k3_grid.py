"""K3 (+ its fold) on the grids of the documented ENS command lines (240 x 121,
64 x 32: docs/source/official-evaluation.md:765-860) beside the 0.25-degree
grid, at launches of about the same bytes: does the kernel hold its fraction
on small slabs?

  python tools/k3_grid.py [--rows R] [--members M]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--rows', type=int, default=None)
  ap.add_argument('--members', type=int, default=50)
  ap.add_argument('--gb', type=float, default=2.7)
  ap.add_argument('--grids', default='1440x721,240x121,64x32')
  ap.add_argument('--iters', type=int, default=20)
  args = ap.parse_args()
  import torch
  import bench
  from weatherbench2_amd import engine, plan as plan_lib
  dev = torch.device('cuda:0')
  rows = args.rows or plan_lib.ENSEMBLE_ROWS_PER_CHUNK
  m = args.members
  out = {'rows_per_chunk': rows, 'members': m}
  for grid in args.grids.split(','):
    n_lon, n_lat = (int(x) for x in grid.split('x'))
    lat = np.linspace(-90, 90, n_lat)
    lon = np.linspace(0, 360, n_lon, endpoint=False)
    pl = plan_lib.build_plan(lat, lon, plan_lib.LATLON,
                             bench.predefined_regions(), dev,
                             rows_per_chunk=rows)
    slab = n_lat * n_lon
    n_slab = max(1, int(args.gb * 1e9 / (slab * 4 * (m + 1))))
    ens = torch.randn((m, n_slab, n_lat, n_lon), device=dev)
    truth = torch.randn((n_slab, n_lat, n_lon), device=dev)
    run = lambda: engine.ensemble_reduce(pl, ens, n_slab * slab, m, None, truth,
                                         None, n_slab, False)
    for _ in range(max(3, args.iters // 4)):
      run()
    torch.cuda.synchronize()
    e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
    n = args.iters
    e0.record()
    for _ in range(n):
      run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    nbytes = n_slab * slab * 4 * (m + 1)
    out[grid] = {'slabs': n_slab, 'ms': ms, 'GBps': nbytes / ms / 1e6,
                 'frac': nbytes / ms / 1e6 / 8000.0}
    del ens, truth
  print(json.dumps(out))


if __name__ == '__main__':
  main()
