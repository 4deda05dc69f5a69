"""wb2_spatial_accumulate_addr alone (the fused kernel of map_suite.py) at the
official chunk's size: 85 slabs of 721 x 1440 float32 per chunk into float64
running sums of 4 lead rows, a pool of distinct chunks (no cache re-use).

  python tools/map_accumulate_bench.py [--skipna] [--steps N] [--reps R]
  WB2HIP_LIB=build/variants/libwb2hip_X.so python tools/map_accumulate_bench.py

Prints one JSON line: ms per launch (HIP events), GB/s of the algorithmic
56 B/pt (8 read + 3 x 16 read-modify-write; 104 B/pt with skipna counts)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from weatherbench2_amd import _lib, engine

N_POINT, N_SLAB = 721 * 1440, 85


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--skipna', action='store_true')
  ap.add_argument('--steps', type=int, default=1, help='time steps per launch')
  ap.add_argument('--pool', type=int, default=12)
  ap.add_argument('--reps', type=int, default=60)
  args = ap.parse_args()
  dev = torch.device('cuda:0')
  lib = _lib.load()
  f = torch.randn((args.pool, N_SLAB, N_POINT), device=dev)
  t = torch.randn((args.pool, N_SLAB, N_POINT), device=dev)
  total = torch.zeros((4, 3, N_SLAB, N_POINT), dtype=torch.float64, device=dev)
  count = torch.zeros_like(total) if args.skipna else None
  tables = []
  for i in range(args.pool):
    picks = [(i + k) % args.pool for k in range(args.steps)]
    fa = np.stack([f[p].data_ptr() + 4 * N_POINT * np.arange(N_SLAB)
                   for p in picks])
    ta = np.stack([t[p].data_ptr() + 4 * N_POINT * np.arange(N_SLAB)
                   for p in picks])
    row = i % 4
    off = 8 * N_POINT * (np.arange(3)[:, None] * N_SLAB + np.arange(N_SLAB))
    sa = total[row].data_ptr() + off
    ca = (count[row].data_ptr() + off) if args.skipna else np.zeros_like(off)
    tab = torch.as_tensor(np.concatenate(
        [fa.ravel(), ta.ravel(), sa.ravel(), ca.ravel()])).to(dev)
    tables.append(tab)
  stream = engine.current_stream_ptr(dev)
  n = args.steps * N_SLAB

  def launch(i):
    base = tables[i % args.pool].data_ptr()
    _lib.check(lib.wb2_spatial_accumulate_addr(
        _lib.WB2_F32, int(args.skipna), 1, base, base + 8 * n, args.steps,
        N_SLAB, N_POINT, base + 16 * n,
        (base + 16 * n + 24 * N_SLAB) if args.skipna else None, stream),
               'wb2_spatial_accumulate_addr')
  for i in range(5):
    launch(i)
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(
      enable_timing=True)
  a.record()
  for i in range(args.reps):
    launch(i)
  b.record()
  torch.cuda.synchronize()
  ms = a.elapsed_time(b) / args.reps
  per_pt = 8.0 * args.steps + (96.0 if args.skipna else 48.0)
  gbps = N_SLAB * N_POINT * per_pt / ms / 1e6
  print(json.dumps({'lib': os.environ.get('WB2HIP_LIB', 'default'),
                    'skipna': args.skipna, 'steps': args.steps,
                    'ms_per_launch': ms, 'bytes_per_point': per_pt,
                    'GBps': gbps, 'frac_of_8TBps': gbps / 8000.0}))


if __name__ == '__main__':
  main()
