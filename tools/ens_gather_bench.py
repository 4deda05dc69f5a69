"""forecast := probabilistic climatology (evaluation.py:458-470): K3 over an
ensemble whose members are slabs of the resident observations, gathered in
place (wb2_ens_partials_gather) against the copy the strided kernel needs
(index_select + member-major layout).  30 members x 13 levels x 721 x 1440 f32.

  python tools/ens_gather_bench.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from weatherbench2_amd import engine, plan as plan_lib  # noqa: E402


def main():
  dev = torch.device('cuda', 0)
  n_lat, n_lon, n_lev, m, n_pool = 721, 1440, 13, 30, 40 * 13 * 4
  pl = plan_lib.build_plan(
      np.linspace(-90, 90, n_lat), np.linspace(0, 360, n_lon, endpoint=False),
      plan_lib.LATLON, {'global': None}, dev,
      rows_per_chunk=plan_lib.ENSEMBLE_ROWS_PER_CHUNK)
  gen = torch.Generator(device=dev).manual_seed(1)
  pool = torch.randn((n_pool, n_lat, n_lon), generator=gen, device=dev)  # 8.6 GB
  truth = torch.randn((n_lev, n_lat, n_lon), generator=gen, device=dev)
  rs = np.random.RandomState(0)
  slab = n_lat * n_lon
  steps = 20
  tables = []
  for _ in range(steps + 3):
    years = rs.choice(n_pool // n_lev, size=m, replace=False)
    idx = years[None, :] * n_lev + np.arange(n_lev)[:, None]  # [outer, member]
    tables.append(idx.astype(np.int64))

  def gathered(idx):
    ptrs = engine.upload_table(
        np.ascontiguousarray(engine.gather_pointers(pool, idx, slab)).ravel(),
        dev).reshape(n_lev, m)
    return engine.ensemble_reduce(pl, pool, 0, m, None, truth, None, n_lev,
                                  False, member_ptrs=ptrs)[0]

  def copied(idx):
    sel = torch.as_tensor(idx.T.ravel(), device=dev)  # member-major
    ens = torch.index_select(pool, 0, sel)
    return engine.ensemble_reduce(pl, ens, n_lev * slab, m, None, truth, None,
                                  n_lev, False)[0]

  ready = [engine.upload_table(np.ascontiguousarray(
      engine.gather_pointers(pool, idx, slab)).ravel(), dev).reshape(n_lev, m)
           for idx in tables]

  def gathered_kernel_only(idx, _it=iter(range(10 ** 9))):
    return engine.ensemble_reduce(pl, pool, 0, m, None, truth, None, n_lev,
                                  False,
                                  member_ptrs=ready[next(_it) % len(ready)])[0]

  out = {}
  for name, fn in (('gathered', gathered), ('copied', copied),
                   ('gathered_tables_resident', gathered_kernel_only)):
    for i in range(3):
      r = fn(tables[i])
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for i in range(steps):
      r = fn(tables[3 + i])
    ev[1].record()
    torch.cuda.synchronize()
    out[name + '_ms_per_chunk'] = ev[0].elapsed_time(ev[1]) / steps
    out[name + '_check'] = float(r.sum().item())
  a, b = gathered(tables[0]), copied(tables[0])
  out['bit_identical'] = bool(torch.equal(a, b))
  pts = n_lev * slab
  out['gathered_TBps'] = pts * (m + 1) * 4 / out['gathered_ms_per_chunk'] / 1e9
  out['config'] = f'{m} members x {n_lev} levels x {n_lat} x {n_lon} f32'
  print(json.dumps(out))


if __name__ == '__main__':
  main()
