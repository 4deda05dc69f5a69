"""K3's production instantiations at the BASELINE configs[2] launch size (13
slabs of 721 x 1440 float32 per launch): the official `probabilistic` config's
16 regions incl. three land-sea-mask regions (scripts/evaluate.py:345-395,
496-520: the WF = true instantiation), skipna (clean data: every wave takes
the NaN-free fast path; with NaN patches: 1 in 8 column tiles takes the general
path), a global-only region set (`regions=None`), the member counts with
kernels of their own (4 ... 100: csrc/ensemble_exact.hip, one object per entry of
WB2_SORT3_SIZES), one runtime-M count (44) and float64.

  python tools/k3_variants.py         -> one JSON line

Every variant is timed `reps` times (30 launches each, HIP events on the launch
stream), the repetitions of all variants interleaved; the line carries the
median with min / max.  `bench.py` embeds the same object as `k3_variants`.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)



def variants(dev, reps: int = 3, launches: int = 30, only=None) -> dict:
  import torch
  import bench
  from weatherbench2_amd import build, engine, plan as plan_lib
  exact_sizes = [m for m, _ in build.exact_sizes()]
  n_lat, n_lon, n_slab = bench.N_LAT, bench.N_LON, 13
  lat = np.linspace(-90, 90, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  rows = plan_lib.ENSEMBLE_ROWS_PER_CHUNK
  plans = {}

  def plan(name):
    if name not in plans:
      regions = {'slice13': bench.predefined_regions,
                 'official16': bench.official_regions,
                 'global': lambda: {'global': None}}[name]()
      plans[name] = plan_lib.build_plan(lat, lon, plan_lib.LATLON, regions,
                                        dev, rows_per_chunk=rows)
    return plans[name]
  gen = torch.Generator(device=dev).manual_seed(5)
  slab = n_lat * n_lon
  specs = [('headline_slice13', 'slice13', 50, torch.float32, False, 0.0),
           ('official16_landmask', 'official16', 50, torch.float32, False, 0.0),
           ('global_only', 'global', 50, torch.float32, False, 0.0),
           ('skipna', 'slice13', 50, torch.float32, True, 0.0),
           ('skipna_nan_patches', 'slice13', 50, torch.float32, True, 0.125)]
  specs += [(f'members{m}', 'slice13', m, torch.float32, False, 0.0)
            for m in exact_sizes]
  # member counts without a program of their own: hosted by the next larger
  # one (ens_point_hosted; with skipna the dead slots are NaN members of the
  # general exact code)
  specs += [(f'members{m}_hosted', 'slice13', m, torch.float32, False, 0.0)
            for m in (7, 13, 24, 33, 44, 47, 63, 77)]
  specs += [('members51_skipna', 'slice13', 51, torch.float32, True, 0.0),
            ('members44_skipna_hosted', 'slice13', 44, torch.float32, True,
             0.0),
            ('f64_members50', 'slice13', 50, torch.float64, False, 0.0)]
  if only:
    specs = [s for s in specs if s[0] in only]
  samples = {s[0]: [] for s in specs}
  for _ in range(reps):
    for name, pname, m, dtype, skipna, nan_tiles in specs:
      # pool: distinct slab sets so that the 256 MiB Infinity Cache cannot
      # serve a re-read (one set of 50 float32 members is 2.7 GB)
      pool = 3 if m * np.dtype(str(dtype).split('.')[-1]).itemsize <= 128 else 2
      ens = torch.randn((m, pool * n_slab, n_lat, n_lon), generator=gen,
                        device=dev, dtype=dtype)
      if nan_tiles:  # NaN patches: whole 64-column tiles of some members
        n_tile = n_lon // 64
        hit = torch.rand((pool * n_slab, n_lat, n_tile), generator=gen,
                         device=dev) < nan_tiles
        mask = hit.repeat_interleave(64, dim=2)
        ens[3, :, :, :n_tile * 64][mask] = float('nan')
      truth = torch.randn((pool * n_slab, n_lat, n_lon), generator=gen,
                          device=dev, dtype=dtype)
      tabs = [torch.arange(n_slab, device=dev) + k * n_slab
              for k in range(pool)]
      stride = pool * n_slab * slab
      it = [0]

      def step():
        it[0] += 1
        tab = tabs[it[0] % pool]
        engine.ensemble_reduce(plan(pname), ens, stride, m, tab, truth, tab,
                               n_slab, skipna)
      for _ in range(3):
        step()
      bench.ramp(step, 20.0)
      timer = bench.KernelTimer()
      engine.set_launch_hook(timer)
      for _ in range(launches):
        step()
      engine.set_launch_hook(None)
      torch.cuda.synchronize()
      samples[name].append(timer.mean_ms())
      del ens, truth
      torch.cuda.empty_cache()
  out = {}
  for name, pname, m, dtype, skipna, nan_tiles in specs:
    ms = sorted(samples[name])
    med = ms[len(ms) // 2]
    nbytes = n_slab * slab * (m + 1) * (4 if dtype == torch.float32 else 8)
    frac = lambda t: nbytes / t / 1e6 / bench.HBM_PEAK_GBPS
    out[name] = {'kernel_ms': med, 'frac': frac(med),
                 'frac_min': frac(ms[-1]), 'frac_max': frac(ms[0]),
                 'repetitions': len(ms), 'members': m,
                 'dtype': str(dtype).split('.')[-1], 'skipna': skipna,
                 'regions': plan(pname).n_region,
                 'algorithmic_bytes_per_launch': nbytes}
    if nan_tiles:
      out[name]['nan_column_tiles'] = nan_tiles
  return out


def main():
  import argparse
  import torch
  ap = argparse.ArgumentParser()
  ap.add_argument('--only', default='',
                  help='comma-separated variant names (default: all)')
  ap.add_argument('--reps', type=int, default=3)
  args = ap.parse_args()
  only = [s for s in args.only.split(',') if s] or None
  print(json.dumps(variants(torch.device('cuda', 0), reps=args.reps,
                            only=only)))


if __name__ == '__main__':
  main()
