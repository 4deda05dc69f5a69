"""The tier-2 kernels at the BASELINE launch
sizes -- 16 units = 208 slabs of 721 x 1440 float32 per call for the
deterministic ones (configs[1], like the headline), 13 slabs x 50 members for
the ensemble ones (configs[2]) --, each against the HBM roofline of its own
algorithmic bytes:

  spatial_maps            SpatialBias/MSE/MAE maps      read 8 B/pt, write 12
  spatial_accumulate      the same, summed over 16 time steps
                                                        read 8 B/pt.time + 48 B/pt
  seeps_map               SpatialSEEPS                  read 12 (+ p1), write 8
  gaussian_crps           K1 mode GAUSS                 read 12 B/pt
  gaussian_thresholds     K1 mode GAUSS_THR             read 16 B/pt
  seeps                   K1 mode SEEPS                 read 12 B/pt (+ p1 from L2)
  energy_score            score + spread + skill, M = 50, fused (one read of
                          the ensemble)                 read (M + 1) * 4 B/pt
  ens_thresholds          Brier / RPS partials, M = 50  read (M + 2) * 4 B/pt
  ens_threshold_maps      the same, unreduced           + 32 B/pt written
  rank_histogram          M = 50, 51 bins, mean over the 13 slabs
                                                        read (M + 1) * 4 B/pt
                                                        (+ the 408 B/pt counts,
                                                        once per call)
  rank_histogram_onehot   the per-sample one-hot form   read (M + 1) * 4, write 408
  axis_moments            mean over 16 x 13 slabs       read 4 B/pt.slab, write 16 B/pt

  python tools/tier2_variants.py          -> one JSON line

Every variant: `reps` repetitions of `calls` calls between two HIP events on
the launch stream (the K2 fold of the reducing ones included), repetitions of
all variants interleaved, median with min / max.  Inputs come from pools larger
than the 256 MiB Infinity Cache.  `bench.py` embeds the object as
`tier2_variants`.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def variants(dev, reps: int = 3, calls: int = 10, only=None) -> dict:
  import torch
  import bench
  from weatherbench2_amd import _lib, engine, plan as plan_lib
  n_lat, n_lon, n_member = bench.N_LAT, bench.N_LON, 50
  n_slab = 16 * 13   # deterministic kernels: 16 units per call
  n_eslab = 13       # ensemble kernels: one unit of 50 members
  n_point = n_lat * n_lon
  pts, epts = n_slab * n_point, n_eslab * n_point
  lat = np.linspace(-90, 90, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  gen = torch.Generator(device=dev).manual_seed(9)
  regions = bench.predefined_regions()
  plan = plan_lib.build_plan(lat, lon, plan_lib.LATLON, regions, dev)
  eplan = plan_lib.build_plan(lat, lon, plan_lib.LATLON, regions, dev,
                              rows_per_chunk=plan_lib.ENSEMBLE_ROWS_PER_CHUNK)
  pool = 2  # 2 x 208 slabs x 4.15 MB = 1.7 GB per input
  shape = (pool * n_slab, n_lat, n_lon)
  randn = lambda *s: torch.randn(s, generator=gen, device=dev)
  tabs = [torch.arange(n_slab, device=dev) + k * n_slab for k in range(pool)]

  def det_inputs(n):
    return [randn(*shape) for _ in range(n)]

  makers = {}

  def maker(name, nbytes):
    def deco(fn):
      makers[name] = (fn, nbytes)
      return fn
    return deco

  @maker('spatial_maps', pts * 20.0)
  def _():
    f, t = det_inputs(2)
    return lambda i: engine.spatial_maps(f, tabs[i % pool], t, tabs[i % pool],
                                         n_slab, n_point)

  n_time, n_rest = 16, 13  # 16 time steps of 13 levels into one set of maps

  @maker('spatial_accumulate', n_rest * n_point * (n_time * 8.0 + 48.0))
  def _():
    f = randn(pool * n_time * n_rest, n_lat, n_lon)
    t = randn(pool * n_time * n_rest, n_lat, n_lon)  # truth at valid time
    total = torch.zeros((3, n_rest, n_point), dtype=torch.float64, device=dev)
    f_tabs = [torch.arange(n_time * n_rest, device=dev) + k * n_time * n_rest
              for k in range(pool)]
    return lambda i: engine.spatial_accumulate(
        f, f_tabs[i % pool], t, f_tabs[i % pool], n_time, n_rest, n_point,
        False, total, None)

  p1 = torch.rand((n_point,), generator=gen, device=dev,
                  dtype=torch.float64) * 0.7 + 0.1

  @maker('seeps_map', pts * 20.0)
  def _():
    ins = [x.abs() for x in det_inputs(3)]
    return lambda i: engine.seeps_map(ins, [tabs[i % pool]] * 3, n_slab,
                                      n_point, p1, 0.25)

  def k1(mode, n_in, aux=None, scalar=0.0, positive=False):
    def make():
      ins = det_inputs(n_in)
      if positive:
        ins = [x.abs() + 0.1 for x in ins]
      return lambda i: engine.stream_reduce(
          plan, mode, ins, [tabs[i % pool]] * n_in, n_slab, False, aux=aux,
          scalar=scalar)
    return make
  makers['gaussian_crps'] = (k1(_lib.MODE_GAUSS, 3, positive=True), pts * 12.0)
  makers['gaussian_thresholds'] = (k1(_lib.MODE_GAUSS_THR, 4, positive=True),
                                   pts * 16.0)
  makers['seeps'] = (k1(_lib.MODE_SEEPS, 3, aux=p1.reshape(n_lat, n_lon),
                        scalar=0.25, positive=True), pts * 12.0)

  epool = 3
  etabs = [torch.arange(n_eslab, device=dev) + k * n_eslab
           for k in range(epool)]
  stride = epool * n_eslab * n_point

  def ens_inputs():
    ens = torch.randn((n_member, epool * n_eslab, n_lat, n_lon), generator=gen,
                      device=dev)
    truth = randn(epool * n_eslab, n_lat, n_lon)
    return ens, truth

  @maker('energy_score', epts * (n_member + 1) * 4.0)
  def _():
    ens, truth = ens_inputs()
    gplan = plan_lib.build_plan(lat, lon, plan_lib.LATLON, regions, dev,
                                rows_per_chunk=plan_lib.ENERGY_ROWS_PER_CHUNK)
    return lambda i: engine.energy_score(
        gplan, ens, stride, n_member, etabs[i % epool], truth,
        etabs[i % epool], n_eslab, False)

  @maker('ens_thresholds', epts * (n_member + 2) * 4.0)
  def _():
    ens, truth = ens_inputs()
    thr = randn(epool * n_eslab, n_lat, n_lon)
    return lambda i: engine.ensemble_threshold_reduce(
        eplan, ens, stride, n_member, etabs[i % epool], truth,
        etabs[i % epool], thr, etabs[i % epool], n_eslab, False)

  @maker('ens_threshold_maps', epts * ((n_member + 2) * 4.0 + 32.0))
  def _():
    ens, truth = ens_inputs()
    thr = randn(epool * n_eslab, n_lat, n_lon)
    return lambda i: engine.ensemble_threshold_maps(
        ens, stride, n_member, etabs[i % epool], truth, etabs[i % epool], thr,
        etabs[i % epool], n_eslab, n_point, False)

  n_bins = n_member + 1

  @maker('rank_histogram', epts * (n_member + 1) * 4.0 + n_point * n_bins * 8.0)
  def _():
    ens, truth = ens_inputs()
    return lambda i: engine.rank_histogram(
        ens, stride, n_member, etabs[i % epool], truth, etabs[i % epool],
        n_eslab, n_point, n_bins, True, 1234 + i, mean_over=(1, n_eslab, 1))

  @maker('rank_histogram_onehot',
         epts * ((n_member + 1) * 4.0 + n_bins * 8.0))
  def _():
    ens, truth = ens_inputs()
    return lambda i: engine.rank_histogram(
        ens, stride, n_member, etabs[i % epool], truth, etabs[i % epool],
        n_eslab, n_point, n_bins, True, 1234 + i)

  n_red = 16

  @maker('axis_moments', (n_red * 4.0 + 16.0) * epts)
  def _():
    x = randn(2, n_red, n_eslab * n_point)
    return lambda i: engine.axis_moments(x[i % 2], 1, n_red,
                                         n_eslab * n_point, None, False)

  names = [n for n in makers if not only or n in only]
  samples = {n: [] for n in names}
  for _ in range(reps):
    for name in names:
      make, _ = makers[name]
      step = make()
      it = [0]

      def one():
        it[0] += 1
        return step(it[0])
      for _ in range(3):
        one()
      bench.ramp(one, 20.0)
      a, b = (torch.cuda.Event(enable_timing=True),
              torch.cuda.Event(enable_timing=True))
      a.record()
      for _ in range(calls):
        one()
      b.record()
      torch.cuda.synchronize()
      samples[name].append(a.elapsed_time(b) / calls)
      del step
      torch.cuda.empty_cache()
  out = {}
  for name in names:
    ms = sorted(samples[name])
    med, nbytes = ms[len(ms) // 2], makers[name][1]
    frac = lambda t: nbytes / t / 1e6 / bench.HBM_PEAK_GBPS
    out[name] = {'ms_per_call': med, 'frac': frac(med), 'frac_min': frac(ms[-1]),
                 'frac_max': frac(ms[0]), 'repetitions': len(ms),
                 'algorithmic_bytes_per_call': nbytes}
  return out


def main():
  import argparse
  import torch
  ap = argparse.ArgumentParser()
  ap.add_argument('--only', default='')
  ap.add_argument('--reps', type=int, default=3)
  args = ap.parse_args()
  only = [x for x in args.only.split(',') if x] or None
  print(json.dumps(variants(torch.device('cuda', 0), reps=args.reps,
                            only=only)))


if __name__ == '__main__':
  main()
