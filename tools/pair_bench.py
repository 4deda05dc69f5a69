"""K1p against the two launches it replaces, at the official chunk's size.

  python tools/pair_bench.py [--chunks 4] [--reps 5] [--json OUT]
  python tools/pair_bench.py --variants      # every build/variants/libwb2hip_pair_*.so

One "chunk" = 85 slabs of 721 x 1440 float32 (forecast, truth, climatology),
14 of them u / v pairs (13 levels + the 10 m pair), the 16 official regions (13
slices + 3 land-mask regions: a 2-D float32 weight field).  Times, with HIP
events on the launch stream, interleaved:
  separate   wb2_det_suite_step(DET_ACC, 85 slabs) + wb2_det_suite_step(WIND,
             14 pairs)                      -- 1 058 + 233 MB read
  pairs      wb2_det_wind_suite_step(DET_ACC, 85 slabs, 14 pairs) -- 1 058 MB
Inputs rotate over `--chunks` pools (>> the 256 MiB Infinity Cache).
"""
import argparse
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def official_regions(lat, lon):
  import numpy as np
  from tests import helpers
  from weatherbench2_amd import regions as gr
  from weatherbench2_amd import xarray_lite as xl
  regions = helpers.predefined_regions(oracle=False)
  rs = np.random.RandomState(3)
  lsm = np.clip(rs.uniform(-0.5, 1.2, size=(len(lat), len(lon))), 0.0,
                1.0).astype(np.float32)
  mask = xl.DataArray(lsm, ('latitude', 'longitude'),
                      {'latitude': lat, 'longitude': lon})
  regions['global_land'] = gr.LandRegion(mask)
  regions['tropics_land'] = gr.CombinedRegion(
      [gr.SliceRegion(lat_slice=slice(-20, 20)), gr.LandRegion(mask)])
  regions['extra-tropics_land'] = gr.CombinedRegion(
      [gr.SliceRegion(lat_slice=[slice(None, -20), slice(20, None)]),
       gr.LandRegion(mask)])
  return regions


def run(args):
  import numpy as np
  import torch
  from weatherbench2_amd import _lib, engine, plan as plan_lib
  dev = torch.device('cuda')
  n_lat, n_lon = 721, 1440
  lat = np.linspace(-90, 90, n_lat)
  lon = np.linspace(0, 360, n_lon, endpoint=False)
  regions = official_regions(lat, lon) if not args.no_field else None
  if args.no_field:
    from tests import helpers
    regions = helpers.predefined_regions(oracle=False)
  pl = plan_lib.build_plan(lat, lon, plan_lib.LATLON, regions, dev,
                           rows_per_chunk=int(os.environ.get("WB2HIP_EVALUATE_ROWS", 48)))
  n_outer, n_pair = 85 * args.window, 14 * args.window
  pools = []
  for c in range(args.chunks):
    gen = torch.Generator(device=dev).manual_seed(c)
    pools.append([torch.randn((n_outer, n_lat, n_lon), dtype=torch.float32,
                              device=dev, generator=gen) for _ in range(3)])
  ident = torch.arange(n_outer, dtype=torch.int64, device=dev)
  first = n_outer - 2 * n_pair
  u_tab = ident[first:first + n_pair].contiguous()
  v_tab = ident[first + n_pair:].contiguous()
  det = engine.SuiteStep(pl, _lib.MODE_DET_ACC, torch.float32, False, n_outer)
  wind = engine.SuiteStep(pl, _lib.MODE_WIND, torch.float32, False, n_pair)
  pair = engine.PairSuiteStep(pl, _lib.MODE_DET_ACC, torch.float32, False,
                              n_outer, n_pair)

  def separate(pool):
    det.run(pool, [ident, ident, ident])
    wind.run([pool[0], pool[1], pool[0], pool[1]],
             [u_tab, u_tab, v_tab, v_tab])

  def det_only(pool):
    det.run(pool, [ident, ident, ident])

  def pairs(pool):
    pair.run(pool, [ident, ident, ident])
  legs = {'separate': separate, 'det_only': det_only, 'pairs': pairs}
  if args.legs:
    legs = {k: v for k, v in legs.items() if k in args.legs.split(',')}
  times = {k: [] for k in legs}
  for fn in legs.values():
    fn(pools[0])
  torch.cuda.synchronize()
  for rep in range(args.reps):
    for name, fn in legs.items():
      start, stop = torch.cuda.Event(True), torch.cuda.Event(True)
      start.record()
      for pool in pools:
        fn(pool)
      stop.record()
      stop.synchronize()
      times[name].append(start.elapsed_time(stop) / len(pools))
  alg = 85 * 3 * n_lat * n_lon * 4
  out = {'lib': os.path.basename(_lib.lib_path()), 'field': not args.no_field,
         'window': args.window,
         'algorithmic_bytes_per_chunk': alg}
  for name, ts in times.items():
    ms = float(np.median(ts))
    ms, ts = ms / args.window, [x / args.window for x in ts]
    out[name] = {'ms_per_chunk': round(ms, 4),
                 'min_ms': round(min(ts), 4), 'max_ms': round(max(ts), 4),
                 'frac_of_8TBps_strict': round(alg / (ms * 1e-3) / 8e12, 4)}
  # parity on the way: the bits of the separate launches
  a = det.run(pools[0], [ident, ident, ident]).clone()
  w = wind.run([pools[0][0], pools[0][1], pools[0][0], pools[0][1]],
               [u_tab, u_tab, v_tab, v_tab]).clone()
  b, bw = pair.run(pools[0], [ident, ident, ident])
  same = lambda x, y: bool(((x == y) | (torch.isnan(x) & torch.isnan(y))).all())
  out['bit_identical'] = same(a, b) and same(w[:2], bw[:2])
  print(json.dumps(out))
  if args.json:
    with open(args.json, 'a') as f:
      f.write(json.dumps(out) + '\n')


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--chunks', type=int, default=4)
  ap.add_argument('--reps', type=int, default=5)
  ap.add_argument('--json')
  ap.add_argument('--no-field', action='store_true')
  ap.add_argument('--variants', action='store_true')
  ap.add_argument('--legs', help='comma-separated subset of the legs')
  ap.add_argument('--window', type=int, default=1,
                  help='chunks per launch (85 slabs, 14 pairs each)')
  args = ap.parse_args()
  if not args.variants:
    return run(args)
  libs = [None] + sorted(glob.glob(os.path.join(ROOT, 'build', 'variants',
                                                'libwb2hip_pair_*.so')))
  for rep in range(2):
    for lib in libs:
      env = dict(os.environ)
      if lib:
        env['WB2HIP_LIB'] = lib
      cmd = [sys.executable, os.path.abspath(__file__), '--chunks',
             str(args.chunks), '--reps', str(args.reps), '--window',
             str(args.window)]
      if args.json:
        cmd += ['--json', args.json]
      if args.no_field:
        cmd += ['--no-field']
      r = subprocess.run(cmd, env=env, capture_output=True, text=True)
      print(r.stdout.strip().splitlines()[-1] if r.stdout.strip()
            else 'FAILED ' + r.stderr[-400:])


if __name__ == '__main__':
  main()
