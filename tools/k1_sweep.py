"""Kernel-tuning sweep for K1 (run on the GPU box; see profiles/).

  python tools/k1_sweep.py            # runs every variant lib in build/variants
"""
import ctypes, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
  import numpy as np, torch
  import bench
  from weatherbench2_amd import _lib, engine, plan as plan_lib
  dev = torch.device('cuda', 0)
  units, pool = 16, 48
  lat = np.linspace(-90, 90, bench.N_LAT)
  lon = np.linspace(0, 360, bench.N_LON, endpoint=False)
  gen = torch.Generator(device=dev).manual_seed(0)
  mk = lambda: torch.randn((pool * 13, 721, 1440), generator=gen, device=dev)
  f, t, c = mk(), mk(), mk()
  res = {}
  for rpc in [int(x) for x in os.environ.get('RPC', '8,16,32,64').split(',')]:
    for nreg in ('13', '1'):
      regions = bench.predefined_regions() if nreg == '13' else None
      pl = plan_lib.build_plan(lat, lon, plan_lib.LATLON, regions, dev, rows_per_chunk=rpc)
      lev = torch.arange(13, device=dev)
      times = []
      for i in range(12):
        u = (i * units + torch.arange(units, device=dev)) % pool
        fu = (u[:, None] * 13 + lev[None]).reshape(-1).contiguous()
        tu = (((u + 7) % pool)[:, None] * 13 + lev[None]).reshape(-1).contiguous()
        cu = (((u * 5 + 3) % pool)[:, None] * 13 + lev[None]).reshape(-1).contiguous()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        engine.set_launch_hook(lambda w, k, e=(e0, e1): e[0 if w == "begin" else 1].record())
        engine.stream_reduce(pl, _lib.MODE_DET_ACC, [f, t, c], [fu, tu, cu], units * 13, False)
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
      ms = float(np.median(times[2:]))
      res[f'rpc{rpc}_reg{nreg}'] = round(units * bench.PTS_PER_UNIT * 12 / ms / 1e6, 1)
  print(json.dumps(res))


def membw():
  import torch
  lib = ctypes.CDLL(os.path.join(ROOT, 'build', 'membw.so'))
  dev = torch.device('cuda', 0)
  n = 1 << 29  # 2 GiB per array
  a, b, c = (torch.randn(n, device=dev) for _ in range(3))
  out = torch.zeros(4, device=dev)
  for narr in (1, 3):
    for nt in (0, 1):
      for blocks in (1024, 2048, 4096, 8192):
        ts = []
        for _ in range(6):
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record()
          lib.membw_read(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                         ctypes.c_void_p(c.data_ptr()), ctypes.c_longlong(n // 4), narr, nt, blocks,
                         ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
          e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[len(ts) // 2]
        print(f'membw narr={narr} nt={nt} blocks={blocks}: {narr * n * 4 / ms / 1e6:.0f} GB/s')


if __name__ == '__main__':
  if len(sys.argv) > 1 and sys.argv[1] == 'child':
    child()
  elif len(sys.argv) > 1 and sys.argv[1] == 'membw':
    membw()
  else:
    libs = sorted(glob.glob(os.path.join(ROOT, 'build', 'variants', '*.so')))
    for lib in libs:
      env = dict(os.environ, WB2HIP_LIB=lib)
      r = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True)
      print(os.path.basename(lib), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
