"""K4f at the row lengths added in round 6 (third session) against the hipFFT
path of the same lengths: ms per launch and fraction of the HBM peak on the
materialised spectrum's 8 B/pt, float32, 16 x 13 fields."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHILD = r'''
import json, sys, numpy as np, torch
sys.path.insert(0, %r)
from weatherbench2_amd import engine
dev = torch.device("cuda")
out = {}
for n_lon in %r:
  n_lat = n_lon // 2 + 1
  n_field = max(1, int(2.2e8 // (n_lat * n_lon)))
  x = torch.randn((n_field, n_lat, n_lon), device=dev)
  circ = torch.ones(n_lat, dtype=torch.float64, device=dev)
  for _ in range(3): engine.zonal_spectrum(x, circ, n_lat)
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(10): y = engine.zonal_spectrum(x, circ, n_lat)
  b.record(); torch.cuda.synchronize()
  ms = a.elapsed_time(b) / 10
  pts = n_field * n_lat * n_lon
  out[n_lon] = {"ms": ms, "frac": pts * 8.0 / (ms * 1e-3) / 8e12}
  del x, y
print(json.dumps(out))
'''
sizes = [int(a) for a in sys.argv[1:]] or [96, 288, 320, 384, 480, 640, 768, 1280, 1800, 2048, 2560, 2880, 3600, 1440]
res = {}
for backend in ('fused', 'rocfft'):
  env = dict(os.environ)
  if backend == 'rocfft':
    env['WB2HIP_SPECTRUM_BACKEND'] = 'rocfft'
  r = subprocess.run([sys.executable, '-c', CHILD % (ROOT, sizes)], env=env,
                     capture_output=True, text=True, timeout=600)
  res[backend] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else r.stderr[-500:]
print(json.dumps(res))
if isinstance(res['fused'], dict) and isinstance(res['rocfft'], dict):
  for n in sizes:
    f, h = res['fused'][str(n)], res['rocfft'][str(n)]
    print(f"N={n:5d}  fused {f['ms']:.3f} ms {f['frac']:.3f}   hipFFT path {h['ms']:.3f} ms {h['frac']:.3f}")
