import ctypes, os, sys, itertools
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, 'build', 'pattern.so'))
dev = torch.device('cuda', 0)
rows = 208 * 721
a, b, c = (torch.randn(rows * 1440, device=dev) for _ in range(3))
out = torch.zeros(4, device=dev); counter = torch.zeros(4, dtype=torch.int32, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def run(rpi, mode, U, nt, grid, lds):
  n_items = rows // rpi
  g = n_items if mode == 0 else min(grid, n_items)
  ts = []
  for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.probe_launch(P(a), P(b), P(c), n_items, rpi, mode, U, nt, g, lds, P(counter), P(out),
                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    assert rc == 0
  ms = sorted(ts)[2]
  return n_items * rpi * 1440 * 12 / ms / 1e6
print('mode0 = one item per block; mode1 = persistent static; mode2 = persistent dynamic; lds limits WGs/CU')
for nt in (1,):
  for U in (1, 2, 4):
    for rpi in (4, 8, 16, 32):
      row = []
      for lds_kb in (0, 40, 64):   # 0 -> VGPR-limited (5 WG/CU), 40 KB -> 4/CU, 64 KB -> 2/CU
        row.append(f'm0/lds{lds_kb}={run(rpi, 0, U, nt, 0, lds_kb * 1024):.0f}')
      for grid in (512, 768, 1024, 1280):
        row.append(f'm2/g{grid}={run(rpi, 2, U, nt, grid, 0):.0f}')
      row.append(f'm1/g1024={run(rpi, 1, U, nt, 1024, 0):.0f}')
      print(f'nt={nt} U={U} rpi={rpi}: ' + ' '.join(row), flush=True)
