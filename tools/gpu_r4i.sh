cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
timeout 1200 python -m pytest -x -q -m gpu tests/test_spectrum_gpu.py tests/test_reductions.py > gpurun_out/r4i/pytest.txt 2>&1; tail -12 gpurun_out/r4i/pytest.txt
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu
import time, numpy as np, torch
from weatherbench2_amd import engine
dev=torch.device('cuda',0)
n_lat,n_lon=721,1440
circ=torch.ones(n_lat,dtype=torch.float64,device=dev)
for dtype in (torch.float64, torch.float32):
  units=8 if dtype==torch.float64 else 16
  pool=3
  xs=[torch.randn((units,13,n_lat,n_lon),device=dev,dtype=dtype) for _ in range(pool)]
  for mode in ('materialise','time_mean'):
    def step(i):
      if mode=='materialise': return engine.zonal_spectrum(xs[i%pool],circ,n_lat)
      return engine.zonal_spectrum(xs[i%pool],circ,n_lat,n_time=units)
    for i in range(5): step(i)
    torch.cuda.synchronize(); t0=time.perf_counter()
    n=20
    for i in range(n): step(i)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/n
    pts=units*13*n_lat*n_lon
    isz=8 if dtype==torch.float64 else 4
    nbytes=pts*isz+(pts/n_lon*721*8 if mode=='materialise' else pts/units/n_lon*721*8)
    print(dtype, mode, f'{dt*1e3:.3f} ms', f'{nbytes/dt/1e9:.0f} GB/s', f'frac {nbytes/dt/8e12:.3f}')
PY
