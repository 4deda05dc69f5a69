# One-off GPU checks of a kernel change (edit freely; not part of the product):
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
timeout 1200 python -m pytest -x -q -m gpu tests/test_ens_gpu.py tests/test_ens_exact_gpu.py tests/test_fuzz_gpu.py tests/test_evalall.py tests/test_reference_vectors.py > gpurun_out/ab/pytest.txt 2>&1; tail -4 gpurun_out/ab/pytest.txt
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu
import sys, json, torch
sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import numpy as np
import bench
from weatherbench2_amd import engine, plan as plan_lib
dev=torch.device('cuda',0)
lat=np.linspace(-90,90,721); lon=np.linspace(0,360,1440,endpoint=False)
pl=plan_lib.build_plan(lat,lon,plan_lib.LATLON,bench.predefined_regions(),dev,rows_per_chunk=plan_lib.ENSEMBLE_ROWS_PER_CHUNK)
gen=torch.Generator(device=dev).manual_seed(1)
for m in (12, 24, 28, 36, 40, 48, 60, 64, 100):
  pool=3; n_slab=13
  ens=torch.randn((m,pool*n_slab,721,1440),generator=gen,device=dev); truth=torch.randn((pool*n_slab,721,1440),generator=gen,device=dev)
  tabs=[torch.arange(n_slab,device=dev)+k*n_slab for k in range(pool)]
  def step(i):
    engine.ensemble_reduce(pl,ens,pool*n_slab*721*1440,m,tabs[i%pool],truth,tabs[i%pool],n_slab,False)
  for i in range(3): step(i)
  bench.ramp(lambda: step(0), 20.0)
  timer=bench.KernelTimer(); engine.set_launch_hook(timer)
  for i in range(20): step(i)
  engine.set_launch_hook(None); torch.cuda.synchronize()
  ms=timer.mean_ms(); nb=n_slab*721*1440*(m+1)*4
  print(m, round(ms,4), 'frac', round(nb/ms/1e6/8000,3))
  del ens, truth
PY
