# Round 3: the halving-tree fold -- whole GPU suite + K1 timings (headline and variants)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -8 | tee $O/pytest.txt
for rep in 1 2; do
  timeout 300 python - <<PY | tee -a $O/variants.txt
import json, sys, torch
sys.path.insert(0, '.')
import bench
import numpy as np
dev = torch.device('cuda', 0)
gen = torch.Generator(device=dev).manual_seed(1)
pool = 48
mk = lambda: torch.randn((pool * 13, 721, 1440), generator=gen, device=dev)
f, t, c = mk(), mk(), mk()
from weatherbench2_amd import _lib, engine, plan as plan_lib
lat = np.linspace(-90, 90, 721); lon = np.linspace(0, 360, 1440, endpoint=False)
pl = plan_lib.build_plan(lat, lon, plan_lib.LATLON, bench.predefined_regions(), dev, rows_per_chunk=32)
lev = torch.arange(13, device=dev)
timer = bench.KernelTimer()
def tabs(s):
  u = (s * 16 + torch.arange(16, device=dev)) % pool
  return [(((u * (2 * j + 1) + 3 * j) % pool)[:, None] * 13 + lev[None]).reshape(-1).contiguous() for j in range(3)]
T = [tabs(s) for s in range(64)]
for i in range(10): engine.stream_reduce(pl, _lib.MODE_DET_ACC, [f, t, c], T[i], 208, False)
engine.set_launch_hook(timer)
for i in range(50): engine.stream_reduce(pl, _lib.MODE_DET_ACC, [f, t, c], T[10 + i], 208, False)
engine.set_launch_hook(None); torch.cuda.synchronize()
ms = timer.mean_ms()
res = ['headline=%.4f(%.3f)' % (ms, 16 * 13 * 721 * 1440 * 12 / ms / 1e6 / 8000)]
out = bench.k1_variants(dev, f, t, c, 16, pool)
res += ['%s=%.4f(%.3f)' % (k, v['kernel_ms'], v['frac']) for k, v in out.items()]
print(' '.join(res))
PY
done
for w in ensemble; do timeout 120 python bench.py --workload $w --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$w step_ms=%.4f kernel_ms=%.4f frac=%.3f' % (d['ms_per_step'], r['kernel_ms'], r['frac']))" | tee -a $O/variants.txt; done
